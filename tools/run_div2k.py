"""dev: bench.div2k_image alone (the 2040x1356 tiling driver: eager, graph, launch chain)"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import bench
import control_gic_amd as cg
dev = torch.device("cuda", 0)
cb = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
r = bench.div2k_image(dev, cb, vq, codec, iters=int(sys.argv[1]) if len(sys.argv) > 1 else 8)
for k in ("ms_per_image", "graph_replay", "chain", "four_in_flight", "batch_of_8", "batch_of_8_uint8_frames"):
    print(k, json.dumps(r.get(k)))
