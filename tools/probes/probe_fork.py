"""GPU probe: one-batch graph with VQ on a side stream next to entropy -> router (HotPathPipeline fork_vq), vs the fused launch"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, seed=3)
for fork in (0, 1, 2):
    hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
    hp.pipe = cg.pipeline.HotPathPipeline(hp.vq, 0.1, 0.8, frequency=hp.codec.huffman, fork_vq=fork)
    g = hp.capture()
    for _ in range(5): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): g.replay()
    torch.cuda.synchronize()
    print(f"fork_vq={fork}: {1e6*(time.perf_counter()-t0)/200:.1f} us/step")
