"""GPU probe: entropy kernel alone vs LDS padding, and the two-stream pipelined step rate (dbg lib + CGIC_ENT_PAD)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
import bench
from tools_probe import graph_time
dev = torch.device("cuda", 0)
slots_np = [bench.make_inputs(64, 256, 256, seed=s) for s in range(6)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
hist = torch.zeros(1024, dtype=torch.int64, device=dev)
slots = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
print("entropy alone: best %.2f mean %.2f us" % graph_time(lambda: cg.entropy_maps(slots[0][0])))
bs = cg.pipeline.BatchStream(vq, 0.1, 0.8, slots, frequency=codec.huffman, hist=hist)
bs.capture()
bs.submit(20); bs.join(); torch.cuda.synchronize()
for n in (200, 200):
    t0 = time.perf_counter(); bs.submit(n); bs.join(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pipelined: {1e6*(t2-t0)/n:.1f} us/step")
seq = bench.SequentialStream(dev, slots, cb, (0.1, 0.8), vq, codec, hist)
seq.submit(20); torch.cuda.synchronize()
t0 = time.perf_counter(); seq.submit(200); torch.cuda.synchronize(); t2 = time.perf_counter()
print(f"sequential: {1e6*(t2-t0)/200:.1f} us/step")
