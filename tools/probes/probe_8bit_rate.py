"""4-in-flight LaneStream throughput on 8-bit content families vs the bench's fp32 noise, refinement queues on / off"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, control_gic_amd as cg
from control_gic_amd import _lib
from oracle.content_families import families
dev = torch.device("cuda", 0)
ratio = (0.1, 0.8)
x0, z0, cb = bench.make_inputs(64, 256, 256, 1)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
fam = families(n=64)
sets = {"rand_f32": x0, "noise8": fam["noise8"], "smooth8": fam["smooth8"], "flat_edges": fam["flat_edges"], "blocky8": fam["blocky8"]}
zd = torch.from_numpy(z0).to(dev)
out = {}
import time
def rate(xz, lanes, steps, copies, fuse):
    slots = [p for p in xz for _ in range(copies)]
    ls = cg.pipeline.LaneStream(vq, ratio[0], ratio[1], slots, lanes=lanes, frequency=codec.huffman, fuse_router=fuse)
    ls.capture()
    ls.submit(len(slots)); ls.join(); torch.cuda.synchronize()
    ls.prepare(steps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ls.submit(steps); ls.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps
for fuse in (True, False):
    for name, x in sets.items():
        xd = torch.from_numpy(x).to(dev)
        for lanes in (4, 1):
            dt = min(rate([(xd, zd)], lanes, 80, 8, fuse) for _ in range(2))
            out[f"{name}_{'fused' if fuse else 'split'}_{lanes}"] = round(64 * 65536 / dt / 1e6)
        print(name, {k: v for k, v in out.items() if k.startswith(name)}, flush=True)
json.dump(out, open("gpurun_out/rate8.json", "w"), indent=1)
