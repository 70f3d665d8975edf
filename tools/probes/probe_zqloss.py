import os, sys
sys.path.insert(0, ".")
import numpy as np, torch
import control_gic_amd as cg, bench
from control_gic_amd.quantize import _vq_forward
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
w, prep = hp.vq.embedding.weight, hp.pipe.prepared
for rep in range(2):
    for name, zq, ls in (("idx only", False, False), ("idx + z_q", True, False), ("idx + loss", False, True), ("idx + z_q + loss", True, True)):
        t = bench.graph_kernel_time(lambda: _vq_forward(hp.z, w, 0.25, True, None, zq, ls, prepared=prep))
        print(f"{name}: {t:.2f} us", end=" | ")
    print()
