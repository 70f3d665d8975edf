"""GPU driver for profiles: the whole hot path (sequential, eager) for B images of SxS; usage: run_hotpath.py B S iters"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
B, S, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(B, S, S, seed=77)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
for _ in range(iters):
    hp.step()
torch.cuda.synchronize()
print("done")
