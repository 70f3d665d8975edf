"""GPU probe: the bench's stage breakdown only (standalone kernel timings)"""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, control_gic_amd as cg
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, seed=1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
print(json.dumps(bench.stage_breakdown(hp)))
