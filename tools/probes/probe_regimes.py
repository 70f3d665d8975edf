import sys, json; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench
dev = torch.device('cuda', 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1)
r = bench.input_regimes(dev, x, (0.1, 0.8))
for k, v in r.items(): print(k, json.dumps(v))
