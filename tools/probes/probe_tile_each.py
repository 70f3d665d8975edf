import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import control_gic_amd as cg
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
from bench import graph_kernel_time
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
t = families(n=2, H=768, W=768, seed=11)
tiles = np.concatenate([t[k] for k in ("noise8", "smooth8", "flat_edges", "blocky8")])
names = [f"{k}{i}" for k in ("noise8", "smooth8", "flat_edges", "blocky8") for i in range(2)]
zt = torch.from_numpy(np.random.default_rng(5).standard_normal((1, 4, 192, 192)).astype(np.float32)).to(dev)
router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
for k in range(8):
    xd = torch.from_numpy(tiles[k:k + 1]).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    f = graph_kernel_time(lambda: vq_forward_route(zt, cb, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd), per_graph=5, reps=3)
    n = graph_kernel_time(lambda: vq_forward_route(zt, cb, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=None), per_graph=5, reps=3)
    r = graph_kernel_time(lambda: router(e16, e8, want_gate=False, pixels=xd), per_graph=5, reps=3)
    print(names[k], "fused", round(f, 1), "no-refine", round(n, 1), "router alone (queues)", round(r, 1), flush=True)
