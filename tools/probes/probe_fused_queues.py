"""The fused VQ + router launch per content family, with the launch's refinement queues and without: HIP-event time per launch and
whether the masks agree.  usage: [family ...]   (CGIC_LIB picks the build)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench, control_gic_amd as cg
from control_gic_amd.quantize import vq_forward_route, prepare_codebook
from oracle.content_families import families
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
fam = families(n=64)
fam["rand_f32"] = rng.random((64, 3, 256, 256)).astype(np.float32)
z = torch.from_numpy(rng.standard_normal((64, 4, 64, 64)).astype(np.float32)).to(dev)
w = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
prep = prepare_codebook(w)
for name in (sys.argv[1:] or ["rand_f32", "noise8", "smooth8", "flat_edges", "blocky8"]):
    xd = torch.from_numpy(fam[name]).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    res, masks = {}, {}
    for q in (False, True):
        f = lambda: vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=xd, refine_queues=q)
        masks[q] = [m.clone() for m in f()[3]]
        res["queues" if q else "plain"] = round(bench.graph_kernel_time(f, per_graph=10, reps=5), 2)
    res["same_masks"] = all(torch.equal(a, b) for a, b in zip(masks[False], masks[True]))
    print(name, json.dumps(res), flush=True)
