"""Refinement queues A/B (round 5): per content family, masks with the launch's refinement queues == masks without them == masks
of the router on reference-arithmetic maps; time of the fused VQ + router launch and of the stand-alone router, queues on / off /
no refinement.  Usage: python tools/probes/probe_refine_queue.py [--fast]"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
from bench import graph_kernel_time

dev = torch.device("cuda", 0)
ratio = (0.1, 0.8)
fast = "--fast" in sys.argv
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
router = cg.TripleGrainFixedEntropyRouter(*ratio, per_image=True)
out = {}


def masks_of(xd, zd, e8, e16, queues, fused=True):
    _lib.REFINE_QUEUES = queues
    if fused:
        r = vq_forward_route(zd, cb, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True, pixels=xd)
        m = r[3]
    else:
        m = router(e16, e8, want_gate=False, pixels=xd)[0]
    torch.cuda.synchronize()
    return [t.clone() for t in m]


def one(name, x, z):
    xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    r8, r16 = cg.entropy_maps(xd, reference_order=True)
    ref = router(r16, r8, want_gate=False, pixels=None)[0]
    res = {}
    for fused in (True, False):
        for q in (False, True):
            for rep in range(3 if q else 1):
                m = masks_of(xd, zd, e8, e16, q, fused)
                d = sum(int((a != b).sum()) for a, b in zip(m, ref))
                res[f"diff_vs_refmaps_{'fused' if fused else 'router'}_{'q' if q else 'noq'}"] = max(d, res.get(f"diff_vs_refmaps_{'fused' if fused else 'router'}_{'q' if q else 'noq'}", 0))
    for q in (False, True):
        _lib.REFINE_QUEUES = q
        tag = "q" if q else "noq"
        res[f"fused_us_{tag}"] = round(graph_kernel_time(lambda: vq_forward_route(zd, cb, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True, pixels=xd), per_graph=5, reps=3), 2)
        res[f"router_us_{tag}"] = round(graph_kernel_time(lambda: router(e16, e8, want_gate=False, pixels=xd), per_graph=5, reps=3), 2)
    res["fused_us_norefine"] = round(graph_kernel_time(lambda: vq_forward_route(zd, cb, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True, pixels=None), per_graph=5, reps=3), 2)
    res["router_us_norefine"] = round(graph_kernel_time(lambda: router(e16, e8, want_gate=False, pixels=None), per_graph=5, reps=3), 2)
    out[name] = res
    print(name, json.dumps(res), flush=True)


z = rng.standard_normal((64, 4, 64, 64)).astype(np.float32)
fam = families(n=64)
fam["rand_f32"] = rng.random((64, 3, 256, 256)).astype(np.float32)
names = [a for a in sys.argv[1:] if not a.startswith("--")]
if "--one" in sys.argv:          # single images: family:index ...
    for spec in names:
        f_, i_ = spec.split(":")
        one(spec, fam[f_][int(i_):int(i_) + 1], z[:1])
    sys.exit(0)
for name in (names if names else ("smooth8", "flat_edges") if fast else ("rand_f32", "noise8", "smooth8", "flat_edges", "blocky8")):
    if name in fam:
        one(name, fam[name], z)
if names and "tiles" not in names:
    sys.exit(0)
t = families(n=2, H=768, W=768, seed=11)
tiles = np.concatenate([t[k] for k in ("noise8", "smooth8", "flat_edges", "blocky8")])
zt = np.random.default_rng(5).standard_normal((tiles.shape[0], 4, 192, 192)).astype(np.float32)
one("tiles_768", tiles, zt)
one("tile_768_smooth_only", t["smooth8"][:1], zt[:1])
os.makedirs("gpurun_out", exist_ok=True)
json.dump(out, open("gpurun_out/refine_queue.json", "w"), indent=1)
