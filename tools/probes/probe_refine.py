"""GPU probe: what the router's threshold-band refinement costs per content family (fused VQ + router launch, graph-timed,
with / without the pixels / without the constant-patch map) and the stand-alone router."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import control_gic_amd as cg
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
dev = torch.device("cuda", 0)
x0, z, cb = bench.make_inputs(64, 256, 256, 1)
vq = bench.make_quantizer(dev, cb)
prep = cg.quantize.prepare_codebook(vq.embedding.weight)
sets = {"uniform_noise(bench)": x0}
sets.update(families(n=64))
t = families(n=2, H=768, W=768, seed=11)
tiles = np.concatenate([t[k] for k in ("noise8", "smooth8", "flat_edges", "blocky8")])
zt = np.random.default_rng(5).standard_normal((tiles.shape[0], 4, 192, 192), dtype=np.float32)
sets["tiles_768"] = tiles
router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
for name, x in sets.items():
    xd = torch.from_numpy(x).to(dev)
    zd = torch.from_numpy(zt if name == "tiles_768" else z).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    nan = torch.full_like(e8, float("nan"))
    f = lambda px, fl=None: bench.graph_kernel_time(lambda: vq_forward_route(zd, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, prepared=prep, pixels=px, flat8=fl), per_graph=5, reps=3)
    r = lambda px, fl=None: bench.graph_kernel_time(lambda: router(e16, e8, want_gate=False, pixels=px, flat8=fl), per_graph=5, reps=3)
    ent = bench.graph_kernel_time(lambda: cg.entropy_maps(xd), per_graph=5, reps=3)
    ent0 = bench.graph_kernel_time(lambda: cg.entropy_maps(xd, want_flat=False), per_graph=5, reps=3)
    print(f"{name:22s} fused: refine {f(xd):8.2f}  no-flat-map {f(xd, nan):8.2f}  off {f(None):7.2f} | router alone: refine {r(xd):8.2f} no-flat-map {r(xd, nan):8.2f} off {r(None):7.2f} | entropy {ent:6.2f} (no flat8 {ent0:6.2f}) us", flush=True)
