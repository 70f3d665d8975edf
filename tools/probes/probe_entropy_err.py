"""error of the default entropy kernel against the reference-arithmetic kernel, by value bucket, over the content families"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import control_gic_amd as cg
from oracle.content_families import families
dev = torch.device("cuda", 0)
rng = np.random.default_rng(1)
sets = dict(families(n=64))
sets["rand"] = rng.random((64, 3, 256, 256)).astype(np.float32)
t = families(n=2, H=768, W=768, seed=11)
for k, v in t.items(): sets["tile_" + k] = v
# smooth variants with other noise amplitudes / channel mixes
yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
g = np.empty((64, 3, 256, 256), np.float32)
for i in range(64):
    a = rng.uniform(0, 2 * np.pi); f = rng.uniform(0.05, 3.0)
    base = rng.uniform(0.1, 0.9) + rng.uniform(0.01, 0.4) * np.sin((np.cos(a) * xx + np.sin(a) * yy) * f * 2 * np.pi / 256)
    for c in range(3):
        g[i, c] = base * rng.uniform(0.5, 1.0) + rng.integers(-2, 3, (256, 256)) / 255.0 * rng.integers(0, 2)
sets["smooth_var"] = np.round(np.clip(g, 0, 1) * 255.0).astype(np.float32) / 255.0
sets["smooth_f32"] = np.clip(g, 0, 1).astype(np.float32)
edges = [0, 1e-6, 1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 0.5, 1, 2, 4]
worst = {p: np.zeros(len(edges) - 1) for p in (8, 16)}
for name, x in sets.items():
    xd = torch.from_numpy(np.ascontiguousarray(x)).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    r8, r16 = cg.entropy_maps(xd, reference_order=True)
    for p, a, r in ((8, e8, r8), (16, e16, r16)):
        a, r = a.cpu().numpy().ravel(), r.cpu().numpy().ravel()
        err = np.abs(a.astype(np.float64) - r)
        b = np.digitize(r, edges) - 1
        for k in range(len(edges) - 1):
            m = b == k
            if m.any(): worst[p][k] = max(worst[p][k], err[m].max())
    print(name, "max err p8 %.3g p16 %.3g" % (np.abs(e8.cpu().numpy() - r8.cpu().numpy()).max(), np.abs(e16.cpu().numpy() - r16.cpu().numpy()).max()), flush=True)
for p in (8, 16):
    print("p", p, " ".join(f"[{edges[k]:g},{edges[k+1]:g}): {worst[p][k]:.2e}" for k in range(len(edges) - 1)))
