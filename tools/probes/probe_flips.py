"""GPU probe: pixels -> masks on tie-heavy content, GPU entropy -> GPU router vs the reference's torch-CPU arithmetic
(oracle/entropy_torch.py) -> oracle router, per content family."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from oracle import cgic_oracle as orc, entropy_torch as et
from oracle.content_families import families
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
fam = families(n=n)
fam["rand_f32"] = np.random.default_rng(3).random((n, 3, 256, 256), dtype=np.float32)
router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
for name, x in fam.items():
    xd = torch.from_numpy(x).cuda()
    e8, e16 = cg.entropy_maps(xd)
    mask, _, _, mode = router(e16, e8)
    mk = [m.cpu().numpy() for m in mask]
    g8, g16 = e8.cpu().numpy(), e16.cpu().numpy()
    flips = imgs = flips_c = imgs_c = 0
    d8 = d16 = 0.0
    t0 = time.time()
    for b0 in range(0, n, 8):
        xt = torch.from_numpy(x[b0:b0 + 8])
        r8, r16 = et.entropy_map(xt, 8).numpy(), et.entropy_map(xt, 16).numpy()
        c8, c16 = orc.entropy(x[b0:b0 + 8], 8), orc.entropy(x[b0:b0 + 8], 16)
        d8 = max(d8, float(np.abs(r8 - g8[b0:b0 + 8]).max())); d16 = max(d16, float(np.abs(r16 - g16[b0:b0 + 8]).max()))
        for i in range(r8.shape[0]):
            a = orc.router(r16[i:i + 1], r8[i:i + 1], 0.1, 0.8)
            d = sum(int((mk[g][b0 + i, 0] != a[g][0, 0]).sum()) for g in range(3))
            flips += d; imgs += d > 0
            c = orc.router(c16[i:i + 1], c8[i:i + 1], 0.1, 0.8)
            d = sum(int((c[g] != a[g]).sum()) for g in range(3))
            flips_c += d; imgs_c += d > 0
    print(f"{name}: GPU vs reference arithmetic: {flips} mask elements in {imgs}/{n} images (max |de8| {d8:.2e}, |de16| {d16:.2e}); "
          f"C oracle vs reference arithmetic: {flips_c} in {imgs_c}/{n}; cpu {time.time() - t0:.0f} s", flush=True)
