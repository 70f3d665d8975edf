"""GPU stress: the router's threshold-band refinement -- stand-alone launch with and without the refinement queues, fused VQ + router
launch with and without the row bands' split (tiles of >= 1024 patches here) -- against the routing on the reference-arithmetic maps, over random batch sizes, image sizes, ratios and tie-heavy content
(masks must be identical).  Not a pytest (minutes); usage: python tools/stress_refine.py [seed] [seconds]"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
dev = torch.device("cuda", 0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
w = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
t0 = time.time(); n = npix = 0
sizes = [(64, 64), (128, 192), (256, 256), (512, 384), (768, 768), (768, 592), (16, 16), (48, 80), (768, 768), (1024, 512), (512, 1024), (640, 656)]
_lib.REFINE_SPLIT_MIN_PATCHES = 1024
while time.time() - t0 < budget:
    H, W = sizes[int(rng.integers(0, len(sizes)))]
    B = int(rng.integers(1, 65)) if H * W <= 65536 else int(rng.integers(1, 9)) if rng.integers(0, 3) else int(rng.integers(9, 33))
    fam = families(n=B, H=H, W=W, seed=int(rng.integers(0, 2**31)))
    name = ["noise8", "smooth8", "flat_edges", "blocky8"][int(rng.integers(0, 4))]
    x = fam[name]
    if rng.integers(0, 6) == 0 and H % 8 == 0 and W % 8 == 0:       # every 8x8 block one colour: hundreds of distinct grays of constant patches
        blk = rng.integers(0, 256, (B, 3, H // 8, W // 8)).astype(np.float32) / 255.0
        x = np.ascontiguousarray(np.repeat(np.repeat(blk, 8, axis=2), 8, axis=3))
    elif rng.integers(0, 4) == 0:                        # constant images / images made of two grays
        x = np.full_like(x, np.float32(rng.integers(0, 256) / 255.0)); x[:, :, : H // 2] = np.float32(rng.integers(0, 256) / 255.0)
    c = float(rng.choice([0.0, 0.1, 0.25, 0.3, 0.5])); m = float(rng.choice([0.0, 0.25, 0.4, 0.45, 0.7, 0.8]))
    if c + m > 1.0:
        m = round(1.0 - c, 6)
    xd = torch.from_numpy(x).to(dev)
    e8, e16 = cg.entropy_maps(xd)
    r8, r16 = cg.entropy_maps(xd, reference_order=True)
    router = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)
    want = router(r16, r8, want_gate=False)[0]
    outs = {}
    for q in (False, True):
        _lib.REFINE_QUEUES = q
        outs["router q=%d" % q] = router(e16, e8, want_gate=False, pixels=xd)[0]
    z = torch.from_numpy(rng.standard_normal((B, 4, H // 4, W // 4)).astype(np.float32)).to(dev)
    for q in (False, True):
        _lib.REFINE_QUEUES = q
        outs["fused split=%d" % q] = vq_forward_route(z, w, 0.25, True, e16, e8, c, m, per_image=True, pixels=xd)[3]
    for k, got in outs.items():
        if not all(torch.equal(a, b) for a, b in zip(got, want)):
            print("MISMATCH", k, dict(B=B, H=H, W=W, name=name, c=c, m=m), [int((a != b).sum()) for a, b in zip(got, want)])
            os.makedirs("gpurun_out", exist_ok=True)
            np.savez_compressed("gpurun_out/stress_refine_fail.npz", x=(x * 255.0).round().astype(np.uint8), c=c, m=m)       # (8-bit content: exact)
            sys.exit(1)
    n += 1; npix += B * H * W
print(f"{n} random batches, {npix / 1e6:.1f} MPixel: refined routing (in-workgroup, queues, fused, fused with split row bands) == routing on the reference-arithmetic maps everywhere")
