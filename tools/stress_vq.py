"""GPU stress: the bf16-filter VQ path against the plain-VALU restatement over many random shapes, scales and
codebooks (indices and z_q must be bit-identical).  Not a pytest (minutes); run by hand."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd.quantize import _vq_forward, prepare_codebook
if "--telemetry" in sys.argv:
    # margin telemetry of the candidate filter (cgic_vq_filter_probe_f32): worst observed |f - F| against the budget and the share
    # of the candidate margin the reference's winners used, over the stress families (tests/test_gpu_stress.py holds both under 0.5)
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
    import test_gpu_stress as tgs
    from oracle import cgic_oracle as orc
    orc.build()
    seed = int(sys.argv[sys.argv.index("--telemetry") + 1]) if len(sys.argv) > sys.argv.index("--telemetry") + 1 else 12
    for rep in range(3):
        for name, (z, cb) in tgs.telemetry_families(np.random.default_rng(seed + rep), hw=2048).items():
            print(rep, name, tgs.filter_margin_telemetry(orc, z, cb), flush=True)
    sys.exit(0)
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
t0 = time.time(); n = 0; nvec = 0
while time.time() - t0 < float(sys.argv[2] if len(sys.argv) > 2 else 60):
    B = int(rng.integers(1, 9)); h = int(rng.integers(1, 40)); w = int(rng.integers(1, 40))
    K = int(rng.choice([64, 128, 256, 512, 1024]))
    zs = float(10 ** rng.uniform(-3, 3)); cs = float(10 ** rng.uniform(-3, 3))
    kind = int(rng.integers(0, 5))
    g = torch.Generator().manual_seed(int(rng.integers(0, 2**31)))
    cb = torch.randn(K, 4, generator=g) * cs
    z = torch.randn(B, 4, h, w, generator=g) * zs
    if kind == 1:   # clustered codebook (many near-duplicates)
        cb = cb[torch.randint(0, 8, (K,), generator=g)] + torch.randn(K, 4, generator=g) * cs * 1e-6
    if kind == 2:   # latents exactly on codes, plus tiny noise
        z = cb[torch.randint(0, K, (B * h * w,), generator=g)].reshape(B, h, w, 4).permute(0, 3, 1, 2).contiguous() + torch.randn(B, 4, h, w, generator=g) * cs * 1e-7
    if kind == 3:   # quantised values: exact ties are common
        cb = torch.round(cb / cs * 2) * cs / 2; z = torch.round(z / zs * 2) * zs / 2
    if kind == 4:   # trained-VQGAN-like init
        cb = (torch.rand(K, 4, generator=g) * 2 - 1) / K
    if rng.integers(0, 3) == 0:   # exact duplicates of rows at random places (ties go to the lowest original index)
        m = int(rng.integers(1, max(2, K // 8)))
        cb[torch.randint(0, K, (m,), generator=g)] = cb[torch.randint(0, K, (m,), generator=g)]
    z, cb = z.cuda(), cb.cuda()
    a = _vq_forward(z, cb, 0.25, True, None, kernel="mfma"); b = _vq_forward(z, cb, 0.25, True, None, kernel="valu")
    # ... and through the prepared image (clusters of near-duplicate rows packed into tiles: the PERM kernels when there are any)
    p = _vq_forward(z, cb, 0.25, True, None, prepared=prepare_codebook(cb))
    if not (torch.equal(p[2], b[2]) and torch.equal(p[0], b[0])):
        bad = (p[2] != b[2]).nonzero().flatten()[:5].tolist()
        print("MISMATCH (prepared image)", dict(B=B, h=h, w=w, K=K, zs=zs, cs=cs, kind=kind), "first bad vectors", bad); sys.exit(1)
    if not (torch.equal(a[2], b[2]) and torch.equal(a[0], b[0])):
        bad = (a[2] != b[2]).nonzero().flatten()[:5].tolist()
        print("MISMATCH", dict(B=B, h=h, w=w, K=K, zs=zs, cs=cs, kind=kind), "first bad vectors", bad); sys.exit(1)
    la, lb = float(a[1]), float(b[1])
    if not (la == lb or abs(la - lb) <= 1e-6 * abs(lb)):
        print("LOSS MISMATCH", la, lb, dict(B=B, h=h, w=w, K=K, zs=zs, cs=cs, kind=kind)); sys.exit(1)
    n += 1; nvec += B * h * w
print(f"{n} random cases, {nvec} vectors: filter path == filter path through the prepared image == VALU path everywhere")
