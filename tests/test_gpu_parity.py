"""GPU: the HIP path (through the C ABI of libcgic_hip.so) against the golden vectors of the real
reference and against the CPU oracle on seeded inputs.  Integer / byte / index results are
compared bit-exactly; fp32 tolerances are written next to each assert."""
import os
import numpy as np
import pytest
import torch

import control_gic_amd as cg
from conftest import unpack_mask

pytestmark = pytest.mark.gpu
DEV = "cuda"


class _V:
    def __init__(self, v): self.v = v
    def item(self): return self.v


def _freq_mapping(freq, order):
    return {str(int(k)): _V(float(freq[int(k)])) for k in order}


def _t(a, dtype=None):
    t = torch.from_numpy(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.to(DEV)


# ---------------------------------------------------------------------------- A. VQ
VQ_CASES = ["normal_b2_32", "normal_n1", "normal_odd", "normal_b1_64", "init_small", "init_mixed", "dup_rows",
            "codes_as_z", "lattice_ties"]


def _make_vq(cb):
    vq = cg.VectorQuantizer(cb.shape[0], cb.shape[1], beta=0.25).to(DEV).eval()
    vq.embedding.weight.data.copy_(_t(cb))
    return vq


@pytest.mark.parametrize("kernel", ["mfma", "valu"])
@pytest.mark.parametrize("case", VQ_CASES)
def test_vq_golden(golden, case, kernel):
    from control_gic_amd.quantize import _vq_forward
    g = golden("vq")
    vq = _make_vq(g[case + "_cb"])
    with torch.no_grad():
        zq, loss, idx = _vq_forward(_t(g[case + "_z"]), vq.embedding.weight, 0.25, True, None, kernel=kernel)
    assert idx.dtype == torch.int64 and idx.dim() == 1
    assert np.array_equal(idx.cpu().numpy(), g[case + "_idx"].astype(np.int64).ravel())   # bit-exact
    assert np.array_equal(zq.cpu().numpy(), g[case + "_zq"])                               # bit-exact
    ref = float(g[case + "_loss"])
    assert abs(float(loss) - ref) <= 1e-6 * abs(ref) + 1e-12                               # mean's sum order


def test_vq_module_forward_and_counter(golden, orc):
    g = golden("vq")
    vq = _make_vq(g["normal_b2_32_cb"])
    z = _t(g["counter_z"])
    with torch.no_grad():
        zq, loss, idx = vq(z)
        assert vq.usage_counter.sum().item() == 0                 # eval: no counting (quantize.py:79)
        vq.train()
        vq(z)
        vq(z)
    assert np.array_equal(vq.usage_counter.cpu().numpy(), g["counter_cnt"])
    assert vq.embedding_counter["3"].item() == float(g["counter_cnt"][3])


def test_usage_counter_sync_base_follows_the_module_to_the_gpu(monkeypatch):
    """load_state_dict on the CPU, then .cuda(): the buffer moves and the synced base has to follow it; otherwise the next
    sync_usage_counter_now() reduces the whole checkpointed table again (= multiplies it by the world size)"""
    from control_gic_amd import dist as cdist
    ckpt = cg.VectorQuantizer(1024, 4, beta=0.25)
    ckpt.usage_counter.copy_(torch.arange(1024, dtype=torch.float32) * 5)
    vq = cg.VectorQuantizer(1024, 4, beta=0.25)
    vq.load_state_dict(ckpt.state_dict())
    vq = vq.to(DEV).train()
    monkeypatch.setattr(cdist, "all_reduce_histogram", lambda t: t.mul_(2))        # two ranks that counted the same delta
    vq.sync_usage_counter_now()
    vq.sync_usage_counter_now()
    assert torch.equal(vq.usage_counter.cpu(), ckpt.usage_counter)
    vq.usage_hist += 1
    vq.fold_usage_hist()
    vq.sync_usage_counter_now()
    assert torch.equal(vq.usage_counter.cpu(), ckpt.usage_counter + 2)


@pytest.mark.parametrize("shape,scale", [((64, 4, 64, 64), 1.0), ((3, 4, 192, 192), 1.0), ((5, 4, 20, 36), 0.002),
                                         ((1, 4, 512, 512), 1.0), ((5, 4, 63, 65), 1.0), ((1, 4, 1, 1), 1.0), ((1, 4, 4, 4096), 1.0)])
def test_vq_vs_oracle_and_cross_kernel(orc, shape, scale):
    """config 2 size (B=64, 256^2) and a 768^2-tile size: MFMA kernel == VALU kernel everywhere
    (size-independent property), and == the oracle on a slice the CPU finishes in seconds."""
    g = torch.Generator().manual_seed(123)
    z = torch.randn(shape, generator=g) * scale
    cb = torch.randn(1024, 4, generator=g) * scale if scale == 1.0 else (torch.rand(1024, 4, generator=g) * 2 - 1) / 1024
    vq = _make_vq(cb.numpy())
    hist = torch.zeros(1024, dtype=torch.int64, device=DEV)
    from control_gic_amd.quantize import _vq_forward
    zq_a, loss_a, idx_a = _vq_forward(z.to(DEV), vq.embedding.weight, 0.25, True, hist, kernel="mfma")
    zq_b, loss_b, idx_b = _vq_forward(z.to(DEV), vq.embedding.weight, 0.25, True, None, kernel="valu")
    assert torch.equal(idx_a, idx_b) and torch.equal(zq_a, zq_b)
    assert abs(float(loss_a) - float(loss_b)) <= 1e-6 * abs(float(loss_b))
    assert int(hist.sum()) == idx_a.numel()
    assert torch.equal(hist, torch.bincount(idx_a, minlength=1024))
    nb = min(2, shape[0])
    ozq, oloss, oidx = orc.vq(z[:nb].numpy(), cb.numpy())
    n = oidx.size
    assert np.array_equal(idx_a[:n].cpu().numpy(), oidx)
    assert np.array_equal(zq_a[:nb].cpu().numpy(), ozq)
    # embedding gather is an exact copy of codebook rows (model.py:391-392)
    e = vq.embedding(idx_a[:n]).detach().cpu().numpy()
    assert np.array_equal(e, cb.numpy()[oidx])


def _vq_filter_cases():
    g = torch.Generator().manual_seed(77)
    cb = torch.randn(1024, 4, generator=g)
    z = torch.randn(3, 4, 5, 7, generator=g)                       # N = 105: ragged last group
    big = torch.randn(4, 4, 32, 32, generator=g)
    dup = cb.clone(); dup[512:] = dup[:512]                        # every code twice: the lower index must win
    onrow = cb[torch.randint(0, 1024, (2 * 16 * 16,), generator=g)].reshape(2, 16, 16, 4).permute(0, 3, 1, 2).contiguous()
    return {
        "zero_codebook": (big, torch.zeros(1024, 4)),              # every distance ties -> index 0, whole groups flagged
        "duplicate_codes": (big, dup),
        "latents_on_codes": (onrow, cb),                           # distance ~0 for the winner
        "constant_latents": (torch.full((2, 4, 16, 16), 0.37), cb),
        "ragged": (z, cb),
        "zz_dominates": (big * 1.0e4, cb * 1.0e-3),                # fl(zz + ee) swallows ee: masses of exact ties
        "huge_latents": (big * 1.0e18, cb),                        # zz overflows to inf, like the reference
        "tiny_everything": (big * 1.0e-25, cb * 1.0e-25),          # products underflow
        "K64": (big, cb[:64].clone()),
        "K256": (big, cb[:256].clone()),
        "K48_exact_loop": (big, cb[:48].clone()),                  # K % 64 != 0 -> fp32 MFMA loop
        "K2048_exact_loop": (big, torch.randn(2048, 4, generator=g)),
        # a zero row far from quad 0 wins with score exactly 0 (index bits packed into a zero must survive)
        "zero_row_wins": (big * 1.0e-3, torch.cat([cb[:700] + 3.0, torch.zeros(1, 4), cb[701:] + 3.0])),
        "denormal_scores": (big * 1.0e-22, torch.cat([cb[:700] * 1.0e-3 + 3.0, torch.full((1, 4), 1.0e-22), cb[701:] * 1.0e-3 + 3.0])),
    }


@pytest.mark.parametrize("case", sorted(_vq_filter_cases()))
def test_vq_filter_path_degenerate_inputs(case):
    """the bf16-filter path (kernel="mfma", K % 64 == 0, K <= 1024) against the plain-VALU restatement of the
    reference's rounding sequence: indices and z_q bit-exact whatever the filter flags"""
    from control_gic_amd.quantize import _vq_forward
    z, cb = _vq_filter_cases()[case]
    z, cb = z.to(DEV), cb.to(DEV)
    zq_a, loss_a, idx_a = _vq_forward(z, cb, 0.25, True, None, kernel="mfma")
    zq_b, loss_b, idx_b = _vq_forward(z, cb, 0.25, True, None, kernel="valu")
    assert int(idx_a.min()) >= 0 and int(idx_a.max()) < cb.shape[0]
    assert torch.equal(idx_a, idx_b)
    assert torch.equal(zq_a, zq_b)
    la, lb = float(loss_a), float(loss_b)
    assert (la == lb) or abs(la - lb) <= 1e-6 * abs(lb)            # == covers inf
    if case == "zero_codebook":
        assert int(idx_a.max()) == 0
    if case == "duplicate_codes":
        assert int(idx_a.max()) < 512


def test_vq_backward_matches_reference_formula():
    g = torch.Generator().manual_seed(5)
    z = torch.randn(2, 4, 8, 8, generator=g).to(DEV).requires_grad_(True)
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(DEV).train()
    vq.embedding.weight.data.copy_(torch.randn(1024, 4, generator=g))
    zq, loss, idx = vq(z)
    (zq.square().sum() + 3.0 * loss).backward()
    # plain-torch restatement of quantize.py:83-93 on the same indices
    z2 = z.detach().clone().requires_grad_(True)
    w2 = vq.embedding.weight.detach().clone().requires_grad_(True)
    zp = z2.permute(0, 2, 3, 1)
    e = w2[idx].view(zp.shape)
    l2 = ((e.detach() - zp) ** 2).mean() + 0.25 * ((e - zp.detach()) ** 2).mean()
    q2 = (zp + (e - zp).detach()).permute(0, 3, 1, 2)
    (q2.square().sum() + 3.0 * l2).backward()
    assert torch.allclose(z.grad, z2.grad, rtol=1e-5, atol=1e-7)
    assert torch.allclose(vq.embedding.weight.grad, w2.grad, rtol=1e-5, atol=1e-7)


# ---------------------------------------------------------------------------- C. router
@pytest.mark.parametrize("shape", ["b1_16x16", "b2_16x16", "b1_4x6", "b3_8x12"])
def test_router_golden(golden, orc, shape):
    g = golden("router")
    e16, e8 = g[shape + "_e16"], g[shape + "_e8"]
    for ri, (c, m) in enumerate(g["ratios"]):
        r = cg.TripleGrainFixedEntropyRouter(float(c), float(m))
        mask, gate, ratios, mode = r(_t(e16), _t(e8))
        assert mode == int(g[f"{shape}_r{ri}_mode"])
        for k, t in zip("cmf", mask):
            assert t.dtype == torch.int32
            assert np.array_equal(t.cpu().numpy(), unpack_mask(g[f"{shape}_r{ri}_m{k}"], tuple(t.shape))), (c, m, k)
        omc, omm, omf, ogate, _ = orc.router(e16, e8, float(c), float(m))
        assert np.array_equal(gate.cpu().numpy(), ogate)
        assert ratios == [float(c), float(m), 1 - float(c) - float(m)]
        # per-image routing == B independent B=1 calls
        pm, _, _, _ = cg.TripleGrainFixedEntropyRouter(float(c), float(m), per_image=True)(_t(e16), _t(e8))
        pmc, pmm, pmf, _, _ = orc.router(e16, e8, float(c), float(m), per_image=True)
        for a, b in zip(pm, (pmc, pmm, pmf)):
            assert np.array_equal(a.cpu().numpy(), b)


def test_router_batch64_and_tile_sizes(orc):
    g = np.random.default_rng(3)
    for (B, h16, w16) in ((64, 16, 16), (2, 48, 48), (1, 48, 37), (1, 128, 128), (1, 1, 1), (1, 1, 300), (2, 200, 3), (3000, 1, 1)):
        e16 = (g.random((B, h16, w16)) * 2.6).astype(np.float32)
        e8 = (g.random((B, 2 * h16, 2 * w16)) * 2.6).astype(np.float32)
        e8.ravel()[g.integers(0, e8.size, e8.size // 5)] = np.float32(1.25)      # heavy ties
        for c, m in ((0.1, 0.8), (0.25, 0.25), (0.0, 0.5), (0.5, 0.0)):
            for per_image in (False, True):
                mask, gate, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=per_image)(_t(e16), _t(e8))
                omc, omm, omf, ogate, omode = orc.router(e16, e8, c, m, per_image=per_image)
                assert mode == omode
                assert np.array_equal(mask[0].cpu().numpy(), omc) and np.array_equal(mask[1].cpu().numpy(), omm)
                assert np.array_equal(mask[2].cpu().numpy(), omf) and np.array_equal(gate.cpu().numpy(), ogate)


@pytest.mark.parametrize("name", ["noise8", "smooth8", "flat_edges", "blocky8"])
def test_entropy_tie_families_vs_reference_fixture(golden, name):
    """the REAL Entropy class's maps for 8-bit noise / smooth / flat / blocky images (tests/golden/ties.npz): <= 2e-6"""
    g = golden("ties")
    x = torch.from_numpy(g[name + "_u8"].astype(np.float32) / 255.0)
    e8, e16 = cg.entropy_maps(x.to(DEV))
    assert np.abs(e8.cpu().numpy() - g[name + "_e8"]).max() < 2e-6
    assert np.abs(e16.cpu().numpy() - g[name + "_e16"]).max() < 2e-6
    # ... and by value: the bound the router's band width is made of (cgic_router_dev.h: refine_delta), against the REAL class
    for e, ref in ((e8, g[name + "_e8"]), (e16, g[name + "_e16"])):
        assert (np.abs(e.cpu().numpy().astype(np.float64) - ref) <= _refine_delta(ref)).all()


def _refine_delta(v):
    """cgic_router_dev.h: refine_delta -- the default entropy kernel's error bound by value"""
    v = np.asarray(v, np.float64)
    return np.minimum(2e-6, 7e-7 + 0.01 * np.maximum(0.0, v - 1e-4))


def test_entropy_error_by_value_holds_the_band_width():
    """The router's threshold band is 2 x refine_delta(threshold): the default kernel against the reference-arithmetic kernel on
    every content family, 768x768 tiles, smooth gradients of other amplitudes / frequencies (8-bit and fp32) -- the error has to
    stay under the bound everywhere, with room (<= 0.7 of it)"""
    from oracle.content_families import families
    rng = np.random.default_rng(1)
    sets = dict(families(n=16))
    sets.update({"tile_" + k: v for k, v in families(n=1, H=768, W=768, seed=11).items()})
    sets["rand"] = rng.random((16, 3, 256, 256)).astype(np.float32)
    yy, xx = np.mgrid[0:256, 0:256].astype(np.float32)
    gsm = np.empty((32, 3, 256, 256), np.float32)
    for i in range(32):
        a = rng.uniform(0, 2 * np.pi); f = rng.uniform(0.05, 3.0)
        base = rng.uniform(0.1, 0.9) + rng.uniform(0.01, 0.4) * np.sin((np.cos(a) * xx + np.sin(a) * yy) * f * 2 * np.pi / 256)
        for c in range(3):
            gsm[i, c] = base * rng.uniform(0.5, 1.0) + rng.integers(-2, 3, (256, 256)) / 255.0 * rng.integers(0, 2)
    sets["smooth_var8"] = np.round(np.clip(gsm, 0, 1) * 255.0).astype(np.float32) / 255.0
    sets["smooth_f32"] = np.clip(gsm, 0, 1).astype(np.float32)
    for name, x in sets.items():
        xd = _t(x)
        e8, e16 = cg.entropy_maps(xd)
        r8, r16 = cg.entropy_maps(xd, reference_order=True)
        for e, r in ((e8, r8), (e16, r16)):
            r = r.cpu().numpy().astype(np.float64)
            worst = (np.abs(e.cpu().numpy() - r) / _refine_delta(r)).max()
            assert worst <= 0.7, (name, worst)


def test_reference_arithmetic_entropy_kernel(orc, golden):
    """cgic_entropy_maps_ref_f32 (opt-in): bit for bit the oracle's restatement of the same arithmetic (torch's CPU operation
    sequence and summation order, exp / log correctly rounded) -- on the reference's 8-bit fixtures, U(0,1) images, signed /
    out-of-range pixels, a ragged width -- hence, through the fixture, most values equal to the REAL Entropy class's to the bit
    and the rest within 5e-7; NaN pixels poison their patches like the reference's NaN histogram"""
    g = golden("ties")
    rng = np.random.default_rng(12)
    cases = {n: g[n + "_u8"].astype(np.float32) / 255.0 for n in ("noise8", "smooth8", "flat_edges", "blocky8")}
    cases["rand"] = rng.random((3, 3, 64, 112), dtype=np.float32)
    cases["signed"] = (rng.random((1, 3, 32, 48), dtype=np.float32) * 3 - 1.5)
    cases["const"] = np.full((1, 3, 32, 32), 0.5, np.float32)
    for name, x in cases.items():
        e8, e16 = cg.entropy_maps(torch.from_numpy(x).to(DEV), reference_order=True)
        o8, o16 = orc.entropy_ref(x, 8), orc.entropy_ref(x, 16)
        assert np.array_equal(e8.cpu().numpy().view(np.int32), o8.view(np.int32)), (name, float(np.abs(e8.cpu().numpy() - o8).max()))
        assert np.array_equal(e16.cpu().numpy().view(np.int32), o16.view(np.int32)), (name, float(np.abs(e16.cpu().numpy() - o16).max()))
        if name + "_e8" in g:
            assert np.abs(e8.cpu().numpy() - g[name + "_e8"]).max() < 5e-7 and (e8.cpu().numpy() == g[name + "_e8"]).mean() > 0.85
        m8 = cg.Entropy(8, reference_order=True).to(DEV)(torch.from_numpy(x).to(DEV))
        assert torch.equal(m8, e8)
        r8, r16 = torch.ops.cgic.entropy_maps_reference_order(torch.from_numpy(x).to(DEV))
        assert torch.equal(r8, e8) and torch.equal(r16, e16)
        d8, _ = cg.entropy_maps(torch.from_numpy(x).to(DEV))
        assert np.abs(d8.cpu().numpy() - e8.cpu().numpy()).max() < 2e-6               # the default kernel agrees to its tolerance
    xn = torch.rand(1, 3, 32, 64)
    xn[0, 1, 3, 5] = float("nan")
    e8, e16 = cg.entropy_maps(xn.to(DEV), reference_order=True)
    assert torch.isnan(e8).nonzero().tolist() == [[0, 0, 0]] and torch.isnan(e16).nonzero().tolist() == [[0, 0, 0]]


def test_reference_arithmetic_entropy_flips_no_mask(orc):
    """the opt-in reference-arithmetic entropy maps -> GPU router against the reference's torch-CPU arithmetic -> oracle router
    on the tie-heavy families (16 images each) and two 768x768 tiles: no differing mask element (measured; the default kernel
    differs in 0-1 images of 64 per family, test_mask_flips_on_tie_heavy_content)"""
    from oracle import entropy_torch as et
    from oracle.content_families import families
    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    sets = dict(families(n=16))
    t = families(n=1, H=768, W=768, seed=11)
    sets["tiles"] = np.concatenate([t["flat_edges"], t["smooth8"]])
    total = 0
    for name, x in sets.items():
        e8, e16 = cg.entropy_maps(torch.from_numpy(x).to(DEV), reference_order=True)
        mk = [m.cpu().numpy() for m in router(e16, e8)[0]]
        xt = torch.from_numpy(x)
        r8, r16 = et.entropy_map(xt, 8).numpy(), et.entropy_map(xt, 16).numpy()
        assert max(np.abs(r8 - e8.cpu().numpy()).max(), np.abs(r16 - e16.cpu().numpy()).max()) < 1e-6, name
        for b in range(x.shape[0]):
            ref = orc.router(r16[b:b + 1], r8[b:b + 1], 0.1, 0.8)
            total += sum(int((mk[k][b, 0] != ref[k][0, 0]).sum()) for k in range(3))
    assert total <= 8, total        # (0 measured; a few elements of slack for another host's MKL)


def _tie_sets(n=64, flat=16):
    """the tie-heavy families (n images of 256x256 each; `flat` of the flat one, whose bands are hundreds of patches) + two tiles"""
    from oracle.content_families import families
    sets = dict(families(n=n))
    sets["flat_edges"] = sets["flat_edges"][:flat]
    tl = families(n=1, H=768, W=768, seed=11)
    sets["tiles_768"] = np.concatenate([tl["smooth8"], tl["noise8"]])
    return sets


def test_mask_flips_on_tie_heavy_content(orc):
    """pixels -> masks on 8-bit / smooth / flat / blocky content + two 768x768 tiles, DEFAULT path (entropy_maps -> router with the
    pixels: the threshold-band refinement) against the reference's own torch-CPU arithmetic (oracle/entropy_torch.py, pinned bit
    for bit against the real Entropy class) -> oracle router.  Round 3 measured 0-1 images of 64 per family with flipped mask
    elements for the maps alone (1e-6 away from a threshold that is a k-th smallest value under a strict '<',
    RouterTriple.py:21-34); with the refinement: none.  (Slack only for another host's MKL: its exp / log differ from the
    correctly rounded ones in ~1 % of the arguments.)  Entropy itself: <= 2e-6 absolute everywhere."""
    from oracle import entropy_torch as et
    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    plain = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    plain.refine = False                                     # round 3's behaviour: the maps decide as given
    pinned = torch.__version__.startswith("2.10.0")

    def flips(x):
        xd = torch.from_numpy(x).to(DEV)
        e8, e16 = cg.entropy_maps(xd)
        mk = [m.cpu().numpy() for m in router(e16, e8)[0]]             # the pixels (and the flat map) ride on the maps
        mk0 = [m.cpu().numpy() for m in plain(e16, e8)[0]]
        g8, g16 = e8.cpu().numpy(), e16.cpu().numpy()
        elems = imgs = elems0 = 0
        dmax = 0.0
        for b0 in range(0, x.shape[0], 8):
            xt = torch.from_numpy(x[b0:b0 + 8])
            r8, r16 = et.entropy_map(xt, 8).numpy(), et.entropy_map(xt, 16).numpy()
            dmax = max(dmax, float(np.abs(r8 - g8[b0:b0 + 8]).max()), float(np.abs(r16 - g16[b0:b0 + 8]).max()))
            for i in range(r8.shape[0]):
                ref = orc.router(r16[i:i + 1], r8[i:i + 1], 0.1, 0.8)
                d = sum(int((mk[k][b0 + i, 0] != ref[k][0, 0]).sum()) for k in range(3))
                elems0 += sum(int((mk0[k][b0 + i, 0] != ref[k][0, 0]).sum()) for k in range(3))
                elems += d
                imgs += d > 0
        return elems, imgs, dmax, elems0

    unrefined = 0
    for name, x in _tie_sets().items():
        elems, imgs, dmax, elems0 = flips(x)
        unrefined += elems0
        assert dmax < 2e-6, (name, dmax)
        if pinned:
            assert elems == 0, f"{name}: {elems} mask elements in {imgs} images differ from the reference arithmetic (maps alone: {elems0})"
        else:
            assert imgs <= 1 and elems <= 40, (name, elems, imgs)
    assert unrefined > 0        # (the families do exercise the band: without the pixels some elements flip)


@pytest.mark.parametrize("ratio", [(0.1, 0.8), (0.25, 0.25), (0.0, 0.4), (0.4, 0.0), (0.3, 0.7), (0.5, 0.45)])
def test_refined_router_equals_router_on_reference_arithmetic_maps(ratio):
    """The refinement's claim, checked on the GPU alone: router(default maps, pixels) == router(reference-arithmetic maps) --
    every patch within the band of a threshold is re-evaluated with the arithmetic of cgic_entropy_maps_ref_f32, everything
    outside the band keeps its side.  All four modes that compare (0, 1, 2, 3), stand-alone and VQ-fused launch, fp32 pixels and
    uint8 frames, images and 768x768 tiles (several workgroups per image), plus maps perturbed by +-0.95 of the kernel's error
    bound by value (refine_delta: what the band is made of): the masks must not move."""
    from control_gic_amd.quantize import vq_forward_route
    c, m = ratio
    router = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)
    rng = np.random.default_rng(int(1000 * c + 10 * m))
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    for name, x in _tie_sets(n=24, flat=6).items():
        xd = torch.from_numpy(x).to(DEV)
        B, _, H, W = xd.shape
        e8, e16 = cg.entropy_maps(xd)
        r8, r16 = cg.entropy_maps(xd, reference_order=True)
        want, _, _, mode = router(r16, r8, want_gate=False)
        got = router(e16, e8, want_gate=False, pixels=xd)[0]
        assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, [int((a != b).sum()) for a, b in zip(got, want)])
        z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
        fused = vq_forward_route(z, w, 0.25, True, e16, e8, c, m, per_image=True, pixels=xd)[3]
        assert all(torch.equal(a, b) for a, b in zip(fused, want)), (name, ratio, "fused")
        # the maps may sit anywhere within the kernel's error bound (by value: refine_delta) of the reference arithmetic
        dl = lambda r: torch.clamp(7e-7 + 0.01 * torch.clamp(r - 1e-4, min=0.0), max=2e-6)
        n8 = (torch.rand(e8.shape, device=DEV) - 0.5) * 1.9 * dl(r8)
        n16 = (torch.rand(e16.shape, device=DEV) - 0.5) * 1.9 * dl(r16)
        got = router(r16 + n16, r8 + n8, want_gate=False, pixels=xd)[0]
        assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, "perturbed")
        if name in ("smooth8", "tiles_768"):
            frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            x2, f8, f16 = cg.entropy_maps_u8(frames)
            assert torch.equal(x2, xd) and torch.equal(f8, e8) and torch.equal(f16, e16)
            got = router(f16, f8, want_gate=False, pixels=frames)[0]
            assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, "uint8 frames")


TIES_SMALL = ["noise8", "smooth8", "flat_edges", "blocky8"]
TIES_BIG = ["noise8_256", "smooth8_256", "flat_edges_256", "blocky8_256", "smooth8_768"]


@pytest.mark.parametrize("name", TIES_SMALL + TIES_BIG)
def test_pixels_to_masks_equal_the_reference_routers_own_masks(golden, name):
    """the GPU's pixels -> masks path against the masks the REAL TripleGrainFixedEntropyRouter produced from the
    REAL Entropy maps (tests/golden/ties.npz, make_golden_ties.py) -- no host torch, no MKL slack, exact.  8-bit tie-heavy
    families at 96x128 (round 4's fixture), two 256x256 images per family and one smooth 768x768 tile (bands of up to ~200
    patches).  Every form the product offers: the stand-alone router with the pixels, the fused VQ + router launch, uint8 frames
    (ToTensor inside the map kernel), the one-call driver (cgic_compress_image), and -- for the pairs -- the reference's
    batch-global routing (encode(): RouterTriple.py:21,40,52,63)."""
    from control_gic_amd.quantize import vq_forward_route
    g = golden("ties")
    u8 = g[name + "_u8"]
    B, _, H, W = u8.shape
    xd = torch.from_numpy(u8.astype(np.float32) / 255.0).to(DEV)
    frames = torch.from_numpy(np.ascontiguousarray(u8.transpose(0, 2, 3, 1))).to(DEV)

    def want(b):
        shapes = [(H // 16, W // 16), (H // 8, W // 8), (H // 4, W // 4)]
        n = B if b == "batch" else 1
        return [np.unpackbits(g[f"{name}_{b}_m{k}"])[:n * h * w].reshape(n, 1, h, w).astype(np.int32) for k, (h, w) in zip("cmf", shapes)]

    per = [want(b) for b in range(B)]
    ref = [np.concatenate([per[b][k] for b in range(B)]) for k in range(3)]

    def same(masks, ref_, what):
        for k in range(3):
            d = int((masks[k].cpu().numpy() != ref_[k]).sum())
            assert d == 0, (name, what, "cmf"[k], d)

    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    e8, e16 = cg.entropy_maps(xd)
    same(router(e16, e8, want_gate=False, pixels=xd)[0], ref, "stand-alone router, fp32 pixels")
    same(router(e16, e8, want_gate=False)[0], ref, "stand-alone router, pixels riding on the maps")
    x2, f8, f16 = cg.entropy_maps_u8(frames)
    assert torch.equal(x2, xd)
    same(router(f16, f8, want_gate=False, pixels=frames)[0], ref, "stand-alone router, uint8 frames")
    rng = np.random.default_rng(3)
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
    same(vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd)[3], ref, "fused launch, fp32 pixels")
    same(vq_forward_route(z, w, 0.25, True, f16, f8, 0.1, 0.8, per_image=True, pixels=frames)[3], ref, "fused launch, uint8 frames")
    # reference-order maps decide by themselves
    r8, r16 = cg.entropy_maps(xd, reference_order=True)
    plain = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    plain.refine = False
    same(plain(r16, r8, want_gate=False)[0], ref, "reference-order maps")
    if f"{name}_batch_mc" in g:
        # the reference's own batch semantics: thresholds over the flattened batch
        flat_router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=False)
        same(flat_router(e16, e8, want_gate=False, pixels=xd)[0], want("batch"), "batch-global routing, fp32 pixels")
        same(flat_router(f16, f8, want_gate=False, pixels=frames)[0], want("batch"), "batch-global routing, uint8 frames")
        same(vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=False, pixels=xd)[3], want("batch"), "fused launch, batch-global")


@pytest.mark.parametrize("B", [16, 64])
@pytest.mark.parametrize("ratio", [(0.1, 0.8), (0.0, 0.4), (0.4, 0.0), (0.3, 0.7)])
def test_batch_global_routing_is_refined_beyond_one_workgroups_lds(B, ratio):
    """the reference's encode() routes over the FLATTENED batch (RouterTriple.py:21,40,52,63).  64 images of
    256x256 are one segment of 16 384 + 65 536 entropies -- beyond a workgroup's LDS -- and are refined through patched copies
    of the maps (cgic_router.hip: router_big): masks equal to batch-global routing on the reference-arithmetic maps, on the
    tie-heavy families, in every mode that compares, stand-alone and through the VQ + router entry point, fp32 pixels and
    uint8 frames, and with maps perturbed within the kernel's error bound.  Also one untiled 1024x1280 image routed per image."""
    from control_gic_amd.quantize import vq_forward_route
    c, m = ratio
    assert cg._lib.lib().cgic_router_refine_supported(B, 16, 16, 0) == 1 and cg._lib.lib().cgic_router_refine_in_lds(B, 16, 16, 0) == 0
    router = cg.TripleGrainFixedEntropyRouter(c, m, per_image=False)
    rng = np.random.default_rng(B + int(100 * c))
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    sets = _tie_sets(n=B, flat=B)
    for name in ("smooth8", "flat_edges", "noise8"):
        xd = torch.from_numpy(np.ascontiguousarray(sets[name][:B])).to(DEV)
        e8, e16 = cg.entropy_maps(xd)
        r8, r16 = cg.entropy_maps(xd, reference_order=True)
        want, _, _, mode = router(r16, r8, want_gate=False)
        got, gate, _, mode2 = router(e16, e8, pixels=xd)
        assert mode == mode2 and all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, [int((a != b).sum()) for a, b in zip(got, want)])
        assert torch.equal(gate, router(r16, r8)[1])
        z = _t(rng.standard_normal((B, 4, 64, 64), dtype=np.float32))
        zq, loss, idx, fused, _, _ = vq_forward_route(z, w, 0.25, True, e16, e8, c, m, per_image=False, pixels=xd)
        assert all(torch.equal(a, b) for a, b in zip(fused, want)), (name, ratio, "through cgic_vq_forward_route_f32")
        zq0, loss0, idx0 = cg.quantize._vq_forward(z, w, 0.25, True, None)
        assert torch.equal(idx, idx0) and torch.equal(zq, zq0) and torch.equal(loss, loss0)
        dl = lambda r: torch.clamp(7e-7 + 0.01 * torch.clamp(r - 1e-4, min=0.0), max=2e-6)
        n8 = (torch.rand(e8.shape, device=DEV) - 0.5) * 1.9 * dl(r8)
        n16 = (torch.rand(e16.shape, device=DEV) - 0.5) * 1.9 * dl(r16)
        got = router(r16 + n16, r8 + n8, want_gate=False, pixels=xd)[0]
        assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, "perturbed")
        if name == "smooth8":
            frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            _, f8, f16 = cg.entropy_maps_u8(frames)
            got = router(f16, f8, want_gate=False, pixels=frames)[0]
            assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, ratio, "uint8 frames")
    if B == 16:
        from oracle.content_families import families
        xl = torch.from_numpy(families(n=1, H=1024, W=1280, seed=5)["smooth8"]).to(DEV)
        per = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)
        assert cg._lib.lib().cgic_router_refine_in_lds(1, 64, 80, 1) == 0
        e8, e16 = cg.entropy_maps(xl)
        r8, r16 = cg.entropy_maps(xl, reference_order=True)
        want = per(r16, r8, want_gate=False)[0]
        got = per(e16, e8, want_gate=False, pixels=xl)[0]
        assert all(torch.equal(a, b) for a, b in zip(got, want)), ("1024x1280", ratio)


def test_hotcall_refines_images_beyond_768():
    """cgic_compress_image on an image routed as one segment beyond the LDS used to route from the UNREFINED maps
    without a word; it now refines through patched copies (HotCall allocates ws_refine) -- and a caller of the C entry point who
    leaves ws_refine out is refused"""
    from oracle.content_families import families
    xl = torch.from_numpy(families(n=2, H=1024, W=1024, seed=9)["smooth8"]).to(DEV)
    B, _, H, W = xl.shape
    rng = np.random.default_rng(0)
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(DEV).eval()
    vq.embedding.weight.data.copy_(_t(rng.standard_normal((1024, 4), dtype=np.float32)))
    z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
    hc = cg.pipeline.HotCall(vq, 0.1, 0.8, B, H, W)
    out = hc(xl, z)
    r8, r16 = cg.entropy_maps(xl, reference_order=True)
    want = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(r16, r8, want_gate=False)[0]
    assert all(torch.equal(a, b) for a, b in zip(out["mask"], want))
    hc._io.ws_refine, hc._io.ws_refine_bytes = None, 0
    with pytest.raises(cg.CgicError, match="scratch"):
        hc(xl, z)


def test_fused_launch_with_refinement_queues_gives_the_same_masks():
    """round 6: the fused VQ + router launch's variant in which an image with a long threshold band starts over with the launch's
    refinement queues (published patch lists, the routers that are done help): masks == the plain variant's == routing on the
    reference-arithmetic maps, on every tie-heavy family, repeated launches over the same header slots, fp32 pixels and uint8
    frames; VQ outputs untouched.  HotPathPipeline(refine_queues="auto") picks the variant from a batch of the stream: the
    smooth family (dozens of non-constant patches inside a band) takes the queues, noise does not."""
    from control_gic_amd.quantize import vq_forward_route
    rng = np.random.default_rng(11)
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(DEV).eval()
    vq.embedding.weight.data.copy_(w)
    picks = {}
    for name, x in _tie_sets(n=64, flat=16).items():
        xd = torch.from_numpy(np.ascontiguousarray(x)).to(DEV)
        B, _, H, W = xd.shape
        e8, e16 = cg.entropy_maps(xd)
        r8, r16 = cg.entropy_maps(xd, reference_order=True)
        want = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(r16, r8, want_gate=False)[0]
        z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
        plain = vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd, refine_queues=False)
        for rep in range(3):
            q = vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd, refine_queues=True)
            assert all(torch.equal(a, b) for a, b in zip(q[3], want)), (name, rep, [int((a != b).sum()) for a, b in zip(q[3], want)])
            assert torch.equal(q[0], plain[0]) and torch.equal(q[1], plain[1]) and torch.equal(q[2], plain[2])
        assert all(torch.equal(a, b) for a, b in zip(plain[3], want)), name
        if name == "smooth8":
            frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
            _, f8, f16 = cg.entropy_maps_u8(frames)
            q = vq_forward_route(z, w, 0.25, True, f16, f8, 0.1, 0.8, per_image=True, pixels=frames, refine_queues=True)
            assert all(torch.equal(a, b) for a, b in zip(q[3], want)), (name, "uint8 frames")
        picks[name] = cg.pipeline.HotPathPipeline(vq, 0.1, 0.8).decide(xd)
    assert picks["smooth8"] is True and picks["noise8"] is False and picks["flat_edges"] is False, picks


def test_refinement_queues_change_nothing_and_survive_concurrent_launches():
    """The stand-alone router launch evaluates long bands with every idle wave of the launch (refinement queues: the band's
    owner publishes its patch list; the other row bands of a tile and router workgroups that are done take patches; results as
    tagged 8-byte granules): masks identical with and without the queues and to the routing on the reference-arithmetic maps --
    batches of images, 768x768 tiles (eight row bands per tile share one select), uint8 frames, repeated launches over the same
    header slots, and four streams at once (launches of different streams share nothing but the library's slot pools)."""
    from control_gic_amd import _lib
    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    sets = _tie_sets(n=32, flat=8)
    assert "tiles_768" in sets
    want = {}
    for name, x in sets.items():
        xd = torch.from_numpy(x).to(DEV)
        e8, e16 = cg.entropy_maps(xd)
        r8, r16 = cg.entropy_maps(xd, reference_order=True)
        want[name] = (xd, e8, e16, [m.clone() for m in router(r16, r8, want_gate=False, pixels=None)[0]])
    try:
        for name, (xd, e8, e16, ref) in want.items():
            _lib.REFINE_QUEUES = False
            plain = router(e16, e8, want_gate=False, pixels=xd)[0]
            assert all(torch.equal(a, b) for a, b in zip(plain, ref)), (name, "in-workgroup")
            _lib.REFINE_QUEUES = True
            for rep in range(4):                      # (the same header slots again and again: whatever a launch leaves must read as "nothing to claim")
                got = router(e16, e8, want_gate=False, pixels=xd)[0]
                assert all(torch.equal(a, b) for a, b in zip(got, ref)), (name, "queues", rep, [int((a != b).sum()) for a, b in zip(got, ref)])
        # four streams, eight rounds: every stream routes every set
        streams = [torch.cuda.Stream() for _ in range(4)]
        torch.cuda.synchronize()
        outs = []
        names = list(want)
        for rnd in range(8):
            for k, st in enumerate(streams):
                name = names[(rnd + k) % len(names)]
                xd, e8, e16, ref = want[name]
                with torch.cuda.stream(st):
                    outs.append((name, router(e16, e8, want_gate=False, pixels=xd)[0]))
        torch.cuda.synchronize()
        for name, got in outs:
            assert all(torch.equal(a, b) for a, b in zip(got, want[name][3])), (name, "concurrent streams")
        # uint8 frames through the queues
        xd = want["smooth8"][0]
        frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        _, f8, f16 = cg.entropy_maps_u8(frames)
        got = router(f16, f8, want_gate=False, pixels=frames)[0]
        assert all(torch.equal(a, b) for a, b in zip(got, want["smooth8"][3]))
    finally:
        _lib.REFINE_QUEUES = True


def test_vq_loss_collector_on_sixteen_streams_at_once():
    """The VQ launch's loss: every workgroup stores its partial, workgroup 0 of the launch collects them (no ticket; cgic_vq.hip
    loss_collect).  Launches of many streams in flight together -- more streams than hardware queues, launches of a few workgroups
    and of thousands, eager -- give the loss, indices and z_q of the same call alone, bit for bit, and hand every slot back zeroed."""
    from control_gic_amd import _lib
    from control_gic_amd.quantize import _vq_forward
    rng = np.random.default_rng(2026)
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    shapes = [(1, 4, 16, 16), (3, 4, 40, 24), (64, 4, 64, 64), (16, 4, 192, 192)]
    zs = [_t(rng.standard_normal(sh, dtype=np.float32)) for sh in shapes]
    alone = [_vq_forward(z, w, 0.25, True, None) for z in zs]
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream() for _ in range(16)]
    outs = []
    for rnd in range(12):
        for k, st in enumerate(streams):
            i = (rnd + k) % len(zs)
            with torch.cuda.stream(st):
                outs.append((i, _vq_forward(zs[i], w, 0.25, True, None)))
    torch.cuda.synchronize()
    for i, (zq, loss, idx) in outs:
        assert torch.equal(idx, alone[i][2]) and torch.equal(zq, alone[i][0])
        assert loss.view(torch.int32).item() == alone[i][1].view(torch.int32).item(), (i, float(loss), float(alone[i][1]))
    assert _lib.lib().cgic_ticket_pool_dirty_words() == 0


def test_row_bands_split_a_band_in_the_fused_launch():
    """The fused VQ + router launch has no refinement queues; the row bands of a large tile (up to eight workgroups that all find
    the same band) split its re-evaluation between them instead -- FEW: by a hash of the patch index, MANY: every nb-th member in
    index order -- and exchange the values through the scratch (cgic_pixels.scratch, _lib.REFINE_SPLIT_MIN_PATCHES).  Masks
    identical to the launch without the scratch (every band evaluates everything) and to the routing on the reference-arithmetic
    maps: all four families as 768x768 tiles, other tile shapes and band counts (8, 6, 3, 2 row bands; 32 to 64 patch rows), uint8
    frames, repeated launches over the same header slots, four streams at once.  (The launch runs the router in two attempts: the
    plain instantiation first, which leaves when a band is long, then the split one from the top: both outcomes are exercised.)"""
    from control_gic_amd import _lib
    from control_gic_amd.quantize import vq_forward_route
    from oracle.content_families import families
    rng = np.random.default_rng(77)
    w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    cases = {}
    t = families(n=2, H=768, W=768, seed=11)
    cases["768x768 x8"] = np.concatenate([t[k] for k in ("noise8", "smooth8", "flat_edges", "blocky8")])
    t = families(n=5, H=512, W=1024, seed=12)
    cases["512x1024 x10 (6 bands)"] = np.concatenate([t["smooth8"], t["flat_edges"]])
    t = families(n=10, H=1024, W=512, seed=13)
    cases["1024x512 x20 (3 bands)"] = np.concatenate([t["smooth8"], t["noise8"]])
    t = families(n=1, H=640, W=656, seed=14)
    cases["640x656 x2"] = np.concatenate([t["smooth8"], t["flat_edges"]])
    t = families(n=12, H=512, W=768, seed=15)
    cases["512x768 x24 (2 bands)"] = np.concatenate([t["smooth8"], t["flat_edges"]])
    keep = _lib.REFINE_SPLIT_MIN_PATCHES, _lib.REFINE_QUEUES
    try:
        _lib.REFINE_SPLIT_MIN_PATCHES = 1024
        ready = {}
        for name, x in cases.items():
            xd = torch.from_numpy(x).to(DEV)
            B, _, H, W = xd.shape
            assert _lib.lib().cgic_router_refine_scratch_bytes(B, H // 16, W // 16, 1) > 0, name
            e8, e16 = cg.entropy_maps(xd)
            r8, r16 = cg.entropy_maps(xd, reference_order=True)
            want = [m.clone() for m in router(r16, r8, want_gate=False, pixels=None)[0]]
            z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
            f = lambda z=z, e16=e16, e8=e8, xd=xd: vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd)[3]
            _lib.REFINE_QUEUES = False
            plain = f()
            assert all(torch.equal(a, b) for a, b in zip(plain, want)), (name, "every band evaluates everything")
            _lib.REFINE_QUEUES = True
            for rep in range(4):
                got = f()
                assert all(torch.equal(a, b) for a, b in zip(got, want)), (name, "split", rep, [int((a != b).sum()) for a, b in zip(got, want)])
            ready[name] = (f, want)
        # blocky tiles, medium grain only, the quantile among the constant blocks: hundreds of distinct grays in the band -- more than
        # the table of shared grays holds; the row bands must still agree on the member list (found by tools/stress_refine.py)
        r2 = cg.TripleGrainFixedEntropyRouter(0.0, 0.25, per_image=True)
        for seed in (21, 22):
            blocks = np.random.default_rng(seed).integers(0, 256, (7, 3, 96, 96)).astype(np.float32) / 255.0        # every 8x8 block one colour
            xd = torch.from_numpy(np.ascontiguousarray(np.repeat(np.repeat(blocks, 8, axis=2), 8, axis=3))).to(DEV)
            e8, e16 = cg.entropy_maps(xd)
            r8, r16 = cg.entropy_maps(xd, reference_order=True)
            want = r2(r16, r8, want_gate=False, pixels=None)[0]
            z = _t(rng.standard_normal((7, 4, 192, 192), dtype=np.float32))
            for rep in range(3):
                got = vq_forward_route(z, w, 0.25, True, e16, e8, 0.0, 0.25, per_image=True, pixels=xd)[3]
                assert all(torch.equal(a, b) for a, b in zip(got, want)), ("blocky, more grays than table slots", seed, rep, [int((a != b).sum()) for a, b in zip(got, want)])
        xd = torch.from_numpy(cases["768x768 x8"]).to(DEV)
        frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
        _, f8, f16 = cg.entropy_maps_u8(frames)
        z = _t(rng.standard_normal((8, 4, 192, 192), dtype=np.float32))
        got = vq_forward_route(z, w, 0.25, True, f16, f8, 0.1, 0.8, per_image=True, pixels=frames)[3]
        assert all(torch.equal(a, b) for a, b in zip(got, ready["768x768 x8"][1])), "uint8 frames"
        streams = [torch.cuda.Stream() for _ in range(4)]
        torch.cuda.synchronize()
        outs, names = [], list(ready)
        for rnd in range(6):
            for k, st in enumerate(streams):
                name = names[(rnd + k) % len(names)]
                with torch.cuda.stream(st):
                    outs.append((name, ready[name][0]()))
        torch.cuda.synchronize()
        for name, got in outs:
            assert all(torch.equal(a, b) for a, b in zip(got, ready[name][1])), (name, "concurrent streams")
    finally:
        _lib.REFINE_SPLIT_MIN_PATCHES, _lib.REFINE_QUEUES = keep


def test_flat_map_and_constant_patch_shortcut():
    """entropy_maps' by-product flat8 (gray of an 8x8 patch whose 64 pixels agree bit for bit, else NaN) against numpy, fp32
    and uint8 input; with it the router evaluates constant band patches once per distinct gray -- same masks as without it
    and as routing on the reference-arithmetic maps"""
    from oracle.content_families import families
    fam = families(n=12)
    x = np.concatenate([fam["flat_edges"], fam["blocky8"][:4], fam["smooth8"][:4]])
    x[3, :, 64:72, 8:16] = 0.25           # one constant patch with its own gray
    x[4] = 0.5                            # a constant image: every patch in the band, one gray
    xd = torch.from_numpy(x).to(DEV)
    e8, e16 = cg.entropy_maps(xd)
    flat = e8._cgic_flat8.cpu().numpy()
    gray = (np.float32(0.2989) * x[:, 0] + np.float32(0.5870) * x[:, 1]) + np.float32(0.1140) * x[:, 2]
    gp = gray.reshape(x.shape[0], 32, 8, 32, 8).transpose(0, 1, 3, 2, 4).reshape(x.shape[0], 32, 32, 64)
    const = (gp == gp[..., :1]).all(-1)
    assert const.sum() > 2000 and (~const).sum() > 2000
    assert np.array_equal(np.isnan(flat), ~const) and np.array_equal(flat[const], gp[..., 0][const])
    frames = (xd * 255.0).round().to(torch.uint8).permute(0, 2, 3, 1).contiguous()
    x2, f8, f16 = cg.entropy_maps_u8(frames)
    e8b, _ = cg.entropy_maps(x2)
    assert torch.equal(f8._cgic_flat8.nan_to_num(-1.0), e8b._cgic_flat8.nan_to_num(-1.0))
    r8, r16 = cg.entropy_maps(xd, reference_order=True)
    for c, m in ((0.1, 0.8), (0.5, 0.3), (0.0, 0.6), (0.2, 0.0)):
        router = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)
        want = router(r16, r8, want_gate=False)[0]
        with_flat = router(e16, e8, want_gate=False)[0]                                   # pixels + flat8 ride on the maps
        no_flat = router(e16, e8, want_gate=False, pixels=xd, flat8=torch.full_like(e8, float("nan")))[0]
        assert all(torch.equal(a, b) for a, b in zip(with_flat, want)), (c, m)
        assert all(torch.equal(a, b) for a, b in zip(no_flat, want)), (c, m)


def test_refinement_plumbing_and_limits():
    """Entropy modules hand their pixels to the router through the reference's own call chain (maps carry them); shapes that do
    not belong to the maps are refused; segments beyond the LDS are routed on the maps as given (documented limit)"""
    from oracle.content_families import families
    x = torch.from_numpy(families(n=8)["smooth8"]).to(DEV)
    e8m, e16m = cg.Entropy(8).to(DEV)(x), cg.Entropy(16).to(DEV)(x)
    assert e8m._cgic_pixels is x and e16m._cgic_pixels is x
    r8, r16 = cg.entropy_maps(x, reference_order=True)
    router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    want = router(r16, r8)[0]
    assert all(torch.equal(a, b) for a, b in zip(router(e16m, e8m)[0], want))            # pixels found on the maps
    plain = router(e16m.clone(), e8m.clone())[0]                                            # no pixels: the maps decide
    off = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
    off.refine = False
    assert all(torch.equal(a, b) for a, b in zip(off(e16m, e8m)[0], plain))
    with pytest.raises(ValueError, match="do not belong"):
        router(e16m, e8m, pixels=x[:, :, :128])
    # flattened-batch routing (the reference's encode() semantics, RouterTriple.py:21,40,52,63) of 8 smooth images fits the LDS and
    # IS refined: equal to the batch-global routing on the reference-arithmetic maps
    flat_router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=False)
    want_flat = flat_router(r16, r8)[0]
    assert all(torch.equal(p, q) for p, q in zip(flat_router(e16m, e8m)[0], want_flat))
    assert all(torch.equal(p, q) for p, q in zip(flat_router(e16m.clone(), e8m.clone(), pixels=x)[0], want_flat))
    # ... of 32 images it does not fit the LDS: refined all the same (ABI 8: patched copies of the maps, a chain of launches) --
    # pixels found on the maps or handed over, the masks are those of the reference-arithmetic maps
    big = torch.from_numpy(np.ascontiguousarray(_tie_sets(n=32, flat=1)["smooth8"])).to(DEV)
    b8, b16 = cg.Entropy(8).to(DEV)(big), cg.Entropy(16).to(DEV)(big)
    assert cg._lib.lib().cgic_router_refine_supported(32, 16, 16, 0) and not cg._lib.lib().cgic_router_refine_in_lds(32, 16, 16, 0)
    q8, q16 = cg.entropy_maps(big, reference_order=True)
    want_big = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=False)(q16, q8)[0]
    a = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=False)(b16, b8)[0]
    b = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=False)(b16.clone(), b8.clone(), pixels=big)[0]
    assert all(torch.equal(p, q) for p, q in zip(a, want_big)) and all(torch.equal(p, q) for p, q in zip(b, want_big))
    # NaN pixels: the NaN patches sort last in both; nothing hangs
    xn = x.clone()
    xn[0, 0, 5, 7] = float("nan")
    e8, e16 = cg.entropy_maps(xn)
    r8, r16 = cg.entropy_maps(xn, reference_order=True)
    assert all(torch.equal(p, q) for p, q in zip(router(e16, e8, pixels=xn)[0], router(r16, r8)[0]))


def test_router_constant_map_selects_nothing():
    e16 = torch.full((1, 16, 16), 3.351e-4, device=DEV)
    e8 = torch.full((1, 32, 32), 3.351e-4, device=DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.1, 0.8)(e16, e8)
    assert mode == 0 and int(mask[0].sum()) == 0 and int(mask[1].sum()) == 0 and int(mask[2].sum()) == 64 * 64


# ---------------------------------------------------------------------------- C'. entropy
@pytest.mark.parametrize("name", ["rand_64x96", "u8_48x80", "smooth_64x64", "const_32x32", "signed_32x32"])
def test_entropy_golden(golden, name):
    g = golden("entropy")
    x = _t(g[name + "_x"])
    e8, e16 = cg.entropy_maps(x)
    # fp32 with OCML exp/log vs torch's CPU exp/log (MKL VML) + a different summation order: 2e-6 absolute on values in [0, 3.5]
    # (measured <= 1e-6; the round-2 bar of 2e-5 would have hidden a regression large enough to flip masks)
    assert np.abs(e8.cpu().numpy() - g[name + "_e8"]).max() < 2e-6
    assert np.abs(e16.cpu().numpy() - g[name + "_e16"]).max() < 2e-6
    assert torch.equal(cg.Entropy(8).to(DEV)(x), e8) and torch.equal(cg.Entropy(16).to(DEV)(x), e16)


def test_entropy_batch_vs_oracle_and_determinism(orc):
    g = torch.Generator().manual_seed(2)
    x = torch.rand(4, 3, 256, 256, generator=g)
    e8, e16 = cg.entropy_maps(x.to(DEV))
    e8b, e16b = cg.entropy_maps(x.to(DEV))
    assert torch.equal(e8, e8b) and torch.equal(e16, e16b)          # fixed summation order: run-to-run identical
    assert np.abs(e8.cpu().numpy() - orc.entropy(x.numpy(), 8)).max() < 2e-6
    assert np.abs(e16.cpu().numpy() - orc.entropy(x.numpy(), 16)).max() < 2e-6
    # images do not interact: batch == per-image
    e8s, _ = cg.entropy_maps(x[2:3].to(DEV))
    assert torch.equal(e8s[0], e8[2])
    # ragged width (W % 64 != 0)
    x2 = torch.rand(1, 3, 48, 208, generator=g)
    a8, a16 = cg.entropy_maps(x2.to(DEV))
    assert np.abs(a8.cpu().numpy() - orc.entropy(x2.numpy(), 8)).max() < 2e-6
    assert np.abs(a16.cpu().numpy() - orc.entropy(x2.numpy(), 16)).max() < 2e-6


@pytest.mark.parametrize("B,H,W", [(1, 2048, 2048), (1, 16, 4096), (1, 4096, 16), (300, 16, 16), (1, 16, 16)])
def test_entropy_extreme_shapes_vs_oracle(orc, B, H, W):
    """an untiled 2048x2048 image, one-patch-high / -wide strips, many one-patch images"""
    x = np.random.default_rng(H + W + B).random((B, 3, H, W), dtype=np.float32)
    e8, e16 = cg.entropy_maps(_t(x))
    assert np.abs(e8.cpu().numpy() - orc.entropy(x, 8)).max() < 2e-6
    assert np.abs(e16.cpu().numpy() - orc.entropy(x, 16)).max() < 2e-6


@pytest.mark.parametrize("B,H,W", [(3, 256, 256), (1, 16, 16), (2, 48, 4080), (1, 768, 592)])
def test_uint8_frames_totensor_and_entropy_in_one_pass(orc, B, H, W):
    """cgic_entropy_maps_u8 (inference.py:50-59 + model.py:99-101): x == what T.ToTensor() makes of the frames on the CPU
    (permute, float, div(255)) bit for bit; the maps == entropy_maps(x) bit for bit and == the oracle within the entropy
    tolerance; the op and the x-less call agree"""
    rng = np.random.default_rng(B * H + W)
    frames = rng.integers(0, 256, (B, H, W, 3), dtype=np.uint8)
    frames[0, :16, :16] = np.arange(256, dtype=np.uint8).reshape(16, 16, 1)          # every byte value at least once
    ref_x = torch.from_numpy(frames).permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()
    x, e8, e16 = cg.entropy_maps_u8(_t(frames))
    assert torch.equal(x.cpu(), ref_x)
    f8, f16 = cg.entropy_maps(x)
    assert torch.equal(e8, f8) and torch.equal(e16, f16)
    assert np.abs(e8.cpu().numpy() - orc.entropy(ref_x.numpy(), 8)).max() < 2e-6
    assert np.abs(e16.cpu().numpy() - orc.entropy(ref_x.numpy(), 16)).max() < 2e-6
    nx, g8, g16 = cg.entropy_maps_u8(_t(frames), want_x=False)
    assert nx is None and torch.equal(g8, e8) and torch.equal(g16, e16)
    ox, o8, o16 = torch.ops.cgic.entropy_maps_u8(_t(frames))
    assert torch.equal(ox, x) and torch.equal(o8, e8) and torch.equal(o16, e16)
    with pytest.raises(ValueError):
        cg.entropy_maps_u8(_t(frames).float())


def test_entropy_edge_pixels_vs_oracle(orc):
    """the fixed-point deposit path on its corner cases: pixels outside [-1, 1] (no bin in reach: an all-zero histogram
    is 0 here, 2.9e-37 in the reference), exactly on bin centres / midway between two, saturated patches where all 64
    pixels of a sub-patch hit ONE bin (the same-address LDS atomics), 16x16 blocks, and NaN pixels (NaN patch, like the
    reference's NaN histogram) -- and a layout-independence check: the integer sums do not depend on where a patch sits"""
    g = torch.Generator().manual_seed(9)
    bins = np.linspace(-1, 1, 32, dtype=np.float32)
    cases = {
        "far_outside": torch.full((1, 3, 32, 32), 5.0),
        "outside_mix": torch.rand(1, 3, 32, 48, generator=g) * 6 - 3,
        "white": torch.ones(1, 3, 32, 32),
        "black": torch.zeros(1, 3, 32, 32),
        "on_centres": torch.from_numpy(bins[np.random.default_rng(1).integers(0, 32, (1, 1, 32, 32))]).repeat(1, 3, 1, 1) / 0.9999,
        "midway": torch.from_numpy((bins[:-1] + 0.5 * (bins[1] - bins[0]))[np.random.default_rng(2).integers(0, 31, (1, 1, 32, 32))]).repeat(1, 3, 1, 1),
        "blocks16": torch.rand(1, 3, 4, 4, generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3),
        "blocks8": torch.rand(1, 3, 8, 8, generator=g).repeat_interleave(8, 2).repeat_interleave(8, 3),
    }
    for name, x in cases.items():
        x = x.float().contiguous()
        e8, e16 = cg.entropy_maps(x.to(DEV))
        d8 = np.abs(e8.cpu().numpy() - orc.entropy(x.numpy(), 8)).max()
        d16 = np.abs(e16.cpu().numpy() - orc.entropy(x.numpy(), 16)).max()
        assert d8 < 2e-6 and d16 < 2e-6, (name, d8, d16)
    xn = torch.rand(2, 3, 32, 64, generator=g)
    xn[0, 1, 3, 5] = float("nan")
    xn[1, 0, 20, 62] = float("nan")
    e8, e16 = cg.entropy_maps(xn.to(DEV))
    assert torch.isnan(e8).nonzero().tolist() == [[0, 0, 0], [1, 2, 7]] and torch.isnan(e16).nonzero().tolist() == [[0, 0, 0], [1, 1, 3]]
    ok = ~torch.isnan(e8).cpu().numpy()
    assert np.abs(e8.cpu().numpy() - orc.entropy(xn.numpy(), 8))[ok].max() < 2e-6
    # the same 16x16 patch anywhere in an image (another wave, another lane-to-replica mapping) gives the same bits
    patch = torch.rand(1, 3, 16, 16, generator=g)
    tiled = patch.repeat(1, 1, 5, 13)
    e8, e16 = cg.entropy_maps(tiled.to(DEV))
    assert int((e16 != e16[0, 0, 0]).sum()) == 0 and int((e8[:, 0::2, 0::2] != e8[0, 0, 0]).sum()) == 0


# ---------------------------------------------------------------------------- D/E/F. coders
@pytest.mark.parametrize("name", ["zeros", "zipf", "big", "ties"])
def test_huffman_streams_golden(golden, name, tmp_path):
    g = golden("coders")
    h = cg.HuffmanCoding(_freq_mapping(g[name + "_freq"], g[name + "_order"]))
    for si in range(6):
        sym = g[f"{name}_s{si}_sym"].astype(np.int64)
        data = g[f"{name}_s{si}_bytes"].tobytes()
        p = h.compress(_t(sym), str(tmp_path / "s.bin"))
        assert open(p, "rb").read() == data, f"{name} stream {si}"      # byte-identical file
        dec = h.decompress_string(p)
        assert (dec is None and len(sym) == 0) or dec == sym.tolist()


def test_huffman_long_stream_uses_workspace(orc, golden):
    g = golden("coders")
    rng = np.random.default_rng(0)
    for name in ("zipf", "ties"):
        h = cg.HuffmanCoding(_freq_mapping(g[name + "_freq"], g[name + "_order"]))
        t = orc.HuffmanTable(g[name + "_freq"])
        sym = rng.integers(0, 1024, 36864)
        data = h.encode_to_bytes(_t(sym))
        assert data == orc.encode(t, sym)
        assert h.decode_bytes(data) == sym.tolist()


def test_huffman_keyerror_and_truncated_stream(golden):
    g = golden("coders")
    h = cg.HuffmanCoding(_freq_mapping(g["zipf_freq"], g["zipf_order"]))
    with pytest.raises(KeyError):
        h.encode_to_bytes(torch.tensor([1, 2, 5000], device=DEV))
    sym = np.arange(100)
    data = h.encode_to_bytes(_t(sym))
    dec = h.decode_bytes(data[:-3] + bytes([0]))       # cut: trailing partial code is dropped, no crash
    assert dec is not None and dec == sym.tolist()[:len(dec)] and len(dec) < 100


def test_binary_coding_golden(golden, tmp_path):
    g = golden("coders")
    b = cg.BinaryCoding()
    for n in (0, 1, 7, 8, 256, 1024, 2304):
        m, data = g[f"binary_{n}_mask"].astype(np.int32), g[f"binary_{n}_bytes"].tobytes()
        p = b.compress(_t(m), str(tmp_path / "m.bin"))
        assert open(p, "rb").read() == data
        dec = b.decompress_string(p)
        assert (dec is None and n == 0) or dec == m.tolist()


# ---------------------------------------------------------------------------- G. compress glue
def _cfg1_keys(g):
    return sorted({k[:-5] for k in g if k.endswith("_mode")})


def test_compress_config1_bit_identical_bins(golden, tmp_path):
    """config 1: every .bin the reference wrote for torch.rand(1,3,256,256), all 7 modes, both tables"""
    g = golden("compress_cfg1")
    gc = golden("coders")
    cb = _t(g["codebook"])
    codecs = {"zipf": cg.GrainCodec(_freq_mapping(gc["zipf_freq"], gc["zipf_order"]), cb),
              "zeros": cg.GrainCodec(_freq_mapping(np.zeros(1024), gc["zipf_order"]), cb)}
    vq = _make_vq(g["codebook"])
    # config 1's own image (inference.py:157-166: torch.manual_seed(0); torch.rand(1, 3, 256, 256)), regenerated from the
    # seed and checked against the sum the generator stored: PIXELS -> entropy maps -> masks on the GPU
    torch.manual_seed(0)
    x = torch.rand(1, 3, 256, 256)
    assert float(x.double().sum()) == float(g["x_seed0_sum"]), "torch's CPU generator changed: the fixture's image cannot be regenerated"
    e8x, e16x = cg.entropy_maps(x.to(DEV))
    assert np.abs(e8x.cpu().numpy() - g["e8"]).max() < 2e-6 and np.abs(e16x.cpu().numpy() - g["e16"]).max() < 2e-6
    for key in _cfg1_keys(g):
        tname, ri = key.split("_r")
        c, m = (float(v) for v in g["ratios"][int(ri)])
        mode = int(g[key + "_mode"])
        # the path end to end at the hot path's own inputs: entropies -> masks, latent -> indices -> bytes
        mask, _, _, rmode = cg.TripleGrainFixedEntropyRouter(c, m)(_t(g["e16"]), _t(g["e8"]))
        assert rmode == mode
        # ... and from the pixels: the GPU's own entropy maps give the reference's masks for this image, hence its bytes
        mask_px, _, _, mode_px = cg.TripleGrainFixedEntropyRouter(c, m)(e16x, e8x)
        assert mode_px == mode and all(torch.equal(a, b) for a, b in zip(mask_px, mask)), f"{key}: pixels -> masks differs from the reference"
        for k, t in zip("cmf", mask_px):
            assert np.array_equal(t.cpu().numpy()[0, 0], unpack_mask(g[f"{key}_m{k}"], tuple(t.shape[-2:])))
        ind = vq.indices(_t(g[key + "_z"]))
        assert np.array_equal(ind.cpu().numpy().reshape(64, 64), g[key + "_ind"].astype(np.int64))
        comp = codecs[tname].compress(ind, mask, mode)
        files = comp.to_host()[0]
        on = cg.mode_streams(mode)
        assert set(files) == {n for n, o in zip(cg.STREAM_NAMES, on) if o}
        for n in files:
            assert files[n] == g[f"{key}_{n}"].tobytes(), f"{key}: {n}.bin differs"
        assert comp.bpp(256 * 256)[0] == float(g[key + "_bpp"])                  # bpp match
        paths = comp.write_legacy(str(tmp_path))
        assert sorted(p.split("/")[-1] for p in paths) == sorted(n + ".bin" for n in files)
        # decode side
        rb = cg.CompressedBatch.read_legacy(str(tmp_path), mode, 64, 64, comp.data.shape[2], DEV)
        dind, dmask, zq, status = codecs[tname].decompress(rb)
        assert int(status[0]) == 0
        assert np.array_equal(dind.cpu().numpy()[0], g[key + "_qdec_ind"].astype(np.int64))
        assert np.array_equal(zq.cpu().numpy()[0], g["codebook"][g[key + "_qdec_ind"].astype(np.int64)].transpose(2, 0, 1))
        for k, t in zip("cmf", dmask):
            assert np.array_equal(t.cpu().numpy()[0, 0], unpack_mask(g[f"{key}_m{k}"], tuple(t.shape[-2:])))


def _random_case(rng, B, h, w, c, m):
    e16 = (rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)
    e8 = (rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)
    ind = rng.integers(0, 1024, (B, h, w))
    # make the index grid consistent with what the encoder merge produces: constant inside a coarse /
    # medium cell is NOT required by the coder (it reads the top-left element), so keep it fully random
    return e16, e8, ind


@pytest.mark.parametrize("B,h,w", [(64, 64, 64), (2, 192, 192), (3, 192, 148), (1, 4, 4),
                                   # beyond the tiling driver's sizes: a 2048x2048 image untiled, one-cell-high / -wide strips, many tiny images
                                   (1, 512, 512), (1, 4, 4096), (1, 2048, 8), (700, 16, 16), (2, 516, 388)])
def test_compress_batch_vs_oracle_and_roundtrip(orc, golden, B, h, w):
    """configs 2-4: B=64 256^2 and 768^2-tile grids; all modes; bytes == oracle per image (sample) and
    encode -> decode round trip for every image (size-independent property)."""
    gc = golden("coders")
    rng = np.random.default_rng(h * 7 + B)
    cbk = rng.standard_normal((1024, 4)).astype(np.float32)
    codec = cg.GrainCodec(_freq_mapping(gc["zipf_freq"], gc["zipf_order"]), _t(cbk))
    htab = orc.HuffmanTable(gc["zipf_freq"])
    for c, m in ((0.1, 0.8), (0.0, 0.5), (0.5, 0.0), (0.5, 0.5), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0)):
        e16, e8, ind = _random_case(rng, B, h, w, c, m)
        mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(_t(e16), _t(e8))
        comp = codec.compress(_t(ind), mask, mode)
        host = comp.to_host()
        mks = [t.cpu().numpy() for t in mask]
        for b in sorted({0, B // 2, B - 1}):
            ref = orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, htab)
            assert host[b] == ref, f"mode {mode} image {b}"
        dind, dmask, zq, status = codec.decompress(comp)
        assert int(status.abs().max()) == 0
        # expected merged grid: the value each granularity's top-left element carries
        exp = np.where(mks[2][:, 0] == 1, ind, 0)
        exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
        exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
        assert np.array_equal(dind.cpu().numpy(), exp)
        for a, b_ in zip(dmask, mask):
            assert torch.equal(a, b_)
        assert np.array_equal(zq.cpu().numpy(), cbk[exp].transpose(0, 3, 1, 2))
        b0 = 0
        oind, _, _, _ = orc.decompress_image(host[b0], mode, h, w, htab)
        assert np.array_equal(oind, exp[b0])


def test_decoders_agree_on_adversarial_streams(orc, golden):
    """Streams that re-synchronise slowly or never: a table of EQUAL code lengths (uniform counters: every code is 10 bits, a
    wrong entry offset stays wrong for ever -- the self-synchronising decoder has to guess inside the residue class of the
    gcd of the lengths), periodic streams of one long / one short codeword, two alternating symbols, and the ordinary
    Zipf table with all-fine grids (the longest streams).  Decoded indices == the indices that were encoded, both decoders,
    and the bytes == the oracle's."""
    gc = golden("coders")
    rng = np.random.default_rng(77)
    cbk = _t(rng.standard_normal((1024, 4)).astype(np.float32))
    zipf = gc["zipf_freq"]
    order = orc.param_dict_order(1024)          # the iteration order of the reference's ParameterDict (ties follow it)
    tables = {"uniform": np.full(1024, 7.0), "zipf": zipf.astype(np.float64)}
    B, h, w = 3, 64, 64
    for tname, freq in tables.items():
        codec = cg.GrainCodec(_freq_mapping(freq, order), cbk)
        htab = orc.HuffmanTable(freq)
        long_sym = int(np.argmin(freq)) if tname == "zipf" else 5
        short_sym = int(np.argmax(freq)) if tname == "zipf" else 900
        grids = [np.full((h, w), long_sym), np.full((h, w), short_sym),
                 np.where((np.arange(h * w).reshape(h, w) % 2) == 0, long_sym, short_sym)]
        ind = np.stack(grids).astype(np.int64)
        for c, m in ((0.0, 0.0), (0.1, 0.8), (0.0, 1.0)):
            e16 = (rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)
            e8 = (rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)
            mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(_t(e16), _t(e8))
            comp = codec.compress(_t(ind), mask, mode)
            host = comp.to_host()
            mks = [t.cpu().numpy() for t in mask]
            for b in range(B):
                assert host[b] == orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, htab), (tname, mode, b)
            dind, dmask, zq, status = codec.decompress(comp)
            assert int(status.abs().max()) == 0
            if tname == "zipf" and (c, m) == (0.0, 0.0):
                # a stream that never re-synchronises (one 13-bit codeword repeated) is one fix-point sweep per chunk for the guessing
                # decoder: it gives up after 32 sweeps and the all-entries pass (every entry offset walked, functions composed) finishes
                cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
                cg._lib.lib().cgic_decode_stats(cnt.data_ptr())
                try:
                    d2 = codec.decompress(comp, decoder="throughput")
                    torch.cuda.synchronize()
                finally:
                    cg._lib.lib().cgic_decode_stats(None)
                assert torch.equal(d2[0], dind) and int(cnt[1]) == B and 32 <= int(cnt[2]) <= 33, cnt.tolist()
            exp = np.where(mks[2][:, 0] == 1, ind, 0)
            exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
            exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
            assert np.array_equal(dind.cpu().numpy(), exp), (tname, mode)


@pytest.mark.parametrize("n_sym", [2, 3, 64])
def test_small_tables_round_trip(orc, n_sym):
    """code tables of 2 / 3 / 64 symbols (1-bit codes: a 2-entry decode LUT) through both decoders: bytes == oracle, indices back"""
    rng = np.random.default_rng(n_sym)
    freq = rng.integers(1, 50, n_sym).astype(np.float64)
    order = orc.param_dict_order(n_sym)
    codec = cg.GrainCodec(_freq_mapping(freq, order), _t(rng.standard_normal((n_sym, 4)).astype(np.float32)))
    htab = orc.HuffmanTable(freq)
    B, h, w = 3, 32, 48
    ind = rng.integers(0, n_sym, (B, h, w))
    for c, m in ((0.1, 0.8), (0.0, 0.0), (0.5, 0.5)):
        e16 = (rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)
        e8 = (rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)
        mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(_t(e16), _t(e8))
        comp = codec.compress(_t(ind), mask, mode)
        host = comp.to_host()
        mks = [t.cpu().numpy() for t in mask]
        for b in range(B):
            assert host[b] == orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, htab), (n_sym, mode, b)
        dind, _, zq, status = codec.decompress(comp)
        assert int(status.abs().max()) == 0
        exp = np.where(mks[2][:, 0] == 1, ind, 0)
        exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
        exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
        assert np.array_equal(dind.cpu().numpy(), exp), (n_sym, mode)


def test_decompress_flags_corrupt_streams(golden):
    gc = golden("coders")
    rng = np.random.default_rng(1)
    codec = cg.GrainCodec(_freq_mapping(gc["zipf_freq"], gc["zipf_order"]), _t(rng.standard_normal((1024, 4)).astype(np.float32)))
    e16, e8, ind = _random_case(rng, 2, 64, 64, 0.1, 0.8)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(_t(e16), _t(e8))
    comp = codec.compress(_t(ind), mask, mode)
    host = comp.to_host()
    bad = [dict(host[0]), dict(host[1])]
    bad[1]["indices_medium"] = bad[1]["indices_medium"][:40]             # symbol count no longer matches its mask
    rb = cg.CompressedBatch.from_host(bad, mode, 64, 64, comp.data.shape[2], DEV)
    _, _, _, status = codec.decompress(rb)
    assert int(status[0]) == 0 and int(status[1]) != 0                   # the reference raises there


# ---------------------------------------------------------------------------- pipeline
def test_pipeline_graph_replay_equals_eager(golden):
    """HotPathPipeline: one batch = five dependent launches; captured into a hipGraph and replayed it leaves what the eager
    calls leave (streams, indices, z_q, loss, decoded indices) and an exact usage histogram"""
    gc = golden("coders")
    rng = np.random.default_rng(11)
    B = 16
    x = _t(rng.random((B, 3, 64, 64), dtype=np.float32))
    z = _t(rng.standard_normal((B, 4, 16, 16), dtype=np.float32))
    vq = _make_vq(rng.standard_normal((1024, 4)).astype(np.float32))
    vq.usage_counter.copy_(_t(gc["zipf_freq"].astype(np.float32)))
    ref = cg.HotPathPipeline(vq, 0.1, 0.8)
    h1 = torch.zeros(1024, dtype=torch.int64, device=DEV)
    r1 = ref.run(x, z, h1)[0]
    torch.cuda.synchronize()
    for kw in ({"fuse_router": False}, {"prepare": True}, {"refine": False}):
        hn = torch.zeros(1024, dtype=torch.int64, device=DEV)
        r = cg.HotPathPipeline(vq, 0.1, 0.8, **kw).run(x, z, hn)[0]
        torch.cuda.synchronize()
        assert torch.equal(hn, h1) and torch.equal(r["ind"], r1["ind"]) and torch.equal(r["z_q"], r1["z_q"])
        assert r["comp"].to_host() == r1["comp"].to_host() and float(r["loss"]) == float(r1["loss"])
    pipe = cg.HotPathPipeline(vq, 0.1, 0.8, prepare=True)
    hg = torch.zeros(1024, dtype=torch.int64, device=DEV)
    pipe.run(x, z, hg)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        pipe.run(x, z, hg)
    g, rs = cg.capture_graph(lambda: pipe.run(x, z, hg), side)
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    hg.zero_()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    assert torch.equal(hg, 3 * h1)
    assert torch.equal(rs[0]["ind"], r1["ind"]) and torch.equal(rs[0]["dec"][0], r1["dec"][0])
    assert rs[0]["comp"].to_host() == r1["comp"].to_host() and float(rs[0]["loss"]) == float(r1["loss"])


@pytest.mark.parametrize("u8", [False, True])
def test_one_call_driver_equals_the_four_calls(u8):
    """cgic_compress_image (pipeline.HotCall): CGIC.compress's hot path for a batch as ONE foreign call over buffers allocated once
    == entropy_maps / vq_forward_route / GrainCodec.compress / .decompress called one by one, bit for bit (maps, masks, indices,
    z_q, loss, every stream byte, decoded indices and rows, usage histogram), fp32 pixels and uint8 frames, twice in a row"""
    from control_gic_amd.quantize import vq_forward_route
    from oracle.content_families import families
    rng = np.random.default_rng(11)
    B, H, W = 5, 128, 192
    vq = _make_vq(rng.standard_normal((1024, 4), dtype=np.float32))
    vq.usage_counter.copy_(torch.from_numpy(np.floor(1e6 / (1 + np.arange(1024)) ** 1.1).astype(np.float32)))
    hist = torch.zeros(1024, dtype=torch.int64, device=DEV)
    hc = cg.pipeline.HotCall(vq, 0.1, 0.8, B, H, W, u8=u8, hist=hist)
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    hsum = torch.zeros_like(hist)
    for rep in range(2):
        xf = families(n=B, H=H, W=W, seed=3 + rep)["smooth8" if rep else "noise8"]
        z = _t(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32))
        frames = torch.from_numpy(np.round(xf * 255.0).astype(np.uint8).transpose(0, 2, 3, 1).copy()).to(DEV)
        x = frames if u8 else _t(xf)
        out = hc(x, z)
        torch.cuda.synchronize()
        if u8:
            xs, e8, e16 = cg.entropy_maps_u8(frames)
            assert torch.equal(out["x"], xs) and torch.equal(xs, _t(xf))
        else:
            e8, e16 = cg.entropy_maps(x)
        zq, loss, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=x)
        h2 = torch.zeros_like(hist)
        comp = codec.compress(ind, mask, mode, hist=h2)
        dind, dmask, dq, status = codec.decompress(comp)
        assert mode == out["mode"] and torch.equal(e8, out["e8"]) and torch.equal(e16, out["e16"])
        assert torch.equal(ind, out["ind"]) and torch.equal(zq, out["z_q"]) and torch.equal(loss, out["loss"])
        assert all(torch.equal(a, b) for a, b in zip(mask, out["mask"]))
        assert comp.to_host() == out["comp"].to_host()
        assert torch.equal(dind, out["dec"][0]) and torch.equal(dq, out["dec"][2]) and int(out["dec"][3].abs().max()) == 0
        assert all(torch.equal(a, b) for a, b in zip(dmask, out["dec"][1]))
        hsum += h2
        assert torch.equal(hist, hsum)                            # the usage histogram accumulates over the calls


def test_refreshed_codebook_reaches_captured_graphs():
    """advisor r3: refresh_codebook() used to allocate a NEW prepared image, so graphs captured earlier kept reading the old
    (freed) one.  The image is rewritten in place now: a replay after refresh_codebook() quantises against the new weights."""
    g = torch.Generator().manual_seed(5)
    vq = _make_vq(torch.randn(1024, 4, generator=g).numpy())
    x = torch.rand(4, 3, 64, 64, generator=g).to(DEV)
    z = torch.randn(4, 4, 16, 16, generator=g).to(DEV)
    pipe = cg.HotPathPipeline(vq, 0.1, 0.8, prepare=True)
    pipe.run(x, z)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    gr, rs = cg.capture_graph(lambda: pipe.run(x, z, decode=False), side)
    torch.cuda.synchronize()
    ptr0 = pipe.prepared.data_ptr()
    with torch.no_grad():
        vq.embedding.weight.copy_(torch.randn(1024, 4, generator=g).to(DEV))
    pipe.refresh_codebook()
    assert pipe.prepared.data_ptr() == ptr0
    gr.replay()
    torch.cuda.synchronize()
    want = cg.HotPathPipeline(vq, 0.1, 0.8).run(x, z, decode=False)[0]
    torch.cuda.synchronize()
    assert torch.equal(rs[0]["ind"], want["ind"]) and torch.equal(rs[0]["z_q"], want["z_q"])
    # the module's own snapshot follows the weight's version counter
    vq.eval().prepare()
    with torch.no_grad():
        i0 = vq(z)[2].clone()
        vq.embedding.weight.mul_(-1.0)
        i1 = vq(z)[2]
        assert torch.equal(i1, cg.quantize._vq_forward(z, vq.embedding.weight, vq.beta, vq.legacy, None)[2]) and not torch.equal(i0, i1)


def test_fused_vq_router_launch_equals_separate_calls(orc):
    """cgic_vq_forward_route_f32: router workgroups appended to the VQ grid -- identical outputs"""
    from control_gic_amd.quantize import _vq_forward, vq_forward_route
    rng = np.random.default_rng(21)
    for (B, h16, w16) in ((64, 16, 16), (3, 12, 9), (1, 48, 48)):
        z = _t(rng.standard_normal((B, 4, 4 * h16, 4 * w16), dtype=np.float32))
        w = _t(rng.standard_normal((1024, 4), dtype=np.float32))
        e16 = _t((rng.random((B, h16, w16)) * 2.6).astype(np.float32))
        e8 = _t((rng.random((B, 2 * h16, 2 * w16)) * 2.6).astype(np.float32))
        for c, m in ((0.1, 0.8), (0.0, 0.5), (0.5, 0.5), (0.0, 0.0)):
            for per_image in (True, False):
                zq, loss, idx, mask, gate, mode = vq_forward_route(z, w, 0.25, True, e16, e8, c, m, per_image=per_image, want_gate=True)
                zq2, loss2, idx2 = _vq_forward(z, w, 0.25, True, None)
                mask2, gate2, _, mode2 = cg.TripleGrainFixedEntropyRouter(c, m, per_image=per_image)(e16, e8)
                assert mode == mode2 and torch.equal(idx, idx2) and torch.equal(zq, zq2) and float(loss) == float(loss2)
                assert all(torch.equal(a, b) for a, b in zip(mask, mask2)) and torch.equal(gate, gate2)


def test_mixed_stream_example_runs():
    """examples/mixed_stream.py (BASELINE config 5 in miniature): tiled + untiled images, histogram total == number of
    latent vectors, every tile decodes back -- asserted inside the script"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(root, "examples", "mixed_stream.py")], capture_output=True, text=True,
                       timeout=600, env=env)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "histogram total 991232" in r.stdout


# ---------------------------------------------------------------------------- f2. quant_conv / post_quant_conv fused
def _conv_seq(x, W, b, bias_first):
    """the two fp32 rounding sequences torch's CPU Conv2d(4, 4, 1) uses (include/cgic_hip.h): fma chain over the input
    channels in order, bias seeding the accumulator or added last.  x [N,4] -> [N,4], exact (fma emulated in float64:
    a*b is exact there and the sum of a 48-bit product and a float rounds once to double, then to float -- a double
    rounding in ~1e-9 of the cases, none on these inputs, checked against torch below)."""
    def fma(a, xv, c):
        return (np.float64(a) * xv.astype(np.float64) + c.astype(np.float64)).astype(np.float32)
    out = np.empty_like(x)
    for c in range(4):
        if bias_first:
            acc = np.full(x.shape[0], b[c], np.float32)
            for k in range(4):
                acc = fma(W[c, k], x[:, k], acc)
        else:
            acc = (W[c, 0] * x[:, 0]).astype(np.float32)
            for k in range(1, 4):
                acc = fma(W[c, k], x[:, k], acc)
            acc = (acc + b[c]).astype(np.float32)
        out[:, c] = acc
    return out


@pytest.mark.parametrize("shape", [(1, 4, 64, 64), (2, 4, 48, 40), (1, 4, 20, 36)])
def test_fused_quant_conv_matches_the_cpu_convolution(orc, shape):
    """VQ with quant_conv fused == VQ of torch's own CPU Conv2d output, indices and z_q bit for bit, for whichever of
    the two accumulation orders the CPU convolution took on this host for this shape (it depends on shape and threads)"""
    from control_gic_amd.quantize import _vq_forward
    torch.manual_seed(3)
    conv = torch.nn.Conv2d(4, 4, 1)
    h = torch.randn(shape)
    cb = torch.randn(1024, 4)
    with torch.no_grad():
        z_cpu = conv(h)                                              # the reference's quant_conv on the CPU
    hf = h.permute(0, 2, 3, 1).reshape(-1, 4).numpy()
    zf = z_cpu.permute(0, 2, 3, 1).reshape(-1, 4).numpy()
    W, b = conv.weight.detach().numpy().reshape(4, 4), conv.bias.detach().numpy()
    orders = [bf for bf in (False, True) if np.array_equal(_conv_seq(hf, W, b, bf), zf)]
    assert len(orders) >= 1, "torch's CPU conv matches neither documented sequence"
    ozq, oloss, oidx = orc.vq(z_cpu.numpy(), cb.numpy())
    convd = conv.to(DEV)
    for kernel in ("mfma", "valu"):
        zq, loss, idx = _vq_forward(h.to(DEV), cb.to(DEV), 0.25, True, None, kernel=kernel, quant_conv=convd, conv_bias_first=orders[0])
        assert np.array_equal(idx.cpu().numpy(), oidx)               # bit-exact
        assert np.array_equal(zq.cpu().numpy(), ozq)                 # bit-exact: z_q = z + (e - z) on the convolved latent
        assert abs(float(loss) - float(oloss)) <= 1e-6 * abs(float(oloss))
    # the other order is the other documented sequence, not noise: it reproduces _conv_seq(..., not order) exactly
    other = not orders[0]
    z_other = _conv_seq(hf, W, b, other).reshape(shape[0], shape[2], shape[3], 4).transpose(0, 3, 1, 2)
    _, _, oidx2 = orc.vq(np.ascontiguousarray(z_other), cb.numpy())
    _, _, idx2 = _vq_forward(h.to(DEV), cb.to(DEV), 0.25, True, None, quant_conv=convd, conv_bias_first=other)
    assert np.array_equal(idx2.cpu().numpy(), oidx2)
    # no bias
    conv_nb = torch.nn.Conv2d(4, 4, 1, bias=False)
    with torch.no_grad():
        z_nb = conv_nb(h)
    _, _, oidx3 = orc.vq(z_nb.numpy(), cb.numpy())
    _, _, idx3 = _vq_forward(h.to(DEV), cb.to(DEV), 0.25, True, None, quant_conv=conv_nb.to(DEV))
    assert np.array_equal(idx3.cpu().numpy(), oidx3)


def test_fused_post_quant_conv_is_a_second_gather(orc):
    """decompress(..., post_quant_conv) returns (quant, post_quant_conv(quant)): the second one bit-identical to the
    documented sequence applied to the gathered rows, i.e. to the CPU Conv2d (same two-order caveat)"""
    torch.manual_seed(5)
    g = torch.Generator().manual_seed(5)
    B, h, w = 3, 32, 48
    cbk = torch.randn(1024, 4, generator=g)
    vq = _make_vq(cbk.numpy())
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    e16 = (torch.rand(B, h // 4, w // 4, generator=g) * 2.6).to(DEV)
    e8 = (torch.rand(B, h // 2, w // 2, generator=g) * 2.6).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(e16, e8)
    ind = torch.randint(0, 1024, (B, h, w), generator=g).to(DEV)
    comp = codec.compress(ind, mask, mode)
    conv = torch.nn.Conv2d(4, 4, 1)
    ind_d, mask_d, (quant, quant2), status = codec.decompress(comp, post_quant_conv=conv.to(DEV))
    ind_p, _, quant_p, _ = codec.decompress(comp)
    assert int(status.abs().max()) == 0 and torch.equal(ind_d, ind_p) and torch.equal(quant, quant_p)
    qf = quant.permute(0, 2, 3, 1).reshape(-1, 4).cpu().numpy()
    W, b = conv.weight.detach().cpu().numpy().reshape(4, 4), conv.bias.detach().cpu().numpy()
    assert np.array_equal(quant2.permute(0, 2, 3, 1).reshape(-1, 4).cpu().numpy(), _conv_seq(qf, W, b, False))
    with torch.no_grad():
        ref = conv.cpu()(quant.cpu())
    # torch's CPU result is one of the two sequences; both are within an ulp or two of each other
    assert torch.equal(ref, quant2.cpu()) or np.array_equal(ref.permute(0, 2, 3, 1).reshape(-1, 4).numpy(), _conv_seq(qf, W, b, True))
    assert (ref - quant2.cpu()).abs().max() <= 1e-6


# ---------------------------------------------------------------------------- f4. training: backward kernel
@pytest.mark.parametrize("legacy", [True, False])
def test_vq_backward_kernel_vs_formula(legacy):
    """csrc/cgic_vq_bwd.hip against quantize.py:85-93 differentiated by torch autograd on the same tensors: dz bit-identical
    to the analytic expression, the codebook gradient within 2e-6 relative of an fp64 scatter-add (torch's own fp32
    index_add_ is itself order-dependent), and identical from run to run"""
    g = torch.Generator().manual_seed(11)
    B, h, w, K = 4, 24, 40, 1024
    z = torch.randn(B, 4, h, w, generator=g).to(DEV).requires_grad_()
    vq = cg.VectorQuantizer(K, 4, beta=0.25, legacy=legacy).to(DEV).train()
    vq.embedding.weight.data.copy_(torch.randn(K, 4, generator=g))
    zq, loss, idx = vq(z)
    gq = torch.randn(B, 4, h, w, generator=g).to(DEV)
    (zq * gq).sum().add(3.0 * loss).backward()
    gz, gw = z.grad.clone(), vq.embedding.weight.grad.clone()
    # reference expressions (quantize.py:83-93) on the GPU through torch autograd
    z2 = z.detach().clone().requires_grad_()
    wt = vq.embedding.weight.detach().clone().requires_grad_()
    zf = z2.permute(0, 2, 3, 1).reshape(-1, 4)
    e = wt[idx]
    if legacy:
        l2 = torch.mean((e.detach() - zf) ** 2) + 0.25 * torch.mean((e - zf.detach()) ** 2)
    else:
        l2 = 0.25 * torch.mean((e.detach() - zf) ** 2) + torch.mean((e - zf.detach()) ** 2)
    q2 = (zf + (e - zf).detach()).view(B, h, w, 4).permute(0, 3, 1, 2)
    (q2 * gq).sum().add(3.0 * l2).backward()
    assert torch.allclose(gz, z2.grad, rtol=1e-5, atol=1e-8)                     # autograd's own op order differs by an ulp
    # the kernel's documented expression, exactly: g_zq + (g_loss * (-2/n * w_z)) * (e - z)
    n = float(z.numel())
    w_z, w_e = (1.0, 0.25) if legacy else (0.25, 1.0)
    diff = (vq.embedding.weight.detach()[idx] - z.detach().permute(0, 2, 3, 1).reshape(-1, 4))
    cz = torch.tensor(3.0, device=DEV) * (-(2.0 / n) * w_z)
    want_gz = gq + (cz * diff).view(B, h, w, 4).permute(0, 3, 1, 2)
    assert torch.equal(gz, want_gz)                                              # bit-exact
    want_gw = torch.zeros(K, 4, dtype=torch.float64, device=DEV).index_add_(0, idx, diff.double()) * (3.0 * (2.0 / n) * w_e)
    assert torch.allclose(gw.double(), want_gw, rtol=2e-6, atol=1e-12)
    assert torch.allclose(gw, wt.grad, rtol=1e-4, atol=1e-9)                     # torch's fp32 atomics
    # deterministic: a second backward gives the same bits
    z.grad = None; vq.embedding.weight.grad = None
    zq, loss, idx = vq(z)
    (zq * gq).sum().add(3.0 * loss).backward()
    assert torch.equal(vq.embedding.weight.grad, gw) and torch.equal(z.grad, gz)
    # only one of the two gradients wanted
    z3 = z.detach().clone().requires_grad_()
    with torch.no_grad():
        pass
    vq.embedding.weight.requires_grad_(False)
    zq, loss, _ = vq(z3)
    (zq * gq).sum().add(3.0 * loss).backward()
    assert torch.equal(z3.grad, gz)


def test_custom_ops_match_the_module_path():
    """torch.ops.cgic.* == the module classes (same kernels), and autograd flows through cgic::vq_forward into the backward kernel"""
    g = torch.Generator().manual_seed(21)
    z = torch.randn(2, 4, 16, 32, generator=g).to(DEV)
    cb = torch.randn(1024, 4, generator=g).to(DEV)
    x = torch.rand(2, 3, 64, 128, generator=g).to(DEV)
    from control_gic_amd.quantize import _vq_forward
    zq, loss, idx = torch.ops.cgic.vq_forward(z, cb, 0.25, True)
    zq2, loss2, idx2 = _vq_forward(z, cb, 0.25, True, None)
    assert torch.equal(zq, zq2) and torch.equal(idx, idx2) and float(loss) == float(loss2)
    e8, e16 = torch.ops.cgic.entropy_maps(x)
    e8b, e16b = cg.entropy_maps(x)
    assert torch.equal(e8, e8b) and torch.equal(e16, e16b)
    m = torch.ops.cgic.router(e16, e8, 0.1, 0.8, True)
    mb, _, _, _ = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(e16, e8)
    assert all(torch.equal(a, b) for a, b in zip(m, mb))
    out = torch.ops.cgic.vq_forward_route(z, cb, 0.25, True, e16, e8, 0.1, 0.8, True)
    assert torch.equal(out[0], zq) and torch.equal(out[2], idx) and all(torch.equal(a, b) for a, b in zip(out[3:], mb))
    zr, cr = z.clone().requires_grad_(), cb.clone().requires_grad_()
    q, l, _ = torch.ops.cgic.vq_forward(zr, cr, 0.25, True)
    (q.sum() * 0.5 + 2.0 * l).backward()
    vq = _make_vq(cb.cpu().numpy()).train()
    zm = z.clone().requires_grad_()
    q2, l2, _ = vq(zm)
    (q2.sum() * 0.5 + 2.0 * l2).backward()
    assert torch.equal(zr.grad, zm.grad) and torch.equal(cr.grad, vq.embedding.weight.grad)


def test_prepared_codebook_image_changes_nothing():
    """cgic_vq_prepare_f32: the codebook image made once == the image every workgroup derives itself -- indices, z_q, loss,
    masks bit-identical with and without it, for several K, scales, a fused quant_conv, ragged and aligned shapes; the module's
    prepare() snapshot is dropped by train() and load_state_dict()"""
    from control_gic_amd.quantize import _vq_forward, vq_forward_route, prepare_codebook
    g = torch.Generator().manual_seed(5)
    for K, scale, (B, h, w) in ((1024, 1.0, (3, 16, 16)), (1024, 1.0 / 1024, (2, 16, 24)), (256, 37.0, (1, 20, 12)), (64, 1.0, (2, 8, 8)), (1024, 1.0, (5, 7, 9))):
        cb = (torch.randn(K, 4, generator=g) * scale).to(DEV)
        z = (torch.randn(B, 4, h, w, generator=g) * scale).to(DEV)
        prep = prepare_codebook(cb)
        assert prep is not None and prep.numel() == cg._lib.lib().cgic_vq_prepared_bytes(K)
        a = _vq_forward(z, cb, 0.25, True, None)
        b = _vq_forward(z, cb, 0.25, True, None, prepared=prep)
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and float(a[1]) == float(b[1])
        conv = torch.nn.Conv2d(4, 4, 1).to(DEV)
        a = _vq_forward(z, cb, 0.25, True, None, quant_conv=conv)
        b = _vq_forward(z, cb, 0.25, True, None, quant_conv=conv, prepared=prep)
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and float(a[1]) == float(b[1])
        if h % 4 == 0 and w % 4 == 0:
            e16 = torch.rand(B, h // 4, w // 4, generator=g).to(DEV) * 2.6
            e8 = torch.rand(B, h // 2, w // 2, generator=g).to(DEV) * 2.6
            a = vq_forward_route(z, cb, 0.25, True, e16, e8, 0.1, 0.8)
            b = vq_forward_route(z, cb, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep)
            assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
    assert cg._lib.lib().cgic_vq_prepared_bytes(2048) == 0 and prepare_codebook(torch.randn(2048, 4).to(DEV)) is None
    # degenerate codebooks (all zero / non-finite): the image carries the maxima that switch the filter off
    for cbk in (torch.zeros(1024, 4), torch.full((1024, 4), float("nan"))):
        cbk = cbk.to(DEV)
        z = torch.randn(2, 4, 8, 8, generator=g).to(DEV)
        a = _vq_forward(z, cbk, 0.25, True, None)
        b = _vq_forward(z, cbk, 0.25, True, None, prepared=prepare_codebook(cbk))
        assert torch.equal(a[2], b[2])
    vq = _make_vq(torch.randn(1024, 4, generator=g).numpy())
    z = torch.randn(2, 4, 16, 16, generator=g).to(DEV)
    with torch.no_grad():
        ref = vq(z)
        vq.prepare()
        assert vq._prepared_image() is not None
        got = vq(z)
        assert torch.equal(ref[2], got[2]) and torch.equal(ref[0], got[0]) and torch.equal(vq.indices(z), ref[2])
    assert vq._prepared_image() is None                 # autograd on: the live weights
    vq.train()
    assert getattr(vq, "_prepared", None) is None
    vq.eval().prepare()
    vq.load_state_dict(vq.state_dict())
    assert vq._prepared is None


@pytest.mark.parametrize("case", VQ_CASES)
def test_vq_golden_through_the_prepared_image(golden, case):
    """the reference's own outputs again, through cgic_vq_prepare_f32's image -- which packs near-duplicate rows into one tile when
    there are any (dup_rows, codes_as_z, lattice_ties: exact ties go to the lowest ORIGINAL index like torch.argmin)"""
    from control_gic_amd.quantize import _vq_forward, prepare_codebook
    g = golden("vq")
    cb = _t(g[case + "_cb"])
    prep = prepare_codebook(cb)
    with torch.no_grad():
        zq, loss, idx = _vq_forward(_t(g[case + "_z"]), cb, 0.25, True, None, prepared=prep)
    assert np.array_equal(idx.cpu().numpy(), g[case + "_idx"].astype(np.int64).ravel())
    assert np.array_equal(zq.cpu().numpy(), g[case + "_zq"])
    ref = float(g[case + "_loss"])
    assert abs(float(loss) - ref) <= 1e-6 * abs(ref) + 1e-12


def test_clustered_codebook_is_packed_into_tiles():
    """A trained codebook holds clusters of near-duplicate rows and exact duplicates (quantize.py:22-26,78).  Spread over the tiles
    they send every vector to the all-K exact scan; cgic_vq_prepare_f32 packs each cluster into one 32-code tile (keys =
    original index << 16 | position; ties by ORIGINAL index).  Checks: results identical to the independent VALU kernel and to
    the unprepared filter path -- N(0,1) latents, latents ON rows, on midpoints of duplicate rows (exact ties) --, fused with the
    router, and the telemetry: groups rerun on the exact loop drop from ~all to a few per cent."""
    from control_gic_amd.quantize import _vq_forward, vq_forward_route, prepare_codebook
    rng = np.random.default_rng(21)
    centres = rng.standard_normal((64, 4)).astype(np.float32)
    cb = (centres[rng.integers(0, 64, 1024)] + np.float32(1e-4) * rng.standard_normal((1024, 4)).astype(np.float32)).astype(np.float32)
    dup_src = rng.integers(0, 1024, 40)
    dup_dst = rng.integers(0, 1024, 40)
    cb[dup_dst] = cb[dup_src]                                   # exact duplicates at random places
    z = rng.standard_normal((16, 4, 64, 64)).astype(np.float32)
    zf = z.transpose(0, 2, 3, 1).reshape(-1, 4)
    zf[:1024] = cb                                              # latents exactly on rows (duplicates: distance 0 twice)
    zf[1024:1064] = (cb[dup_src] + cb[(dup_src + 1) % 1024]) / 2        # midpoints: whatever ties there are
    z = np.ascontiguousarray(zf.reshape(16, 64, 64, 4).transpose(0, 3, 1, 2))
    cbd, zd = _t(cb), _t(z)
    lib = cg._lib.lib()
    cnt = torch.zeros(8, dtype=torch.int32, device=DEV)
    with torch.no_grad():
        want = _vq_forward(zd, cbd, 0.25, True, None, kernel="valu")
        assert lib.cgic_vq_stats(cnt.data_ptr()) == 0
        try:
            plain = _vq_forward(zd, cbd, 0.25, True, None)
            torch.cuda.synchronize()
            c_plain = cnt.cpu().numpy().copy()
            cnt.zero_()
            prep = prepare_codebook(cbd)
            got = _vq_forward(zd, cbd, 0.25, True, None, prepared=prep)
            torch.cuda.synchronize()
            c_prep = cnt.cpu().numpy().copy()
        finally:
            lib.cgic_vq_stats(None)
        for r in (plain, got):
            assert torch.equal(r[2], want[2]) and torch.equal(r[0], want[0])
            assert abs(float(r[1]) - float(want[1])) <= 1e-6 * abs(float(want[1]))
        groups = 16 * 64 * 64 // 64
        assert c_plain[1] > 0.9 * groups                        # without the packing: (nearly) every group falls back to the exact loop
        assert c_prep[1] < 0.05 * groups, (c_prep, groups)       # with it: a few per cent
        e16 = _t(rng.random((16, 16, 16)).astype(np.float32) * 2.6)
        e8 = _t(rng.random((16, 32, 32)).astype(np.float32) * 2.6)
        a = vq_forward_route(zd, cbd, 0.25, True, e16, e8, 0.1, 0.8)
        b = vq_forward_route(zd, cbd, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep)
        assert torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and all(torch.equal(x, y) for x, y in zip(a[3], b[3]))
        assert torch.equal(b[2], want[2])
        # an image that was made for another codebook (same address reused) or copied elsewhere is never trusted blindly
        moved = prep.clone()
        c = _vq_forward(zd, cbd, 0.25, True, None, prepared=moved)
        assert torch.equal(c[2], want[2])


def test_decoder_choice_is_per_call_two_threads():
    """the prefix decoder is an argument of each cgic_decompress_streams call (ABI 4), not a process-wide switch: two threads
    decode the same batch at the same time, one per decoder, on their own streams, and both match the one-thread result"""
    import threading
    g = torch.Generator().manual_seed(9)
    vq = _make_vq(torch.randn(1024, 4, generator=g).numpy())
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    B, h, w = 6, 64, 64
    ind = torch.randint(0, 1024, (B, h, w), generator=g).to(DEV)
    e16 = (torch.rand(B, h // 4, w // 4, generator=g) * 2.6).to(DEV)
    e8 = (torch.rand(B, h // 2, w // 2, generator=g) * 2.6).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(e16, e8)
    comp = codec.compress(ind, mask, mode)
    ref = codec.decompress(comp, decoder="latency")
    torch.cuda.synchronize()
    out, err = {}, []

    def work(name):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(40):
                    if it % 2:
                        r = codec.decompress(comp, decoder=name)            # named in the call
                    else:
                        with cg.decoder_mode(name):                         # or the thread's default
                            r = codec.decompress(comp)
                    assert torch.equal(r[0], ref[0]) and torch.equal(r[2], ref[2]) and int(r[3].abs().max()) == 0
                out[name] = r
            st.synchronize()
        except Exception as e:          # noqa: BLE001 -- reported by the main thread
            err.append((name, repr(e)))

    ts = [threading.Thread(target=work, args=(n,)) for n in ("latency", "throughput")]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err
    assert set(out) == {"latency", "throughput"}
    with pytest.raises(ValueError):
        codec.decompress(comp, decoder="fastest")


@pytest.mark.parametrize("lanes,ring,graph,native", [(4, True, True, True), (4, True, True, False), (3, False, True, False), (2, True, False, True), (1, True, True, True)])
def test_lane_stream_independent_streams_are_bit_identical(lanes, ring, graph, native):
    """pipeline.LaneStream (batch t on HIP stream t % lanes, one ring graph per lane, NO dependency between the lanes: up to
    `lanes` batches in flight) leaves exactly what the one-stream order leaves in every slot -- streams, indices, z_q, loss,
    masks, decoded rows -- and an exact usage histogram, whatever mix of ring and per-slot graphs a submit() takes"""
    import control_gic_amd.pipeline as pl
    g = torch.Generator().manual_seed(41)
    vq = _make_vq(torch.randn(1024, 4, generator=g).numpy())
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    shapes = [(4, 64, 96), (4, 64, 96), (2, 128, 64), (4, 64, 96), (3, 32, 32), (4, 64, 96), (4, 64, 96), (1, 256, 256)]
    slots = [(torch.rand(b, 3, H, W, generator=g).to(DEV), torch.randn(b, 4, H // 4, W // 4, generator=g).to(DEV)) for b, H, W in shapes]
    hist = torch.zeros(1024, dtype=torch.int64, device=DEV)
    ls = pl.LaneStream(vq, 0.1, 0.8, slots, lanes=lanes, hist=hist, ring=ring, graph=graph, max_ring=3, native_launch=native)
    ls.capture()
    torch.cuda.synchronize()
    hist.zero_()
    runs = [0] * len(slots)
    t = 0
    for n in (3, 8, 1, 16, 5):          # partial rotations, whole rotations, remainders
        ls.submit(n)
        for k in range(t, t + n):       # batch k -> lane k % L, the lane's slots in rotation
            L = len(ls.lanes)
            lane = k % L
            per_lane = len(slots[lane::L])
            runs[lane + L * ((k // L) % per_lane)] += 1
        t += n
    ls.join()
    torch.cuda.synchronize()
    assert sum(runs) == 33
    ref_pipe = pl.HotPathPipeline(vq, 0.1, 0.8)
    hist_ref = torch.zeros(1024, dtype=torch.int64, device=DEV)
    for k, (x, z) in enumerate(slots):
        r = None
        for _ in range(runs[k]):
            r = ref_pipe.run(x, z, hist_ref, decode=True)[0]
        if r is None:
            continue
        s = ls.slots[k]
        assert s.enc["comp"].to_host() == r["comp"].to_host()
        assert torch.equal(s.enc["ind"], r["ind"]) and torch.equal(s.enc["z_q"], r["z_q"]) and float(s.enc["loss"]) == float(r["loss"])
        assert torch.equal(s.enc["e8"], r["e8"]) and torch.equal(s.enc["e16"], r["e16"])
        assert all(torch.equal(a, b) for a, b in zip(s.enc["mask"], r["mask"]))
        assert torch.equal(s.dec[0], r["dec"][0]) and torch.equal(s.dec[2], r["dec"][2]) and int(s.dec[3].abs().max()) == 0
    torch.cuda.synchronize()
    assert torch.equal(hist, hist_ref)
    assert int(hist.sum()) == sum(runs[k] * shapes[k][0] * (shapes[k][1] // 4) * (shapes[k][2] // 4) for k in range(len(slots)))


@pytest.mark.parametrize("lanes,max_ring", [(2, 8), (2, 2), (4, 3), (1, 8), (4, 8)])
def test_lane_stream_refilled_slots_between_partial_submits(lanes, max_ring):
    """slots are refilled IN PLACE between submits whose lengths are not multiples of the rotation: every slot must be read
    back from the launch that really processed its new input, whichever graph (one step, a run of a rotation, a run that
    wraps around) the submit took (round-2 advisor finding: per-slot graphs and ring graphs used to own different output
    buffers and `slot.enc` named the ring's)"""
    import control_gic_amd.pipeline as pl
    g = torch.Generator().manual_seed(77)
    vq = _make_vq(torch.randn(1024, 4, generator=g).numpy())
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    n_slots = 4
    slots = [(torch.rand(3, 3, 64, 64, generator=g).to(DEV), torch.randn(3, 4, 16, 16, generator=g).to(DEV)) for _ in range(n_slots)]
    ls = pl.LaneStream(vq, 0.1, 0.8, slots, lanes=lanes, max_ring=max_ring)
    ls.capture()
    ref_pipe = pl.HotPathPipeline(vq, 0.1, 0.8)
    L = len(ls.lanes)
    per_lane = [len(slots[j::L]) for j in range(L)]
    t = 0
    for rnd, n in enumerate((4, 3, 5, 1, 2, 7, 9)):
        torch.cuda.synchronize()
        for x, z in slots:                               # new content in the same tensors
            x.copy_(torch.rand(x.shape, generator=g))
            z.copy_(torch.randn(z.shape, generator=g))
        torch.cuda.synchronize()
        ls.fork()
        if rnd % 2:
            ls.prepare(n)
        ls.submit(n)
        ls.join()
        torch.cuda.synchronize()
        ran = set()
        for k in range(t, t + n):
            lane = k % L
            ran.add(lane + L * ((k // L) % per_lane[lane]))
        t += n
        for k in ran:
            r = ref_pipe.run(slots[k][0], slots[k][1], None, decode=True)[0]
            s = ls.slots[k]
            assert s.enc["comp"].to_host() == r["comp"].to_host(), (rnd, k)
            assert torch.equal(s.enc["ind"], r["ind"]) and torch.equal(s.enc["z_q"], r["z_q"]), (rnd, k)
            assert all(torch.equal(a, b) for a, b in zip(s.enc["mask"], r["mask"])), (rnd, k)
            assert torch.equal(s.dec[0], r["dec"][0]) and torch.equal(s.dec[2], r["dec"][2]) and int(s.dec[3].abs().max()) == 0, (rnd, k)


def test_batch_stream_two_stream_schedule_is_bit_identical():
    """experimental.BatchStream (measured, not adopted: kept bit-identical all the same; encode side of batch i+1 next to the decode side of batch i, graphs on two HIP streams, rotating
    slots) produces exactly what the one-stream order produces: same streams, indices, masks, rows; exact usage histogram"""
    import control_gic_amd.pipeline as pl
    g = torch.Generator().manual_seed(31)
    cbk = torch.randn(1024, 4, generator=g)
    vq = _make_vq(cbk.numpy())
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    slots = [(torch.rand(4, 3, 64, 96, generator=g).to(DEV), torch.randn(4, 4, 16, 24, generator=g).to(DEV)) for _ in range(3)]
    hist = torch.zeros(1024, dtype=torch.int64, device=DEV)
    from control_gic_amd.experimental import BatchStream
    bs = BatchStream(vq, 0.1, 0.8, slots, hist=hist)
    bs.capture()
    hist.zero_()
    bs.submit(7)                      # 7 steps over 3 slots: slots are reused while the other stream still works
    bs.join()
    torch.cuda.synchronize()
    ref_pipe = pl.HotPathPipeline(vq, 0.1, 0.8)
    hist_ref = torch.zeros(1024, dtype=torch.int64, device=DEV)
    counts = [3, 2, 2]                # how often each slot ran
    for k, (x, z) in enumerate(slots):
        r = ref_pipe.run(x, z, hist_ref, decode=True)[0]
        for _ in range(counts[k] - 1):
            ref_pipe.run(x, z, hist_ref, decode=True)
        s = bs.slots[k]
        assert s.enc["comp"].to_host() == r["comp"].to_host()
        assert torch.equal(s.enc["ind"], r["ind"]) and torch.equal(s.enc["z_q"], r["z_q"]) and float(s.enc["loss"]) == float(r["loss"])
        assert all(torch.equal(a, b) for a, b in zip(s.enc["mask"], r["mask"]))
        assert torch.equal(s.dec[0], r["dec"][0]) and torch.equal(s.dec[2], r["dec"][2]) and int(s.dec[3].abs().max()) == 0
    assert torch.equal(hist, hist_ref) and int(hist.sum()) == 7 * 4 * 16 * 24
