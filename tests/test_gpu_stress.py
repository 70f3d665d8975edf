"""GPU: the evidence behind the VQ filter path's exactness claim, under the driver's eyes.

The filter kernel (csrc/cgic_vq.hip) finds candidates with fp16 MFMAs and decides in fp32; its result must be the
reference's argmin for EVERY finite input.  These tests compare it with the CPU ORACLE (not with the repository's own
VALU kernel) on: the full benchmark batch, adversarial near-ties whose reference distances differ by exactly 0 / 1 / 2
ulp with the two codes placed in different tiles and row halves, non-finite inputs, and time-boxed random sweeps
(the former tools/stress_vq.py / stress_codec.py)."""
import os
import time

import numpy as np
import pytest
import torch

import control_gic_amd as cg
from control_gic_amd.quantize import _vq_forward
from conftest import ROOT

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _gpu_vq(z, cb, kernel="mfma"):
    zq, loss, idx = _vq_forward(torch.from_numpy(z).to(DEV), torch.from_numpy(cb).to(DEV), 0.25, True, None, kernel=kernel)
    return zq.cpu().numpy(), float(loss), idx.cpu().numpy()


def test_vq_full_benchmark_batch_vs_oracle(orc):
    """all 262 144 vectors of the benchmark batch (B=64, 64x64 latents, K=1024): indices and z_q == oracle"""
    rng = np.random.default_rng(2024)
    z = rng.standard_normal((64, 4, 64, 64), dtype=np.float32)
    cb = rng.standard_normal((1024, 4), dtype=np.float32)
    zq, loss, idx = _gpu_vq(z, cb)
    ozq, oloss, oidx = orc.vq(z, cb)
    assert np.array_equal(idx, oidx)                         # bit-exact
    assert np.array_equal(zq, ozq)                           # bit-exact
    assert abs(loss - float(oloss)) <= 1e-6 * abs(float(oloss))      # the mean's summation order


def _near_tie_problem(orc, rng, n_vec, K, scale):
    """n_vec latent vectors, each with a planted pair of codes (a, b) at mirrored offsets around it; code b is then moved
    along one component, ulp by ulp, until the REFERENCE-sequence distances d_b - d_a hit a target of 0, +-1, +-2 quanta
    (searched on the oracle's own distance row; a quantum = the spacing of achievable reference distances, i.e. the
    ulp of zz + ee, because d = (zz + ee) - 2 mm cancels).  Every vector plants its pair into ONE shared codebook at
    random rows -- any tile, any row half, either order; the vectors sit far apart, so pairs do not interfere."""
    cb = (rng.standard_normal((K, 4)) * 4.0 * scale).astype(np.float32)          # background codes, far from the planted clusters
    centers = (rng.standard_normal((n_vec, 4)) * scale).astype(np.float32)
    centers += ((np.arange(n_vec, dtype=np.float32)[:, None] % 7 - 3) * np.float32(0.37 * scale)).astype(np.float32)
    rows = rng.permutation(K)[:2 * n_vec].reshape(n_vec, 2)
    z = centers.copy()
    hits = {0: 0, 1: 0, 2: 0, "none": 0}
    targets = [0, 1, -1, 2, -2]
    steps = np.arange(-400, 401)
    for i in range(n_vec):
        c = centers[i]
        delta = (rng.standard_normal(4) * 0.02 * scale).astype(np.float32)
        a, b = int(rows[i, 0]), int(rows[i, 1])
        ea = (c + delta).astype(np.float32)
        eb = (c - delta).astype(np.float32)
        comp = int(np.argmax(np.abs(delta)))
        # the family e_b(s): component `comp` moved by s ulps (integer steps on the bit pattern of a non-zero float)
        cand = np.repeat(eb[None], steps.size, axis=0)
        bits = cand[:, comp].view(np.int32).astype(np.int64) + steps * (1 if eb[comp] > 0 else -1)
        cand[:, comp] = bits.astype(np.int32).view(np.float32)
        d_a = float(orc.vq_distances(c, ea[None])[0])
        d_b = orc.vq_distances(c, cand).astype(np.float64)
        diff = d_b - d_a
        nz = np.abs(diff[diff != 0])
        want = targets[i % len(targets)]
        found = False
        if nz.size:
            q = nz.min()                                        # one quantum of the reference arithmetic here
            ok = np.nonzero(np.abs(diff - want * q) < 0.25 * q)[0]
            if ok.size:
                eb = cand[ok[np.argmin(np.abs(steps[ok]))]].copy()
                found = True
        hits[abs(want) if found else "none"] += 1
        cb[a], cb[b] = ea, eb
    return z, cb, hits


@pytest.mark.parametrize("scale", [1.0, 1.0 / 512, 37.0])
def test_vq_adversarial_near_ties_vs_oracle(orc, scale):
    """two codes in DIFFERENT tiles / row halves whose reference distances differ by 0, 1 or 2 ulp, for a non-lattice z:
    the index must be the oracle's (lowest index on exact ties, the strictly smaller distance otherwise)"""
    rng = np.random.default_rng(int(scale * 1000) + 5)
    n_vec, K = 384, 1024
    z, cb, hits = _near_tie_problem(orc, rng, n_vec, K, scale)
    assert hits[0] >= 20 and hits[1] >= 40 and hits[2] >= 40, hits         # the search really produced the cases
    zz = np.ascontiguousarray(z.T.reshape(1, 4, 16, n_vec // 16))            # [B=1, C, h, w], vector n at position n
    zq, loss, idx = _gpu_vq(zz, cb)
    ozq, oloss, oidx = orc.vq(zz, cb)
    assert np.array_equal(idx, oidx), f"{(idx != oidx).sum()} of {n_vec} planted near-ties resolved differently from the oracle"
    assert np.array_equal(zq, ozq)
    # the planted pair really is the contest: the oracle's winner is one of the two planted rows almost everywhere
    # (and the VALU restatement agrees as well)
    _, _, idx_valu = _gpu_vq(zz, cb, kernel="valu")
    assert np.array_equal(idx_valu, oidx)


def test_vq_same_distance_many_codes_vs_oracle(orc):
    """many exactly tied codes spread over tiles and halves (duplicates of the winner): lowest index wins, as in torch.argmin"""
    rng = np.random.default_rng(77)
    cb = rng.standard_normal((1024, 4), dtype=np.float32)
    z = rng.standard_normal((2, 4, 16, 24), dtype=np.float32)
    _, _, base = orc.vq(z, cb)
    # duplicate each of 40 popular winners into 3 random other rows
    for w in np.unique(base)[:40]:
        for r in rng.integers(0, 1024, 3):
            cb[r] = cb[w]
    zq, loss, idx = _gpu_vq(z, cb)
    ozq, _, oidx = orc.vq(z, cb)
    assert np.array_equal(idx, oidx) and np.array_equal(zq, ozq)


def test_vq_nonfinite_inputs_confined(orc):
    """Deviation from torch.argmin's first-NaN rule (include/cgic_hip.h): it must be confined to vectors whose distance
    row contains a non-finite value.  Every other vector of the same launch still equals the oracle bit for bit."""
    rng = np.random.default_rng(9)
    z = rng.standard_normal((3, 4, 16, 32), dtype=np.float32)
    cb = rng.standard_normal((1024, 4), dtype=np.float32)
    zf = z.transpose(0, 2, 3, 1).reshape(-1, 4)
    bad = rng.permutation(zf.shape[0])[:60]
    poison = [np.nan, np.inf, -np.inf, 3e38, -3e38, 1e30]          # 3e38: zz overflows to inf; 1e30: finite distances
    for i, n in enumerate(bad):
        zf[n, i % 4] = poison[i % len(poison)]
    z = np.ascontiguousarray(zf.reshape(3, 16, 32, 4).transpose(0, 3, 1, 2))
    zq, loss, idx = _gpu_vq(z, cb)
    ozq, _, oidx = orc.vq(z, cb)
    rows_finite = np.array([np.isfinite(orc.vq_distances(v, cb)).all() for v in zf])
    assert rows_finite.sum() >= zf.shape[0] - 60 and (~rows_finite).sum() >= 40
    assert np.array_equal(idx[rows_finite], oidx[rows_finite])                     # untouched by their neighbours
    assert ((idx >= 0) & (idx < 1024)).all()                                       # affected rows: some valid code
    zq_f = zq.transpose(0, 2, 3, 1).reshape(-1, 4)
    ozq_f = ozq.transpose(0, 2, 3, 1).reshape(-1, 4)
    assert np.array_equal(zq_f[rows_finite], ozq_f[rows_finite])
    # the two GPU kernels agree on the affected rows too (same documented rule: non-finite distances never win)
    _, _, idx_valu = _gpu_vq(z, cb, kernel="valu")
    assert np.array_equal(idx, idx_valu)

    # a non-finite CODE puts a non-finite distance into every row: ignore that code, argmin over the finite ones
    cb2 = cb.copy()
    cb2[[5, 700]] = [[np.nan, 0, 0, 0], [np.inf, 1, 1, 1]]
    z2 = rng.standard_normal((1, 4, 16, 16), dtype=np.float32)
    _, _, idx2 = _gpu_vq(z2, cb2)
    cb3 = np.delete(cb, [5, 700], axis=0)
    keep = np.delete(np.arange(1024), [5, 700])
    d = ((z2.transpose(0, 2, 3, 1).reshape(-1, 1, 4).astype(np.float64) - cb3[None].astype(np.float64)) ** 2).sum(-1)
    srt = np.sort(d, axis=1)
    clear = srt[:, 1] - srt[:, 0] > 1e-4                                           # skip rows where rounding could decide
    assert np.array_equal(idx2[clear], keep[np.argmin(d, axis=1)][clear])


def _random_vq_case(rng):
    B = int(rng.integers(1, 9)); h = int(rng.integers(1, 40)); w = int(rng.integers(1, 40))
    K = int(rng.choice([64, 128, 256, 512, 1024]))
    zs = float(10 ** rng.uniform(-3, 3)); cs = float(10 ** rng.uniform(-3, 3))
    kind = int(rng.integers(0, 6))
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
    if kind == 5:   # one giant code among small ones: the fp16 scaling is set by a code that never wins
        cb[int(rng.integers(0, K))] *= 1e4
    return z.numpy(), cb.numpy(), dict(B=B, h=h, w=w, K=K, zs=zs, cs=cs, kind=kind)


def test_vq_random_sweep_time_boxed(orc):
    """~25 s of random shapes, scales 1e-3..1e3, K in {64..1024}, clustered / on-code / quantised / VQGAN-init / one-giant-code
    inputs: filter path == oracle (small cases) and == VALU kernel (all cases), indices and z_q bit-identical"""
    rng = np.random.default_rng(int(os.environ.get("CGIC_STRESS_SEED", "0")))
    t0 = time.time(); n = nvec = n_oracle = 0
    while time.time() - t0 < float(os.environ.get("CGIC_STRESS_SECONDS", "25")):
        z, cb, desc = _random_vq_case(rng)
        zq, loss, idx = _gpu_vq(z, cb)
        zq_v, loss_v, idx_v = _gpu_vq(z, cb, kernel="valu")
        assert np.array_equal(idx, idx_v) and np.array_equal(zq, zq_v), desc
        assert loss == loss_v or abs(loss - loss_v) <= 1e-6 * abs(loss_v), desc
        if idx.size * cb.shape[0] <= 3e6:                        # the scalar oracle: a few ms
            ozq, _, oidx = orc.vq(z, cb)
            assert np.array_equal(idx, oidx) and np.array_equal(zq, ozq), desc
            n_oracle += 1
        n += 1; nvec += idx.size
    assert n >= 50 and n_oracle >= 20, (n, n_oracle)


def test_codec_random_sweep_time_boxed(orc, golden):
    """~20 s of random grid sizes, ratios (all 7 modes) and code tables (max code length 13 / 17 / 128 / 224 bits):
    compressed bytes == oracle, decode == merge of the encoded grids"""
    g = golden("coders")
    rng = np.random.default_rng(int(os.environ.get("CGIC_STRESS_SEED", "0")) + 1)

    class _Item:
        def __init__(self, v): self.v = v
        def item(self): return self.v

    tables = {}
    for name in ("zipf", "big", "ties", "zeros"):
        mapping = {str(int(k)): _Item(float(g[name + "_freq"][int(k)])) for k in g[name + "_order"]}
        tables[name] = (mapping, orc.HuffmanTable(g[name + "_freq"]))
    cbk = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(DEV)
    codecs = {n: cg.GrainCodec(tables[n][0], cbk) for n in tables}
    ratios = [(0.1, 0.8), (0.1, 0.4), (0.7, 0.3), (0.3, 0.7), (0.0, 0.4), (0.4, 0.0), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0), (0.5, 0.5), (0.33, 0.33)]
    t0 = time.time(); n = 0
    while time.time() - t0 < float(os.environ.get("CGIC_STRESS_SECONDS", "20")):
        B = int(rng.integers(1, 6)); h = 4 * int(rng.integers(1, 50)); w = 4 * int(rng.integers(1, 50))
        name = str(rng.choice(list(tables), p=[0.5, 0.3, 0.1, 0.1]))
        if name in ("ties", "zeros") and h * w > 64 * 64:
            h, w = 32, 48                                           # 128/224-bit codes take the one-wave path
        c, m = ratios[int(rng.integers(0, len(ratios)))]
        e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).to(DEV)
        e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).to(DEV)
        ind = rng.integers(0, 1024, (B, h, w))
        mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(c, m, per_image=True)(e16, e8)
        comp = codecs[name].compress(torch.from_numpy(ind).to(DEV), mask, mode)
        host = comp.to_host()
        mks = [t.cpu().numpy() for t in mask]
        b = int(rng.integers(0, B))
        ref = orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, tables[name][1])
        desc = dict(B=B, h=h, w=w, table=name, ratio=(c, m), mode=mode, image=b)
        assert host[b] == ref, desc
        dind, dmask, zq, status = codecs[name].decompress(comp)
        exp = np.where(mks[2][:, 0] == 1, ind, 0)
        exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
        exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
        assert int(status.abs().max()) == 0 and np.array_equal(dind.cpu().numpy(), exp), desc
        assert all(torch.equal(a, b_) for a, b_ in zip(dmask, mask)), desc
        n += 1
    assert n >= 30, n


def test_split_streams_survive_ticket_pool_wrap(orc, golden):
    """Grids beyond 64x64 split their index streams over several workgroups that talk through library-owned ticket slots
    (compress: descriptors + flags, decode: flags).  The eager pool is a ring of 16 384 slots shared by every kernel: run
    enough launches to wrap it several times -- a slot that is not returned all-zero breaks a later launch (bytes differ,
    or workgroups wait forever: the test is time-limited by pytest-timeout / the driver)."""
    g = golden("coders")
    rng = np.random.default_rng(5)

    class _Item:
        def __init__(self, v): self.v = v
        def item(self): return self.v

    mapping = {str(int(k)): _Item(float(g["zipf_freq"][int(k)])) for k in g["zipf_order"]}
    table = orc.HuffmanTable(g["zipf_freq"])
    cbk = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(DEV)
    codec = cg.GrainCodec(mapping, cbk)
    B, h, w = 3, 92, 208
    e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).to(DEV)
    e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).to(DEV)
    ind = rng.integers(0, 1024, (B, h, w))
    ind_d = torch.from_numpy(ind).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.33, 0.33, per_image=True)(e16, e8)
    mks = [t.cpu().numpy() for t in mask]
    first = codec.compress(ind_d, mask, mode).to_host()
    for b in range(B):
        assert first[b] == orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, table)
    dind0 = codec.decompress(codec.compress(ind_d, mask, mode))[0].clone()
    # 28 slots per iteration (18 compress + 9 decode + the router/VQ-free rest): 2400 iterations = four times round the ring
    for it in range(2400):
        comp = codec.compress(ind_d, mask, mode)
        dind, _, _, status = codec.decompress(comp)
        if it % 200 == 199:
            assert comp.to_host() == first, it
            assert int(status.abs().max()) == 0 and torch.equal(dind, dind0), it
    torch.cuda.synchronize()


def test_captured_ticket_slots_are_recycled_and_eager_rings_are_per_stream(orc, golden):
    """(round-2 verdict item 8 / advisor) a serving loop that captures a hipGraph per image shape must not run the library's
    pool of captured ticket slots dry: cgic.capture_graph returns a graph's slots when the graph object dies.  600 captures of
    a split-stream compress + decompress of 40 tiles (360 slots each: the 262 144-slot pool would last 728 captures of ONE
    launch pair) with the graphs dropped as they go: the slots in use stay bounded and every replay gives the oracle's bytes.
    Eager launches take their slots from a ring per STREAM: two streams launching large split batches at the same time keep
    their exchanges apart (one shared ring of 16 384 slots wrapped after four such launches)."""
    import gc
    rng = np.random.default_rng(6)
    codec, table = _zipf_codec(orc, golden, rng)
    B, h, w = 40, 72, 120                        # 8640 fine positions: split streams (6 + 3 slots per image)
    e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).to(DEV)
    e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).to(DEV)
    ind = rng.integers(0, 1024, (B, h, w))
    ind_d = torch.from_numpy(ind).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.2, 0.5, per_image=True)(e16, e8)
    mks = [t.cpu().numpy() for t in mask]
    want = [orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, table) for b in (0, B - 1)]

    def step():
        comp = codec.compress(ind_d, mask, mode)
        return comp, codec.decompress(comp, decoder="latency")
    ind_dec = step()[1][0].clone()                   # (coarse / medium cells come back replicated: compare with the eager decode)
    torch.cuda.synchronize()
    lib = cg._lib.lib()
    base = lib.cgic_ticket_slots_in_use()
    side = torch.cuda.Stream()
    peak = 0
    for it in range(600):
        g, (comp, dec) = cg.capture_graph(step, side)
        with torch.cuda.stream(side):
            g.replay()
        side.synchronize()
        peak = max(peak, lib.cgic_ticket_slots_in_use() - base)
        if it % 150 == 0:
            host = comp.to_host()
            assert host[0] == want[0] and host[B - 1] == want[1] and int(dec[3].abs().max()) == 0 and torch.equal(dec[0], ind_dec)
        del g, comp, dec
        if it % 50 == 49:
            gc.collect()
    gc.collect()
    cg._lib.flush_released()                             # (slots go back at the next capture, or here: behind a device sync)
    assert peak <= 60 * 9 * B, peak                      # at most the captures between two collections are alive at once
    assert lib.cgic_ticket_slots_in_use() - base <= 9 * B
    # per-stream eager rings: two threads, two streams, large split batches back to back
    import threading
    err = []

    def work(k):
        try:
            st = torch.cuda.Stream()
            with torch.cuda.stream(st):
                for it in range(120):
                    comp, dec = step()
                    if it % 40 == 39:
                        host = comp.to_host()
                        assert host[0] == want[0] and host[B - 1] == want[1] and int(dec[3].abs().max()) == 0
            st.synchronize()
        except Exception as e:          # noqa: BLE001
            err.append((k, repr(e)))
    ts = [threading.Thread(target=work, args=(k,)) for k in range(2)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not err, err


def test_decompress_of_more_images_than_one_ticket_request(orc, golden):
    """1500 tiny images: the latency decoder takes 3 ticket slots per image and one request holds 4096, so the batch is cut into
    two launches of decode_split_kernel (rounds 1-2 kept a two-launch pair of kernels for this case: removed in round 3);
    both decoders agree and every image round-trips"""
    rng = np.random.default_rng(8)
    codec, table = _zipf_codec(orc, golden, rng)
    B, h, w = 1500, 8, 8
    e16 = torch.from_numpy((rng.random((B, 2, 2)) * 2.6).astype(np.float32)).to(DEV)
    e8 = torch.from_numpy((rng.random((B, 4, 4)) * 2.6).astype(np.float32)).to(DEV)
    ind = torch.from_numpy(rng.integers(0, 1024, (B, h, w))).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.25, 0.5, per_image=True)(e16, e8)
    comp = codec.compress(ind, mask, mode)
    a = codec.decompress(comp, decoder="latency")
    b = codec.decompress(comp, decoder="throughput")
    assert int(a[3].abs().max()) == 0 and int(b[3].abs().max()) == 0
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and all(torch.equal(x, y) for x, y in zip(a[1], b[1]))
    assert all(torch.equal(x, y) for x, y in zip(a[1], mask))
    fine = mask[2].reshape(B, -1).bool()
    assert torch.equal(a[0].reshape(B, -1)[fine], ind.reshape(B, -1)[fine])
    host = comp.to_host()
    mks = [t.cpu().numpy() for t in mask]
    for i in (0, 1364, 1365, 1366, B - 1):                      # either side of the launch boundary
        assert host[i] == orc.compress_image(ind[i].cpu().numpy(), mks[0][i, 0], mks[1][i, 0], mks[2][i, 0], mode, table)
        oind, _, _, _ = orc.decompress_image(host[i], mode, h, w, table)
        assert np.array_equal(a[0][i].cpu().numpy(), oind)


def _zipf_codec(orc, golden, rng):
    g = golden("coders")

    class _Item:
        def __init__(self, v): self.v = v
        def item(self): return self.v

    mapping = {str(int(k)): _Item(float(g["zipf_freq"][int(k)])) for k in g["zipf_order"]}
    cbk = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(DEV)
    return cg.GrainCodec(mapping, cbk), orc.HuffmanTable(g["zipf_freq"])


def _host_streams(comp, b):
    return comp.to_host()[b]


@pytest.mark.parametrize("pattern", ["one_per_part", "last_rows_only", "first_position_only", "none", "dense_medium", "all"])
def test_split_compress_corner_masks_vs_oracle(orc, golden, pattern):
    """the split-stream encoder (grids beyond 64x64: parts of <= 4096 positions that exchange symbol counts, bit counts and head
    bits): parts with no symbol at all, with ONE short codeword each (an output word then spans three and more parts), a stream
    that only the last / the first part contributes to, an empty stream, and fully selected grids -- bytes == oracle"""
    rng = np.random.default_rng(11)
    codec, table = _zipf_codec(orc, golden, rng)
    B, h, w = 2, 128, 160                       # fine 20480 positions -> 5 parts, medium 5120 -> unsplit short path, coarse 1280
    if pattern == "dense_medium":
        h, w = 192, 192                         # medium 9216 -> 3 parts
    ind = rng.integers(0, 1024, (B, h, w))
    ind[0, :, :] = np.where(rng.random((h, w)) < 0.5, 0, ind[0])        # many 1-3-bit codewords in image 0
    mc = np.zeros((B, 1, h // 4, w // 4), np.int32)
    mm = np.zeros((B, 1, h // 2, w // 2), np.int32)
    mf = np.zeros((B, 1, h, w), np.int32)
    if pattern == "one_per_part":
        flat = mf.reshape(B, -1)
        for p0 in range(0, h * w, 4096): flat[:, p0 + 7] = 1
        flat[1, 5000:5003] = 1
    elif pattern == "last_rows_only":
        mf[:, :, -3:, :] = 1; mm[:, :, -1:, :] = 1
    elif pattern == "first_position_only":
        mf[:, :, 0, 0] = 1; mm[:, :, 0, 0] = 1; mc[:, :, 0, 0] = 1
    elif pattern == "dense_medium":
        mm[:] = 1; mf[:, :, ::7, :] = 1
    elif pattern == "all":
        mc[:] = 1; mm[:] = 1; mf[:] = 1
    masks = [torch.from_numpy(m).to(DEV) for m in (mc, mm, mf)]
    comp = codec.compress(torch.from_numpy(ind).to(DEV), masks, 0)
    for b in range(B):
        ref = orc.compress_image(ind[b], mc[b, 0], mm[b, 0], mf[b, 0], 0, table)
        assert _host_streams(comp, b) == ref, (pattern, b)


def test_split_compress_bad_symbol_is_an_error_not_a_hang(orc, golden):
    """an index outside the code table inside ONE part of a split stream: every part of the stream learns about it through the
    exchange, the stream reports the reference's KeyError, the other streams and images are unaffected"""
    rng = np.random.default_rng(12)
    codec, table = _zipf_codec(orc, golden, rng)
    B, h, w = 2, 128, 160
    ind = rng.integers(0, 1024, (B, h, w))
    mc = (rng.random((B, 1, h // 4, w // 4)) < 0.1).astype(np.int32)
    mm = (rng.random((B, 1, h // 2, w // 2)) < 0.5).astype(np.int32)
    mf = (rng.random((B, 1, h, w)) < 0.3).astype(np.int32)
    mf[1, 0, 70, 3] = 1
    bad = ind.copy(); bad[1, 70, 3] = 4096                      # third part of image 1's fine stream
    masks = [torch.from_numpy(m).to(DEV) for m in (mc, mm, mf)]
    comp = codec.compress(torch.from_numpy(bad).to(DEV), masks, 0)
    nb = comp.nbytes.cpu().numpy()
    assert nb[1, 2] == cg._lib.ERR_INVALID - 10 and (nb[0] > 0).all() and (nb[1, [0, 1, 3, 4]] > 0).all()
    with pytest.raises(KeyError):
        comp.to_host()
    good = codec.compress(torch.from_numpy(ind).to(DEV), masks, 0)          # the slots were handed back clean
    for b in range(B):
        assert _host_streams(good, b) == orc.compress_image(ind[b], mc[b, 0], mm[b, 0], mf[b, 0], 0, table)


@pytest.mark.parametrize("ratio", [(0.001, 0.002), (0.999, 0.0005), (0.0, 0.001), (0.002, 0.0), (0.1, 0.8)])
@pytest.mark.parametrize("shape", [(128, 160), (192, 192), (68, 244)])
def test_split_decode_tiny_and_huge_streams_round_trip(orc, golden, ratio, shape):
    """the one-launch split decoder (grids beyond 64x64: 8 workgroups per stream exchanging range functions): streams of a few
    codewords (fewer 64-bit chunks than workgroups: most parts are empty and still publish) next to streams of tens of
    thousands, odd widths; bytes == oracle, decode == merge of the encoded grids, masks round-trip"""
    rng = np.random.default_rng(13)
    codec, table = _zipf_codec(orc, golden, rng)
    B = 3
    h, w = shape
    e16 = torch.from_numpy((rng.random((B, h // 4, w // 4)) * 2.6).astype(np.float32)).to(DEV)
    e8 = torch.from_numpy((rng.random((B, h // 2, w // 2)) * 2.6).astype(np.float32)).to(DEV)
    ind = rng.integers(0, 1024, (B, h, w))
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(*ratio, per_image=True)(e16, e8)
    comp = codec.compress(torch.from_numpy(ind).to(DEV), mask, mode)
    mks = [t.cpu().numpy() for t in mask]
    for b in range(B):
        assert _host_streams(comp, b) == orc.compress_image(ind[b], mks[0][b, 0], mks[1][b, 0], mks[2][b, 0], mode, table), (b, mode)
    dind, dmask, zq, status = codec.decompress(comp)
    exp = np.where(mks[2][:, 0] == 1, ind, 0)
    exp = exp + np.repeat(np.repeat(np.where(mks[1][:, 0] == 1, ind[:, ::2, ::2], 0), 2, 1), 2, 2)
    exp = exp + np.repeat(np.repeat(np.where(mks[0][:, 0] == 1, ind[:, ::4, ::4], 0), 4, 1), 4, 2)
    assert int(status.abs().max()) == 0 and np.array_equal(dind.cpu().numpy(), exp)
    assert all(torch.equal(a, b_) for a, b_ in zip(dmask, mask))
    assert torch.equal(zq, codec.codebook[dind].permute(0, 3, 1, 2))


def test_telemetry_counters_do_not_change_results(orc):
    """cgic_vq_stats / cgic_decode_stats: device counters of the rare branches (flagged vectors, groups rerun exactly, second
    candidate sets; the self-synchronising decoder's sweeps).  Same outputs with and without them; a clustered codebook of
    near-duplicate rows drives every group onto the exact loop and says so; N(0,1) inputs flag well under 1 % of the vectors."""
    from control_gic_amd.quantize import _vq_forward
    lib = cg._lib.lib()
    rng = np.random.default_rng(3)
    z = torch.from_numpy(rng.standard_normal((8, 4, 64, 64), dtype=np.float32)).to(DEV)
    cb_n = rng.standard_normal((1024, 4), dtype=np.float32)
    cb_c = (rng.standard_normal((64, 4), dtype=np.float32)[rng.integers(0, 64, 1024)] + np.float32(1e-4) * rng.standard_normal((1024, 4), dtype=np.float32)).astype(np.float32)
    N = 8 * 64 * 64
    for cb, clustered in ((cb_n, False), (cb_c, True)):
        w = torch.from_numpy(cb).to(DEV)
        ref = _vq_forward(z, w, 0.25, True, None)
        cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
        assert lib.cgic_vq_stats(cnt.data_ptr()) == 0
        try:
            got = _vq_forward(z, w, 0.25, True, None)
            torch.cuda.synchronize()
        finally:
            lib.cgic_vq_stats(None)
        assert torch.equal(ref[2], got[2]) and torch.equal(ref[0], got[0]) and float(ref[1]) == float(got[1])
        _, _, oidx = orc.vq(z[:1].cpu().numpy(), cb)
        assert np.array_equal(got[2][:4096].cpu().numpy(), oidx)
        c = cnt.cpu().numpy()
        if clustered:
            assert c[1] == N // 64 and c[0] > N // 2, c            # every group reran the exact loop
        else:
            assert c[1] == 0 and 0 < c[0] < N // 100, c
    # decoder sweeps
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(DEV).eval()
    vq.usage_counter.copy_(torch.arange(1024, 0, -1, dtype=torch.float32))
    codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    ind = torch.from_numpy(rng.integers(0, 1024, (6, 64, 64))).to(DEV)
    e16 = torch.from_numpy((rng.random((6, 16, 16)) * 2.6).astype(np.float32)).to(DEV)
    e8 = torch.from_numpy((rng.random((6, 32, 32)) * 2.6).astype(np.float32)).to(DEV)
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(e16, e8)
    comp = codec.compress(ind, mask, mode)
    a = codec.decompress(comp, decoder="throughput")
    cnt = torch.zeros(4, dtype=torch.int32, device=DEV)
    assert lib.cgic_decode_stats(cnt.data_ptr()) == 0
    try:
        b = codec.decompress(comp, decoder="throughput")
        torch.cuda.synchronize()
    finally:
        lib.cgic_decode_stats(None)
    assert torch.equal(a[0], b[0]) and int(b[3].abs().max()) == 0
    c = cnt.cpu().numpy()
    assert c[1] == 6 and 6 <= c[0] <= 6 * 40 and 1 <= c[2] <= 40, c


def filter_margin_telemetry(orc, z, cb):
    """cgic_vq_filter_probe_f32 on z [1,4,hw] / cb [K,4] -> dict of what the exactness argument of the candidate filter budgets
    against what was observed: (a) |f_k - F_k| against 1.8e-6 (ee_k + 2 sum|z_j e_kj|) + the fp16-subnormal floor of 0.02 scaled
    units; (b) how much of the candidate margin thr - f_min the REFERENCE'S winners (argmin set of the oracle's distance row)
    actually used; (c) indices == oracle."""
    import ctypes
    lib = cg._lib.lib()
    zt = torch.from_numpy(z).to(DEV).contiguous()
    w = torch.from_numpy(cb).to(DEV).contiguous()
    _, C, hw = z.shape
    K = cb.shape[0]
    N = hw
    scores = torch.empty((N, K), dtype=torch.float32, device=DEV)
    aux = torch.zeros((N, 6), dtype=torch.float32, device=DEV)
    idx = torch.empty(N, dtype=torch.int64, device=DEV)
    cg._lib.call("cgic_vq_filter_probe_f32", zt.data_ptr(), 1, hw, w.data_ptr(), K, idx.data_ptr(), scores.data_ptr(), aux.data_ptr(), None)
    torch.cuda.synchronize()
    f = scores.cpu().numpy().astype(np.float64)
    ax = aux.cpu().numpy().astype(np.float64)
    zv = z[0].T.astype(np.float32)                                   # [N,4]
    e32 = cb.astype(np.float32)
    ee = (((e32[:, 0] * e32[:, 0] + e32[:, 1] * e32[:, 1]).astype(np.float32) + e32[:, 2] * e32[:, 2]).astype(np.float32) + e32[:, 3] * e32[:, 3]).astype(np.float32)
    z64, e64 = zv.astype(np.float64), e32.astype(np.float64)
    F = ee.astype(np.float64)[None, :] - 2.0 * z64 @ e64.T
    T = ee.astype(np.float64)[None, :] + 2.0 * np.abs(z64) @ np.abs(e64).T
    flagged = ax[:, 3] != 0
    ok = ~flagged
    floor = 0.02 * np.exp2(-ax[:, 4])[:, None]
    over = np.maximum(np.abs(f - F) - floor, 0.0) / (1.8e-6 * T)
    err_ratio = float(over[ok].max()) if ok.any() else 0.0
    plain = float((np.abs(f - F) / T)[ok].max()) if ok.any() else 0.0
    used = 0.0
    got = idx.cpu().numpy()
    bad = 0
    for n in range(N):
        d = orc.vq_distances(zv[n], e32)
        win = np.flatnonzero(d == d.min())
        bad += int(got[n] != win[0])
        if ok[n]:
            span = ax[n, 2] - f[n].min()
            if span > 0:
                used = max(used, float((f[n, win] - f[n].min()).max() / span))
    return {"vectors": N, "flagged": int(flagged.sum()), "max_err_over_budget": err_ratio, "max_err_over_T": plain,
            "max_margin_used_by_reference_winners": used, "wrong_indices": bad}


def telemetry_families(rng, hw=1024):
    cb_n = rng.standard_normal((1024, 4)).astype(np.float32)
    u = lambda shape: ((rng.random(shape) * 2 - 1) / 1024).astype(np.float32)
    fam = {}
    fam["normal"] = (rng.standard_normal((1, 4, hw)).astype(np.float32), cb_n)
    fam["scale_1e-3"] = (fam["normal"][0] * np.float32(1e-3), cb_n * np.float32(1e-3))
    fam["scale_37"] = (fam["normal"][0] * np.float32(37), cb_n * np.float32(37))
    fam["init_uniform"] = (u((1, 4, hw)), u((1024, 4)))
    fam["z_small_next_to_codebook"] = (fam["normal"][0] * np.float32(1e-3), cb_n)
    fam["z_large_next_to_codebook"] = (fam["normal"][0] * np.float32(50), cb_n)
    on = cb_n[rng.integers(0, 1024, hw)].T[None].copy() + np.float32(1e-6) * rng.standard_normal((1, 4, hw)).astype(np.float32)
    fam["on_code"] = (on.astype(np.float32), cb_n)
    dup = cb_n.copy()
    dup[512:] = dup[:512]
    dup[512:, 1] = np.nextafter(dup[512:, 1], np.float32(4))          # every code has a twin one ulp away, 16 tiles further on
    fam["twin_rows_one_ulp_apart"] = ((dup[rng.integers(0, 512, hw)].T[None] + np.float32(0.05) * rng.standard_normal((1, 4, hw))).astype(np.float32), dup)
    spread = (cb_n * np.exp(rng.uniform(-6, 2, (1024, 1)))).astype(np.float32)
    fam["codebook_of_mixed_norms"] = (rng.standard_normal((1, 4, hw)).astype(np.float32), spread)
    return fam


def test_filter_margin_telemetry(orc):
    """the exactness of the fp16 candidate filter rested on an analytic margin plus sampling; here the kernel
    itself (cgic_vq_filter_probe_f32: the production body, instantiated to write its scores and thresholds out) is held against
    the budget on the stress families: the observed |f - F| stays under HALF of 1.8e-6 T_k (+ the subnormal floor), the
    reference's winners use less than half of the candidate margin, and every index equals the oracle's."""
    rng = np.random.default_rng(12)
    worst = {}
    for name, (z, cb) in telemetry_families(rng).items():
        t = filter_margin_telemetry(orc, z, cb)
        worst[name] = t
        assert t["wrong_indices"] == 0, (name, t)
        assert t["max_err_over_budget"] <= 0.5, (name, t)
        assert t["max_margin_used_by_reference_winners"] <= 0.5, (name, t)
    assert worst["normal"]["flagged"] < 16
