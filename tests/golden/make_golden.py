#!/usr/bin/env python3
"""Generate the golden fixtures in tests/golden/ by running the REAL reference.

Runs only in the build container (needs /root/reference; the reference never
travels to the GPU box).  The reference is imported read-only with the
harness-side shims of SURVEY.md Appendix A; nothing from it is copied here --
the outputs below are data (inputs + the reference's outputs on them).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py [--skip-model]

Every fixture is also checked against the oracle (oracle/cgic_oracle.c) on the
spot; a mismatch aborts, so a committed fixture set implies "oracle pinned".
"""
import argparse
import io
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)

# ---- shims -------------------------------------------
pl = types.ModuleType("pytorch_lightning")
pl.LightningModule = torch.nn.Module
pl.LightningDataModule = object
sys.modules["pytorch_lightning"] = pl
tv = MagicMock()
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tv.transforms
sys.modules["torchvision.transforms.functional"] = tv.transforms.functional
sys.modules["torchvision.utils"] = tv.utils
torch.nn.Module.cuda = lambda self, device=None: self

from CGIC.modules.vqvae.quantize import VectorQuantize2  # noqa: E402
from CGIC.modules.vqvae.RouterTriple import TripleGrainFixedEntropyRouter  # noqa: E402
from CGIC.tools.indices_coding import HuffmanCoding  # noqa: E402
from CGIC.tools.mask_coding import BinaryCoding  # noqa: E402

from oracle import cgic_oracle as orc  # noqa: E402

torch.set_num_threads(8)

RATIOS = [(0.1, 0.8), (0.1, 0.4), (0.7, 0.3), (0.3, 0.7), (0.0, 0.4), (0.4, 0.0), (1.0, 0.0),
          (0.0, 1.0), (0.0, 0.0), (0.25, 0.25), (0.5, 0.5), (0.0, 0.5), (0.5, 0.0), (0.301, 0.599)]


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print(f"  wrote {name}.npz ({os.path.getsize(path)} bytes)")


def check(cond, msg):
    if not cond:
        raise SystemExit("ORACLE MISMATCH: " + msg)


def freq_tables():
    g = np.random.default_rng(7)
    zipf = np.floor(2.0e6 / (1 + np.arange(1024)) ** 1.1).astype(np.int64)
    g.shuffle(zipf)
    big = zipf.copy()
    big[:5] = [2 ** 24, 2 ** 24 + 2, 2 ** 25, 3 * 2 ** 24, 2 ** 24]
    ties = g.integers(0, 6, 1024).astype(np.int64)
    return {"zeros": np.zeros(1024, np.int64), "zipf": zipf, "big": big, "ties": ties}


def as_param_dict(freq):
    # same container type the reference hands HuffmanCoding (inference.py:137-139)
    return torch.nn.ParameterDict({str(i): torch.nn.Parameter(torch.tensor([float(v)]))
                                   for i, v in enumerate(freq)}).requires_grad_(False)


def codes_to_arrays(h):
    n = len(h.codes)
    lens = np.array([len(h.codes[i]) for i in range(n)], np.int32)
    words = max(1, (int(lens.max()) + 31) // 32)
    code = np.zeros((n, words), np.uint32)
    for i in range(n):
        for b, ch in enumerate(h.codes[i]):
            if ch == "1":
                code[i, b // 32] |= np.uint32(1 << (31 - b % 32))
    return lens, code


def ref_compress_bytes(coder, t):
    with tempfile.TemporaryDirectory() as d:
        p = coder.compress(t, os.path.join(d, "s.bin"))
        data = open(p, "rb").read()
        dec = coder.decompress_string(p)
    return data, dec


# ---- A. VectorQuantize2 ------------------------------------------------------
def gen_vq():
    print("VQ (quantize.py:69-97)")
    out = {}
    cases = []
    g = torch.Generator().manual_seed(11)
    torch.manual_seed(11)
    vq = VectorQuantize2(1024, 4, beta=0.25).eval()
    init_cb = vq.embedding.weight.data.clone()  # U(+-1/1024), quantize.py:26
    cb_n = torch.randn(1024, 4, generator=g)
    cb_dup = cb_n.clone()
    cb_dup[512:] = cb_dup[:512]           # exact duplicate rows -> exact ties
    cases.append(("normal_b2_32", torch.randn(2, 4, 32, 32, generator=g), cb_n))
    cases.append(("normal_n1", torch.randn(1, 4, 1, 1, generator=g), cb_n))
    cases.append(("normal_odd", torch.randn(1, 4, 3, 7, generator=g), cb_n))
    cases.append(("normal_b1_64", torch.randn(1, 4, 64, 64, generator=g), cb_n))
    cases.append(("init_small", (torch.rand(1, 4, 64, 64, generator=g) * 2 - 1) / 1024, init_cb))
    cases.append(("init_mixed", torch.randn(2, 4, 16, 48, generator=g) * 0.01, init_cb))
    cases.append(("dup_rows", torch.randn(1, 4, 32, 32, generator=g), cb_dup))
    cases.append(("codes_as_z", cb_n[:256].t().reshape(1, 4, 16, 16).contiguous(), cb_n))
    zq_grid = (torch.randint(-8, 9, (1, 4, 32, 32), generator=g).float() / 8)
    cb_grid = (torch.randint(-8, 9, (1024, 4), generator=g).float() / 8)  # many exact ties
    cases.append(("lattice_ties", zq_grid, cb_grid))
    for name, z, cb in cases:
        vq.embedding.weight.data.copy_(cb)
        with torch.no_grad():
            zq, loss, idx = vq(z)
        ozq, oloss, oidx = orc.vq(z.numpy(), cb.numpy())
        nmis = int((oidx != idx.numpy()).sum())
        check(nmis == 0, f"vq {name}: {nmis} index mismatches")
        check(np.array_equal(ozq, zq.numpy()), f"vq {name}: z_q not bit-equal")
        check(abs(float(oloss) - float(loss)) <= 1e-6 * abs(float(loss)) + 1e-12, f"vq {name}: loss")
        out[name + "_z"] = z.numpy()
        out[name + "_cb"] = cb.numpy()
        out[name + "_idx"] = idx.numpy().astype(np.int16)
        out[name + "_zq"] = zq.numpy()
        out[name + "_loss"] = np.float32(loss.item())
        print(f"  {name}: N={idx.numel()} ok (loss {loss.item():.6g})")
    # training-mode usage counter (quantize.py:79-81)
    vq.embedding.weight.data.copy_(cb_n)
    vq.train()
    z = torch.randn(1, 4, 8, 8, generator=g)
    with torch.no_grad():
        vq(z)
        vq(z)
    cnt = np.array([vq.embedding_counter[str(i)].item() for i in range(1024)], np.float32)
    hist = np.zeros(1024, np.int64)
    orc.vq(z.numpy(), cb_n.numpy(), hist=hist)
    check(np.array_equal(cnt, (2 * hist).astype(np.float32)), "vq counter")
    out["counter_z"] = z.numpy()
    out["counter_cnt"] = cnt
    save("vq", **out)


# ---- C. Router ----------------------------------------------------------------
def gen_router():
    print("Router (RouterTriple.py:15-95)")
    out = {"ratios": np.array(RATIOS, np.float64)}
    g = torch.Generator().manual_seed(5)
    shapes = {"b1_16x16": (1, 16, 16), "b2_16x16": (2, 16, 16), "b1_4x6": (1, 4, 6), "b3_8x12": (3, 8, 12)}
    for sname, (B, h16, w16) in shapes.items():
        e16 = torch.rand(B, h16, w16, generator=g) * 2.6
        e8 = torch.rand(B, 2 * h16, 2 * w16, generator=g) * 2.6
        # inject exact ties around typical thresholds
        f16, f8 = e16.flatten(), e8.flatten()
        f16[torch.randperm(f16.numel(), generator=g)[: max(2, f16.numel() // 6)]] = f16.median()
        f8[torch.randperm(f8.numel(), generator=g)[: max(2, f8.numel() // 6)]] = f8.median()
        f8[:3] = 0.0
        out[f"{sname}_e16"] = e16.numpy()
        out[f"{sname}_e8"] = e8.numpy()
        for ri, (c, m) in enumerate(RATIOS):
            r = TripleGrainFixedEntropyRouter(c, m)
            mask, gate, ratios, mode = r(e16, e8)
            omc, omm, omf, ogate, omode = orc.router(e16.numpy(), e8.numpy(), c, m)
            check(omode == mode, f"router {sname} {c},{m}: mode {omode} != {mode}")
            for a, b_, nm in zip((omc, omm, omf), mask, "cmf"):
                check(np.array_equal(a, b_.numpy()), f"router {sname} {c},{m}: mask_{nm}")
            check(np.array_equal(ogate, gate.numpy()), f"router {sname} {c},{m}: gate")
            out[f"{sname}_r{ri}_mc"] = np.packbits(mask[0].numpy().astype(np.uint8))
            out[f"{sname}_r{ri}_mm"] = np.packbits(mask[1].numpy().astype(np.uint8))
            out[f"{sname}_r{ri}_mf"] = np.packbits(mask[2].numpy().astype(np.uint8))
            out[f"{sname}_r{ri}_mode"] = np.int32(mode)
            # per-image routing == looping the reference at B=1
            if B > 1:
                pm = [TripleGrainFixedEntropyRouter(c, m)(e16[b:b + 1], e8[b:b + 1])[0] for b in range(B)]
                pmc, pmm, pmf, _, _ = orc.router(e16.numpy(), e8.numpy(), c, m, per_image=True)
                for b in range(B):
                    check(np.array_equal(pmc[b:b + 1], pm[b][0].numpy()), "router per-image c")
                    check(np.array_equal(pmm[b:b + 1], pm[b][1].numpy()), "router per-image m")
                    check(np.array_equal(pmf[b:b + 1], pm[b][2].numpy()), "router per-image f")
        print(f"  {sname}: {len(RATIOS)} ratio pairs ok")
    # constant entropy map: every value ties, nothing is < threshold -> all fine
    e16 = torch.full((1, 16, 16), 3.351e-4)
    e8 = torch.full((1, 32, 32), 3.351e-4)
    mask, gate, _, mode = TripleGrainFixedEntropyRouter(0.1, 0.8)(e16, e8)
    omc, omm, omf, _, _ = orc.router(e16.numpy(), e8.numpy(), 0.1, 0.8)
    check(np.array_equal(omf, mask[2].numpy()) and int(mask[2].sum()) == 64 * 64, "router const")
    save("router", **out)


# ---- D/E/F. Huffman + binary coders --------------------------------------------
def gen_coders():
    print("Coders (indices_coding.py, mask_coding.py)")
    out = {}
    g = np.random.default_rng(3)
    for name, freq in freq_tables().items():
        h = HuffmanCoding(as_param_dict(freq))
        lens, code = codes_to_arrays(h)
        order = np.array([int(k) for k in as_param_dict(freq).keys()], np.int32)
        check(np.array_equal(order, orc.param_dict_order(1024)), "ParameterDict key order")
        t = orc.HuffmanTable(freq)
        check(np.array_equal(t.len, lens), f"huffman {name}: lengths")
        check(t.words == code.shape[1] and np.array_equal(t.code, code), f"huffman {name}: codes")
        out[f"{name}_freq"] = freq
        out[f"{name}_order"] = order
        # a mapping in natural insertion order (what a plain dict would give)
        class _V:
            def __init__(self, v): self.v = v
            def item(self): return self.v
        hn = HuffmanCoding({str(i): _V(float(v)) for i, v in enumerate(freq)})
        ln, cn = codes_to_arrays(hn)
        tn = orc.HuffmanTable(freq, order="natural")
        check(np.array_equal(tn.len, ln) and np.array_equal(tn.code, cn), f"huffman {name}: natural order")
        out[f"{name}_natural_len"] = ln
        out[f"{name}_len"] = lens
        out[f"{name}_code"] = code
        # symbol streams: empty, 1, short, 4096 skewed to frequent symbols
        p = (freq + 1.0) / (freq + 1.0).sum()
        streams = [np.zeros(0, np.int64), np.array([5], np.int64), g.integers(0, 1024, 25),
                   g.choice(1024, 821, p=p), g.choice(1024, 4096, p=p), np.arange(1024)]
        for si, s in enumerate(streams):
            data, dec = ref_compress_bytes(h, torch.tensor(s, dtype=torch.int64))
            od = orc.encode(t, s)
            check(od == data, f"huffman {name} stream {si}: bytes differ ({len(od)} vs {len(data)})")
            odec = orc.decode(t, data)
            check((dec is None and odec is None) or list(odec) == dec, f"huffman {name} stream {si}: decode")
            out[f"{name}_s{si}_sym"] = np.asarray(s, np.int16)
            out[f"{name}_s{si}_bytes"] = np.frombuffer(data, np.uint8)
        print(f"  huffman {name}: maxlen {lens.max()} minlen {lens.min()} ok")
    b = BinaryCoding()
    bt = orc.HuffmanTable.binary()
    for n in (0, 1, 7, 8, 256, 1024, 2304):
        m = g.integers(0, 2, n).astype(np.int32)
        data, dec = ref_compress_bytes(b, torch.tensor(m, dtype=torch.int32))
        check(orc.encode(bt, m) == data, f"binary n={n}: bytes")
        odec = orc.decode(bt, data)
        check((dec is None and odec is None) or list(odec) == dec, f"binary n={n}: decode")
        out[f"binary_{n}_mask"] = m.astype(np.uint8)
        out[f"binary_{n}_bytes"] = np.frombuffer(data, np.uint8)
    print("  binary ok")
    save("coders", **out)


# ---- C'. Entropy + G. compress() glue via the real CGIC model --------------------
def build_model(seed=0):
    from CGIC.models.model import CGIC
    params = yaml.safe_load(open(os.path.join(REF, "configs/config_inference.yaml")))["model"]["params"]
    params["ckpt_path"] = None
    params["lossconfig"] = None
    torch.manual_seed(seed)
    _stdout = sys.stdout
    sys.stdout = io.StringIO()
    try:
        model = CGIC(**params).eval()
    finally:
        sys.stdout = _stdout
    return model


def gen_entropy():
    print("Entropy (model.py:433-483)")
    from CGIC.models.model import Entropy
    out = {}
    g = torch.Generator().manual_seed(9)
    xs = {"rand_64x96": torch.rand(2, 3, 64, 96, generator=g),
          "u8_48x80": torch.randint(0, 256, (1, 3, 48, 80), generator=g).float() / 255,
          "smooth_64x64": (torch.linspace(0, 1, 64)[None, None, :, None] * torch.ones(1, 3, 64, 64)
                           + 0.02 * torch.rand(1, 3, 64, 64, generator=g)).clamp(0, 1),
          "const_32x32": torch.full((1, 3, 32, 32), 0.5),
          "signed_32x32": torch.rand(1, 3, 32, 32, generator=g) * 2 - 1}
    bins = torch.linspace(-1, 1, 32)
    check(np.array_equal(bins.numpy(), orc.linspace_bins()), "linspace bins")
    out["bins"] = bins.numpy()
    for name, x in xs.items():
        out[name + "_x"] = x.numpy()
        for p in (8, 16):
            e = Entropy(p)(x)
            oe = orc.entropy(x.numpy(), p)
            err = float(np.abs(oe - e.numpy()).max())
            check(err < 2e-5, f"entropy {name} p{p}: max abs err {err}")
            out[f"{name}_e{p}"] = e.numpy()
            print(f"  {name} p{p}: max|oracle-ref| = {err:.2e}, range [{e.min():.4f},{e.max():.4f}]")
    save("entropy", **out)


def gen_compress():
    print("compress() glue (model.py:206-401) through the real CGIC model, CPU")
    model = build_model(0)
    tabs = freq_tables()
    out = {"ratios": np.array(RATIOS, np.float64)}
    torch.manual_seed(0)
    x = torch.rand(1, 3, 256, 256)            # config 1 input
    out["x_seed0_sum"] = np.float64(x.double().sum().item())
    cb = model.quantize.embedding.weight.data.clone()
    out["codebook"] = cb.numpy()
    bcoder = BinaryCoding()
    first = True
    for tname in ("zipf", "zeros"):
        freq = tabs[tname]
        for i, v in enumerate(freq):
            model.quantize.embedding_counter[str(i)].data.fill_(float(v))
        hcoder = HuffmanCoding(model.quantize.embedding_counter)
        htab = orc.HuffmanTable(freq)
        ratios = RATIOS if tname == "zipf" else RATIOS[:1]
        for ri, (c, m) in enumerate(ratios):
            model.encoder.router_config["params"]["coarse_grain_ratio"] = c
            model.encoder.router_config["params"]["medium_grain_ratio"] = m
            with torch.no_grad(), tempfile.TemporaryDirectory() as d:
                quant, _, _, grain_mask, ind, _, mode = model.encode(x)
                if first:
                    e8 = model.entropy_calculation_p8(x)
                    e16 = model.entropy_calculation_p16(x)
                    h = model.quant_conv(model.encoder(x, e16, e8)["h"])
                    out["e8"], out["e16"], out["z"] = e8.numpy(), e16.numpy(), h.numpy()
                    first = False
                # capture quant_decompress by hooking decode (the decoder itself is out of scope)
                cap = {}
                orig = model.decode
                model.decode = lambda q, mk: cap.update(q=q.clone(), mk=[t.clone() for t in mk]) or torch.zeros(1)
                try:
                    _, bpp, _ = model.compress(x, d, hcoder, bcoder, False)
                finally:
                    model.decode = orig
                files = {n: open(os.path.join(d, n + ".bin"), "rb").read()
                         for n in orc.STREAM_NAMES if os.path.exists(os.path.join(d, n + ".bin"))}
            key = f"{tname}_r{ri}"
            ind2 = ind.view(64, 64).numpy()
            mc, mm, mf = (t.numpy().reshape(t.shape[-2], t.shape[-1]) for t in grain_mask)
            # oracle: router on the captured entropies, VQ on the captured latent
            omc, omm, omf, _, omode = orc.router(out["e16"], out["e8"], c, m)
            check(omode == mode and np.array_equal(omc[0, 0], mc) and np.array_equal(omm[0, 0], mm)
                  and np.array_equal(omf[0, 0], mf), f"compress {key}: router")
            # NB: the latent depends on the masks (encoder merge), so re-run VQ per ratio on this ratio's h
            with torch.no_grad():
                hh = model.quant_conv(model.encoder(x, torch.from_numpy(out["e16"]), torch.from_numpy(out["e8"]))["h"])
            _, _, oidx = orc.vq(hh.numpy(), cb.numpy())
            check(np.array_equal(oidx.reshape(64, 64), ind2), f"compress {key}: vq indices")
            streams = orc.compress_image(ind2, mc, mm, mf, mode, htab)
            on = orc.mode_streams(mode)
            for si, n in enumerate(orc.STREAM_NAMES):
                if on[si]:
                    check(streams[n] == files[n], f"compress {key}: {n}.bin differs")
                    out[f"{key}_{n}"] = np.frombuffer(files[n], np.uint8)
            obpp = sum(len(streams[n]) for n in streams) * 8 / (256 * 256)
            check(obpp == bpp, f"compress {key}: bpp {obpp} vs {bpp}")
            oind, omc2, omm2, omf2 = orc.decompress_image(streams, mode, 64, 64, htab)
            oq = orc.gather(oind, cb.numpy())
            check(np.array_equal(oq, cap["q"].numpy()), f"compress {key}: quant_decompress")
            for a, b_ in zip((omc2, omm2, omf2), cap["mk"]):
                check(np.array_equal(a, b_.numpy().reshape(a.shape)), f"compress {key}: decoded masks")
            out[f"{key}_z"] = hh.numpy()
            out[f"{key}_ind"] = ind2.astype(np.int16)
            out[f"{key}_mode"] = np.int32(mode)
            out[f"{key}_bpp"] = np.float64(bpp)
            out[f"{key}_mc"] = np.packbits(mc.astype(np.uint8))
            out[f"{key}_mm"] = np.packbits(mm.astype(np.uint8))
            out[f"{key}_mf"] = np.packbits(mf.astype(np.uint8))
            out[f"{key}_qdec_ind"] = oind.astype(np.int16)
            print(f"  {key} ratio ({c},{m}) mode {mode} sums [{mc.sum()},{mm.sum()},{mf.sum()}] bpp {bpp:.5f} "
                  f"files {[len(files[n]) for n in files]} ok")
    save("compress_cfg1", **out)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--skip-model", action="store_true", help="skip the fixtures that need the full CGIC model")
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    todo = [gen_vq, gen_router, gen_coders, gen_entropy] + ([] if a.skip_model else [gen_compress])
    for fn in todo:
        if a.only and a.only not in fn.__name__:
            continue
        fn()
    print("all fixtures written; oracle agrees with the reference on every one")
