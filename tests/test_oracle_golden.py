"""CPU: the oracle (oracle/cgic_oracle.c) against the golden vectors produced by the real
reference (tests/golden/make_golden.py).  This is what "oracle pinned" means."""
import numpy as np
import pytest

from conftest import unpack_mask

VQ_CASES = ["normal_b2_32", "normal_n1", "normal_odd", "normal_b1_64", "init_small", "init_mixed", "dup_rows",
            "codes_as_z", "lattice_ties"]


@pytest.mark.parametrize("case", VQ_CASES)
def test_vq_matches_reference(orc, golden, case):
    g = golden("vq")
    zq, loss, idx = orc.vq(g[case + "_z"], g[case + "_cb"])
    assert np.array_equal(idx, g[case + "_idx"].astype(np.int64).ravel())        # bit-exact indices
    assert np.array_equal(zq, g[case + "_zq"])                                   # bit-exact z + (e - z)
    assert abs(float(loss) - float(g[case + "_loss"])) <= 1e-6 * abs(float(g[case + "_loss"])) + 1e-12


def test_vq_usage_counter(orc, golden):
    g = golden("vq")
    hist = np.zeros(1024, np.int64)
    orc.vq(g["counter_z"], g["normal_b2_32_cb"], hist=hist)
    assert np.array_equal((2 * hist).astype(np.float32), g["counter_cnt"])      # two training forwards


@pytest.mark.parametrize("shape", ["b1_16x16", "b2_16x16", "b1_4x6", "b3_8x12"])
def test_router_matches_reference(orc, golden, shape):
    g = golden("router")
    e16, e8 = g[shape + "_e16"], g[shape + "_e8"]
    B, h16, w16 = e16.shape
    for ri, (c, m) in enumerate(g["ratios"]):
        mc, mm, mf, gate, mode = orc.router(e16, e8, float(c), float(m))
        assert mode == int(g[f"{shape}_r{ri}_mode"])
        assert np.array_equal(mc, unpack_mask(g[f"{shape}_r{ri}_mc"], mc.shape))
        assert np.array_equal(mm, unpack_mask(g[f"{shape}_r{ri}_mm"], mm.shape))
        assert np.array_equal(mf, unpack_mask(g[f"{shape}_r{ri}_mf"], mf.shape))
        assert np.array_equal(gate[..., :4 * w16], np.repeat(np.repeat(mc, 4, 2), 4, 3).astype(np.float32))


def test_router_float64_mode_quirks(orc):
    # 1 - 0.7 - 0.3 = 5.55e-17 != 0 -> mode 0; 1 - 0.3 - 0.7 == 0.0 -> mode 3
    assert orc.router_mode(0.7, 0.3) == 0 and orc.router_mode(0.3, 0.7) == 3
    assert [orc.router_mode(c, m) for c, m in ((0, .4), (.4, 0), (1, 0), (0, 1), (0, 0))] == [1, 2, 4, 5, 6]


@pytest.mark.parametrize("name", ["rand_64x96", "u8_48x80", "smooth_64x64", "const_32x32", "signed_32x32"])
def test_entropy_matches_reference(orc, golden, name):
    g = golden("entropy")
    assert np.array_equal(orc.linspace_bins(), g["bins"])
    for p in (8, 16):
        e = orc.entropy(g[name + "_x"], p, bins=g["bins"])
        assert np.abs(e - g[f"{name}_e{p}"]).max() < 2e-6        # exp/log/sum-order: tolerance, not bit-exact


@pytest.mark.parametrize("name", ["zeros", "zipf", "big", "ties"])
def test_huffman_table_and_streams(orc, golden, name):
    g = golden("coders")
    assert np.array_equal(g[name + "_order"], orc.param_dict_order(1024))
    t = orc.HuffmanTable(g[name + "_freq"])                      # ParameterDict (string-sorted) push order
    assert np.array_equal(t.len, g[name + "_len"]) and np.array_equal(t.code, g[name + "_code"])
    tn = orc.HuffmanTable(g[name + "_freq"], order="natural")
    assert np.array_equal(tn.len, g[name + "_natural_len"])
    for si in range(6):
        sym = g[f"{name}_s{si}_sym"].astype(np.int64)
        data = g[f"{name}_s{si}_bytes"].tobytes()
        assert orc.encode(t, sym) == data
        dec = orc.decode(t, data)
        assert (dec is None and len(sym) == 0) or np.array_equal(dec, sym)


def test_binary_coder(orc, golden):
    g = golden("coders")
    bt = orc.HuffmanTable.binary()
    for n in (0, 1, 7, 8, 256, 1024, 2304):
        m, data = g[f"binary_{n}_mask"].astype(np.int32), g[f"binary_{n}_bytes"].tobytes()
        assert orc.encode(bt, m) == data
        assert len(data) == (0 if n == 0 else n // 8 + 2)        # header + floor(n/8)+1 payload bytes
        dec = orc.decode(bt, data)
        assert (dec is None and n == 0) or np.array_equal(dec, m)


def _compress_keys(g):
    return sorted({k[:-5] for k in g if k.endswith("_mode")})


def test_compress_glue_all_modes(orc, golden):
    """config 1: the reference's CGIC.compress on torch.rand(1,3,256,256) (seed 0), every ratio pair"""
    g = golden("compress_cfg1")
    cb = g["codebook"]
    tabs = {"zipf": orc.HuffmanTable(golden("coders")["zipf_freq"]), "zeros": orc.HuffmanTable(np.zeros(1024, np.int64))}
    seen_modes = set()
    for key in _compress_keys(g):
        tname, ri = key.split("_r")
        mode = int(g[key + "_mode"])
        seen_modes.add(mode)
        c, m = g["ratios"][int(ri)]
        # router on the captured entropies, VQ on the captured latent
        mc, mm, mf, _, omode = orc.router(g["e16"], g["e8"], float(c), float(m))
        assert omode == mode
        assert np.array_equal(mc[0, 0], unpack_mask(g[key + "_mc"], (16, 16)))
        assert np.array_equal(mm[0, 0], unpack_mask(g[key + "_mm"], (32, 32)))
        assert np.array_equal(mf[0, 0], unpack_mask(g[key + "_mf"], (64, 64)))
        _, _, idx = orc.vq(g[key + "_z"], cb)
        ind = g[key + "_ind"].astype(np.int64)
        assert np.array_equal(idx.reshape(64, 64), ind)
        streams = orc.compress_image(ind, mc[0, 0], mm[0, 0], mf[0, 0], mode, tabs[tname])
        on = orc.mode_streams(mode)
        assert set(streams) == {n for n, o in zip(orc.STREAM_NAMES, on) if o}
        for n in streams:
            assert streams[n] == g[f"{key}_{n}"].tobytes(), f"{key}: {n}.bin"
        assert sum(map(len, streams.values())) * 8 / 65536 == float(g[key + "_bpp"])
        dind, dmc, dmm, dmf = orc.decompress_image(streams, mode, 64, 64, tabs[tname])
        assert np.array_equal(dind, g[key + "_qdec_ind"].astype(np.int64))
        assert np.array_equal(orc.gather(dind, cb)[0], cb[dind].transpose(2, 0, 1))
    assert seen_modes == set(range(7))


def test_first_entropy_map_of_config1(orc, golden):
    """config 1 input is torch.rand under manual_seed(0); regenerate it and check the captured maps"""
    import torch
    g = golden("compress_cfg1")
    torch.manual_seed(0)
    x = torch.rand(1, 3, 256, 256)
    if abs(float(x.double().sum()) - float(g["x_seed0_sum"])) > 1e-9:
        pytest.skip("torch CPU generator differs from the one that made the fixture")
    assert np.abs(orc.entropy(x.numpy(), 8) - g["e8"]).max() < 2e-6
    assert np.abs(orc.entropy(x.numpy(), 16) - g["e16"]).max() < 2e-6


FAMILIES = ["noise8", "smooth8", "flat_edges", "blocky8"]


@pytest.mark.parametrize("name", FAMILIES)
def test_entropy_torch_restatement_pinned_by_reference_fixture(orc, golden, name):
    """oracle/entropy_torch.py (the reference's arithmetic on torch's CPU operators) against what the REAL Entropy class
    produced for 8-bit tie-heavy images (tests/golden/make_golden_ties.py): bit for bit on the torch build that made the
    fixture; the oracle's router then gives the reference's masks"""
    import platform
    import torch
    from oracle import entropy_torch as et
    g = golden("ties")
    x = g[name + "_u8"].astype(np.float32) / 255.0
    xt = torch.from_numpy(x)
    e8, e16 = et.entropy_map(xt, 8).numpy(), et.entropy_map(xt, 16).numpy()
    same_build = str(g["torch_version"]) == torch.__version__ and str(g["machine"]) == platform.machine()
    if same_build:
        assert np.array_equal(e8, g[name + "_e8"]) and np.array_equal(e16, g[name + "_e16"])
    assert np.abs(e8 - g[name + "_e8"]).max() < 2e-6 and np.abs(e16 - g[name + "_e16"]).max() < 2e-6
    for b in range(x.shape[0]):
        mc, mm, mf, _, mode = orc.router(g[name + "_e16"][b:b + 1], g[name + "_e8"][b:b + 1], 0.1, 0.8)
        assert mode == 0
        for k, m in zip("cmf", (mc, mm, mf)):
            assert np.array_equal(np.packbits(m.astype(np.uint8).reshape(-1)), g[f"{name}_{b}_m{k}"])
    # the plain-C oracle agrees to its tolerance (libm exp/log, sequential sums), not to the bit
    assert np.abs(orc.entropy(x, 8) - g[name + "_e8"]).max() < 2e-6
    assert np.abs(orc.entropy(x, 16) - g[name + "_e16"]).max() < 2e-6


@pytest.mark.parametrize("name", FAMILIES)
def test_reference_arithmetic_entropy_port_vs_reference_fixture(orc, golden, name):
    """cgic_oracle_entropy_ref (torch's CPU operation sequence and summation order, exp / log correctly rounded: the CPU
    restatement of the GPU's opt-in cgic_entropy_maps_ref_f32) against the REAL Entropy class's maps: most values to the bit,
    the rest within 5e-7 (torch's exp / log are MKL's), and the reference's masks exactly"""
    g = golden("ties")
    x = g[name + "_u8"].astype(np.float32) / 255.0
    a8, a16 = orc.entropy_ref(x, 8), orc.entropy_ref(x, 16)
    assert np.abs(a8 - g[name + "_e8"]).max() < 5e-7 and np.abs(a16 - g[name + "_e16"]).max() < 5e-7
    assert (a8 == g[name + "_e8"]).mean() > 0.85 and (a16 == g[name + "_e16"]).mean() > 0.85
    for b in range(x.shape[0]):
        mc, mm, mf, _, _ = orc.router(a16[b:b + 1], a8[b:b + 1], 0.1, 0.8)
        for k, m in zip("cmf", (mc, mm, mf)):
            assert np.array_equal(np.packbits(m.astype(np.uint8).reshape(-1)), g[f"{name}_{b}_m{k}"])


BIG_TIES = ["noise8_256", "smooth8_256", "flat_edges_256", "blocky8_256", "smooth8_768"]


@pytest.mark.parametrize("name", BIG_TIES)
def test_reference_masks_of_realistic_bands_follow_from_the_correctly_rounded_port(orc, golden, name):
    """round 6: two 256x256 images per tie-heavy family + one smooth 768x768 tile (bands of up to ~200 patches around the coarse
    threshold), masks of the REAL router on the REAL Entropy maps, per image and over the flattened batch (RouterTriple.py:21,
    40,52,63): the correctly rounded restatement (cgic_oracle_entropy_ref == the GPU's reference-order arithmetic bit for bit)
    gives exactly these masks -- what tests/test_gpu_parity.py then demands of the GPU's pixels -> masks path"""
    g = golden("ties")
    x = g[name + "_u8"].astype(np.float32) / 255.0
    a8, a16 = orc.entropy_ref(x, 8), orc.entropy_ref(x, 16)
    nb = x.shape[0]
    for b in list(range(nb)) + (["batch"] if nb > 1 else []):
        sl = slice(0, nb) if b == "batch" else slice(b, b + 1)
        mc, mm, mf, _, mode = orc.router(a16[sl], a8[sl], 0.1, 0.8)
        assert mode == 0
        for k, m in zip("cmf", (mc, mm, mf)):
            assert np.array_equal(np.packbits(m.astype(np.uint8).reshape(-1)), g[f"{name}_{b}_m{k}"]), (name, b, k)


def test_fast_division_by_sigma_is_the_ieee_quotient(orc):
    """cgic_entropy_dev.h: div_by_sigma001 (x * 100 corrected once by the exact remainder) == x / 0.01f, bit for bit -- what the
    reference computes at model.py:454.  Sampled here (every 257th magnitude of [2^-100, 8] + whole binades around the bin
    pitch + the edges); tools/check_fast_div.py is the exhaustive run (0 mismatches in 2 x 864 026 625 values).  Below
    2^-100 the quotients may differ, but their squares -- all the caller uses -- are 0 either way; that is checked too."""
    import ctypes, struct
    f = orc.lib().cgic_oracle_check_fast_div
    f.restype = ctypes.c_long
    f.argtypes = [ctypes.c_uint, ctypes.c_uint, ctypes.c_uint, ctypes.POINTER(ctypes.c_uint)]
    bits = lambda v: struct.unpack("<I", struct.pack("<f", v))[0]
    lo, hi = bits(2.0 ** -100), bits(8.0) + 1
    first = ctypes.c_uint(0)
    assert f(lo, hi, 257, ctypes.byref(first)) == 0, hex(first.value)
    for a, b in ((2.0 ** -5, 2.0 ** -3), (2.0 ** -100, 2.0 ** -99), (4.0, 8.0)):
        assert f(bits(a), bits(b) + 1, 1, ctypes.byref(first)) == 0, hex(first.value)
    x = np.concatenate([np.float32(2.0) ** -np.arange(101, 150, dtype=np.float32), [np.float32(0.0)]]).astype(np.float32)
    q = (x * np.float32(100.0)).astype(np.float32)
    assert np.all((q * q).astype(np.float32) == 0) and np.all(((x / np.float32(0.01)) ** 2).astype(np.float32) == 0)
