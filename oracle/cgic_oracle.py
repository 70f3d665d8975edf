"""numpy/ctypes front end of the CPU oracle (oracle/cgic_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg as the checker.  The product package never imports
this module.  Parity status: pinned against the reference's own outputs
(tests/golden/make_golden.py); see the header of cgic_oracle.c.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libcgic_oracle.so")


def build(force=False):
    src = os.path.join(_HERE, "cgic_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-s", "-B" if force else "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        _lib = C.CDLL(_SO)
        _lib.cgic_oracle_encode.restype = C.c_long
        _lib.cgic_oracle_decode.restype = C.c_long
    return _lib


def _p(a, t):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def vq(z, codebook, beta=0.25, legacy=True, hist=None):
    """quantize.py:69-97.  z [B,C,h,w] f32, codebook [K,C] f32 ->
    (z_q [B,C,h,w] f32, loss f32 scalar, indices [B*h*w] int64)"""
    z = np.ascontiguousarray(z, np.float32)
    cb = np.ascontiguousarray(codebook, np.float32)
    B, Cc, h, w = z.shape
    K = cb.shape[0]
    idx = np.empty(B * h * w, np.int64)
    zq = np.empty_like(z)
    loss = C.c_float(0)
    rc = lib().cgic_oracle_vq(_p(z, C.c_float), C.c_long(B), C.c_int(Cc), C.c_long(h * w),
                              _p(cb, C.c_float), C.c_int(K), C.c_float(beta), C.c_int(int(legacy)),
                              _p(idx, C.c_int64), _p(zq, C.c_float), C.byref(loss),
                              _p(hist, C.c_int64))
    if rc:
        raise RuntimeError(f"cgic_oracle_vq rc={rc}")
    return zq, np.float32(loss.value), idx


def vq_distances(zvec, codebook):
    """reference distance row d[k] of one latent vector [C] against codebook [K,C] (quantize.py:73-75 rounding sequence)"""
    zv = np.ascontiguousarray(zvec, np.float32)
    cb = np.ascontiguousarray(codebook, np.float32)
    d = np.empty(cb.shape[0], np.float32)
    rc = lib().cgic_oracle_vq_distances(_p(zv, C.c_float), C.c_int(cb.shape[1]), _p(cb, C.c_float), C.c_int(cb.shape[0]), _p(d, C.c_float))
    if rc:
        raise RuntimeError(f"cgic_oracle_vq_distances rc={rc}")
    return d


def linspace_bins():
    """torch.linspace(-1, 1, 32) fp32 values (model.py:480), restated:
    step=(end-start)/(steps-1) in fp32; first half start+i*step, second half end-(steps-1-i)*step."""
    start, end, steps = np.float32(-1), np.float32(1), 32
    step = np.float32((end - start) / np.float32(steps - 1))
    out = np.empty(steps, np.float32)
    for i in range(steps):
        if i < steps // 2:
            out[i] = np.float32(start + np.float32(step * np.float32(i)))
        else:
            out[i] = np.float32(end - np.float32(step * np.float32(steps - 1 - i)))
    return out


def entropy(x, p, bins=None, sigma=0.01):
    """model.py:433-483.  x [B,3,H,W] f32 -> [B,H/p,W/p] f32"""
    x = np.ascontiguousarray(x, np.float32)
    B, ch, H, W = x.shape
    assert ch == 3
    bins = linspace_bins() if bins is None else np.ascontiguousarray(bins, np.float32)
    out = np.empty((B, H // p, W // p), np.float32)
    rc = lib().cgic_oracle_entropy(_p(x, C.c_float), C.c_long(B), C.c_long(H), C.c_long(W), C.c_int(p),
                                   _p(bins, C.c_float), C.c_int(len(bins)), C.c_float(np.float32(sigma)),
                                   _p(out, C.c_float))
    if rc:
        raise RuntimeError(f"cgic_oracle_entropy rc={rc}")
    return out


def entropy_ref(x, p, bins=None, sigma=0.01):
    """model.py:433-483 in torch's CPU operation sequence and summation order, exp / log correctly rounded
    (cgic_oracle_entropy_ref: the CPU restatement of the GPU's opt-in reference-arithmetic kernel)"""
    x = np.ascontiguousarray(x, np.float32)
    B, ch, H, W = x.shape
    assert ch == 3
    bins = linspace_bins() if bins is None else np.ascontiguousarray(bins, np.float32)
    out = np.empty((B, H // p, W // p), np.float32)
    rc = lib().cgic_oracle_entropy_ref(_p(x, C.c_float), C.c_long(B), C.c_long(H), C.c_long(W), C.c_int(p),
                                       _p(bins, C.c_float), C.c_int(len(bins)), C.c_float(np.float32(sigma)),
                                       _p(out, C.c_float))
    if rc:
        raise RuntimeError(f"cgic_oracle_entropy_ref rc={rc}")
    return out


def router_mode(c, m):
    lib().cgic_oracle_router_mode.argtypes = [C.c_double, C.c_double]
    return lib().cgic_oracle_router_mode(float(c), float(m))


def router(e16, e8, c_ratio, m_ratio, per_image=False, want_gate=True):
    """RouterTriple.py:15-95 -> (mask_c, mask_m, mask_f int32 [B,1,.,.], gate f32 [B,1,h,3w], mode)"""
    e16 = np.ascontiguousarray(e16, np.float32)
    e8 = np.ascontiguousarray(e8, np.float32)
    B, h16, w16 = e16.shape
    assert e8.shape == (B, 2 * h16, 2 * w16)
    mc = np.empty((B, 1, h16, w16), np.int32)
    mm = np.empty((B, 1, 2 * h16, 2 * w16), np.int32)
    mf = np.empty((B, 1, 4 * h16, 4 * w16), np.int32)
    gate = np.empty((B, 1, 4 * h16, 12 * w16), np.float32) if want_gate else None
    mode = C.c_int(0)
    rc = lib().cgic_oracle_router(_p(e16, C.c_float), _p(e8, C.c_float), C.c_long(B), C.c_long(h16),
                                  C.c_long(w16), C.c_double(c_ratio), C.c_double(m_ratio),
                                  C.c_long(B if per_image else 1), _p(mc, C.c_int32), _p(mm, C.c_int32),
                                  _p(mf, C.c_int32), _p(gate, C.c_float), C.byref(mode))
    if rc:
        raise RuntimeError(f"cgic_oracle_router rc={rc}")
    return mc, mm, mf, gate, mode.value


def param_dict_order(n):
    """iteration order of nn.ParameterDict({str(i): ... for i in range(n)}): keys sorted as strings"""
    return np.array(sorted(range(n), key=str), np.int32)


class HuffmanTable:
    """indices_coding.py:10-17,46-75: code lengths + MSB-first code words."""

    def __init__(self, freq, order="parameter_dict"):
        """order: the iteration order of the reference's `frequency` mapping --
        "parameter_dict" (default) = keys sorted as strings, which is what
        nn.ParameterDict({str(i): ...}) (quantize.py:28) iterates in;
        "natural" = 0,1,2,...; or an explicit permutation."""
        freq = np.ascontiguousarray(freq, np.int64)
        n = len(freq)
        self.n = n
        if isinstance(order, str):
            order = param_dict_order(n) if order == "parameter_dict" else np.arange(n)
        order = np.ascontiguousarray(order, np.int32)
        self.order = order
        self.len = np.zeros(n, np.int32)
        maxlen = lib().cgic_oracle_huffman_build(_p(freq, C.c_int64), _p(order, C.c_int32), C.c_int(n),
                                                 _p(self.len, C.c_int32), None, C.c_int(0))
        if maxlen < 0:
            raise RuntimeError(f"huffman_build rc={maxlen}")
        self.maxlen = maxlen
        self.words = max(1, (maxlen + 31) // 32)
        self.code = np.zeros((n, self.words), np.uint32)
        rc = lib().cgic_oracle_huffman_build(_p(freq, C.c_int64), _p(order, C.c_int32), C.c_int(n),
                                             _p(self.len, C.c_int32), _p(self.code, C.c_uint32),
                                             C.c_int(self.words))
        if rc < 0:
            raise RuntimeError(f"huffman_build rc={rc}")

    @classmethod
    def binary(cls):
        """mask_coding.py:11-12: codes {0:'0', 1:'1'}"""
        t = cls.__new__(cls)
        t.n, t.maxlen, t.words = 2, 1, 1
        t.len = np.array([1, 1], np.int32)
        t.code = np.array([[0], [0x80000000]], np.uint32)
        return t

    def code_str(self, s):
        return "".join("1" if (int(self.code[s, b // 32]) >> (31 - b % 32)) & 1 else "0"
                       for b in range(int(self.len[s])))


def encode(table, syms):
    """indices_coding.py:113-126 / mask_coding.py:40-55 -> bytes (b'' for empty input)"""
    syms = np.ascontiguousarray(syms, np.int64).ravel()
    cap = 2 + (int(table.maxlen) * len(syms) + 7) // 8 + 1
    out = np.zeros(max(cap, 1), np.uint8)
    n = lib().cgic_oracle_encode(_p(syms, C.c_int64), C.c_long(len(syms)), _p(table.len, C.c_int32),
                                 _p(table.code, C.c_uint32), C.c_int(table.words), C.c_int(table.n),
                                 _p(out, C.c_uint8), C.c_long(cap))
    if n < 0:
        raise RuntimeError(f"cgic_oracle_encode rc={n}")
    return out[:n].tobytes()


def decode(table, data):
    """indices_coding.py:153-168 / mask_coding.py:81-96 -> int64 array, or None for empty input"""
    buf = np.frombuffer(bytes(data), np.uint8)
    cap = max(1, len(buf) * 8)
    out = np.empty(cap, np.int64)
    n = lib().cgic_oracle_decode(_p(buf, C.c_uint8) if len(buf) else None, C.c_long(len(buf)),
                                 _p(table.len, C.c_int32), _p(table.code, C.c_uint32), C.c_int(table.words),
                                 C.c_int(table.n), _p(out, C.c_int64), C.c_long(cap))
    if n == -1:
        return None
    if n < 0:
        raise RuntimeError(f"cgic_oracle_decode rc={n}")
    return out[:n].copy()


STREAM_NAMES = ("indices_coarse", "indices_medium", "indices_fine", "mask_coarse", "mask_medium")


def mode_streams(mode):
    """model.py:225-260: tuple of 5 bools, which .bin files a mode writes"""
    m = lib().cgic_oracle_mode_streams(int(mode))
    return tuple(bool(m >> i & 1) for i in range(5))


def select(ind, mc, mm, mf):
    """model.py:217-221 for one image: ind [h,w] int64 -> three symbol lists"""
    ind = np.ascontiguousarray(ind, np.int64)
    h, w = ind.shape
    mc = np.ascontiguousarray(mc, np.int32).reshape(h // 4, w // 4)
    mm = np.ascontiguousarray(mm, np.int32).reshape(h // 2, w // 2)
    mf = np.ascontiguousarray(mf, np.int32).reshape(h, w)
    sc = np.empty(mc.size, np.int64)
    sm = np.empty(mm.size, np.int64)
    sf = np.empty(mf.size, np.int64)
    n = (C.c_long * 3)()
    lib().cgic_oracle_select(_p(ind, C.c_int64), C.c_long(h), C.c_long(w), _p(mc, C.c_int32),
                             _p(mm, C.c_int32), _p(mf, C.c_int32), _p(sc, C.c_int64), _p(sm, C.c_int64),
                             _p(sf, C.c_int64), n)
    return sc[:n[0]].copy(), sm[:n[1]].copy(), sf[:n[2]].copy()


def compress_image(ind, mc, mm, mf, mode, htab):
    """model.py:217-260 for one image -> dict name -> bytes (only the streams the mode writes)"""
    sc, sm, sf = select(ind, mc, mm, mf)
    btab = HuffmanTable.binary()
    payload = (lambda: encode(htab, sc), lambda: encode(htab, sm), lambda: encode(htab, sf),
               lambda: encode(btab, np.asarray(mc).ravel()), lambda: encode(btab, np.asarray(mm).ravel()))
    return {STREAM_NAMES[i]: payload[i]() for i, on in enumerate(mode_streams(mode)) if on}


def decompress_image(streams, mode, h, w, htab):
    """model.py:269-389 for one image -> (ind [h,w] int64, mask_c, mask_m, mask_f int32)"""
    btab = HuffmanTable.binary()
    on = mode_streams(mode)
    dec = [decode(htab, streams[STREAM_NAMES[i]]) if on[i] else None for i in range(3)]
    mcd = decode(btab, streams["mask_coarse"]) if on[3] else None
    mmd = decode(btab, streams["mask_medium"]) if on[4] else None
    mc_in = np.ascontiguousarray(mcd, np.int32) if mcd is not None else np.zeros((h // 4) * (w // 4), np.int32)
    mm_in = np.ascontiguousarray(mmd, np.int32) if mmd is not None else np.zeros((h // 2) * (w // 2), np.int32)
    if mc_in.size != (h // 4) * (w // 4) or mm_in.size != (h // 2) * (w // 2):
        raise RuntimeError("decoded mask has the wrong number of elements")
    cnt = [len(d) if d is not None else -1 for d in dec]
    arr = [np.ascontiguousarray(d, np.int64) if d is not None else np.zeros(1, np.int64) for d in dec]
    ind = np.empty((h, w), np.int64)
    mco = np.empty((h // 4, w // 4), np.int32)
    mmo = np.empty((h // 2, w // 2), np.int32)
    mfo = np.empty((h, w), np.int32)
    rc = lib().cgic_oracle_merge(C.c_int(mode), C.c_long(h), C.c_long(w), _p(mc_in, C.c_int32),
                                 _p(mm_in, C.c_int32), _p(arr[0], C.c_int64), C.c_long(cnt[0]),
                                 _p(arr[1], C.c_int64), C.c_long(cnt[1]), _p(arr[2], C.c_int64),
                                 C.c_long(cnt[2]), _p(ind, C.c_int64), _p(mco, C.c_int32),
                                 _p(mmo, C.c_int32), _p(mfo, C.c_int32))
    if rc:
        raise RuntimeError(f"cgic_oracle_merge rc={rc}")
    return ind, mco, mmo, mfo


def gather(ind, codebook):
    """model.py:391-392: ind [h,w] -> [1,C,h,w] exact codebook rows"""
    ind = np.ascontiguousarray(ind, np.int64)
    cb = np.ascontiguousarray(codebook, np.float32)
    h, w = ind.shape
    out = np.empty((1, cb.shape[1], h, w), np.float32)
    rc = lib().cgic_oracle_gather(_p(ind, C.c_int64), C.c_long(h * w), _p(cb, C.c_float), C.c_int(cb.shape[0]),
                                  C.c_int(cb.shape[1]), _p(out, C.c_float))
    if rc:
        raise RuntimeError("index out of range")
    return out
