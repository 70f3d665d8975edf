"""HuffmanCoding -- drop-in for CGIC/tools/indices_coding.py:9-168.

Same constructor (`frequency`: mapping str(i) -> object with .item()), same
`.codes` / `.reverse_mapping`, same `compress(info, path) -> path` and
`decompress_string(path) -> list[int] | None`, same bytes on disk.  The table is
built by the native library (CPython-heapq tie-breaking reproduced exactly,
csrc/cgic_table.hip); encode/decode run on the GPU (csrc/cgic_coder.hip); only
the file I/O stays in Python.
"""
import ctypes

import torch

from . import _lib


class _Table:
    """owner of a native cgic_table handle"""

    def __init__(self, handle):
        self.handle = handle
        l = _lib.lib()
        self.n = l.cgic_table_num_symbols(handle)
        self.max_len = l.cgic_table_max_len(handle)
        self.words = l.cgic_table_words(handle)
        self._codes = None

    @classmethod
    def from_freq(cls, freq, order=None):
        n = len(freq)
        f = (ctypes.c_int64 * n)(*[int(v) for v in freq])
        o = (ctypes.c_int32 * n)(*[int(v) for v in order]) if order is not None else None
        h = ctypes.c_void_p()
        _lib.call("cgic_table_create", f, o, n, ctypes.byref(h))
        return cls(h)

    @classmethod
    def binary(cls):
        h = ctypes.c_void_p()
        _lib.call("cgic_table_binary", ctypes.byref(h))
        return cls(h)

    def __del__(self):
        try:
            if self.handle:
                _lib.lib().cgic_table_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def arrays(self):
        """(len [n] int32, code [n, words] uint32) host copies"""
        ln = (ctypes.c_int32 * self.n)()
        cd = (ctypes.c_uint32 * (self.n * self.words))()
        _lib.call("cgic_table_get", self.handle, ln, cd)
        return list(ln), [list(cd[i * self.words:(i + 1) * self.words]) for i in range(self.n)]

    def code_strings(self):
        if self._codes is None:
            ln, cd = self.arrays()
            self._codes = {
                s: "".join("1" if (cd[s][b // 32] >> (31 - b % 32)) & 1 else "0" for b in range(ln[s]))
                for s in range(self.n)}
        return self._codes


def _frequency_items(frequency):
    """(order, freq-by-symbol) from the reference's `frequency` mapping, honouring ITS iteration
    order (make_heap pushes nodes in that order, indices_coding.py:46-49)."""
    as_ints = getattr(frequency, "as_int_list", None)
    keys = [int(k) for k in frequency.keys()]
    n = len(keys)
    if sorted(keys) != list(range(n)):
        raise ValueError("frequency keys must be str(0)..str(n-1)")
    if as_ints is not None:
        freq = as_ints()                                   # one device->host copy
    else:
        freq = [0] * n
        for k, v in frequency.items():
            freq[int(k)] = int(v.item() if hasattr(v, "item") else v)
    return keys, freq


class _StreamCoder:
    """device-side single-stream encode/decode shared by HuffmanCoding and BinaryCoding"""

    _table = None

    def encode_to_bytes(self, info):
        _lib.require_device(info)
        if info.dim() != 1:
            info = info.reshape(-1)
        if info.dtype not in (torch.int64, torch.int32):
            info = info.to(torch.int64)
        info = info.contiguous()
        n = info.numel()
        if n == 0:
            return b""                                     # empty file (indices_coding.py:116-118)
        l = _lib.lib()
        cap = l.cgic_stream_capacity(self._table.handle, n)
        dev = info.device
        out = torch.empty(cap, dtype=torch.uint8, device=dev)
        nbytes = torch.empty(1, dtype=torch.int32, device=dev)
        wsb = l.cgic_stream_workspace_bytes(n)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        with torch.cuda.device(dev):
            _lib.call("cgic_encode_stream", self._table.handle, _lib.ptr(info), info.element_size(), n,
                      _lib.ptr(out), cap, _lib.ptr(nbytes), _lib.ptr(ws), _lib.current_stream(dev))
        nb = int(nbytes.item())
        if nb < 0:
            if nb == _lib.ERR_INVALID:
                raise KeyError("a symbol is not in the code table")   # KeyError in the reference too (:81)
            raise _lib.CgicError(nb, "encode_stream failed on the device")
        return bytes(out[:nb].cpu().numpy().tobytes())

    def decode_bytes(self, data, device=None):
        if len(data) == 0:
            return None                                    # (:158-159)
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else device
        buf = torch.zeros(len(data) + 16, dtype=torch.uint8)
        buf[:len(data)] = torch.frombuffer(bytearray(data), dtype=torch.uint8)
        buf = buf.to(device)
        cap = max(1, (len(data) - 1) * 8)
        syms = torch.empty(cap, dtype=torch.int64, device=device)
        count = torch.empty(1, dtype=torch.int64, device=device)
        with torch.cuda.device(device):
            _lib.call("cgic_decode_stream", self._table.handle, _lib.ptr(buf), len(data), _lib.ptr(syms), cap,
                      _lib.ptr(count), _lib.current_stream(device))
        c = int(count.item())
        if c == -1:
            return None
        if c < 0:
            raise _lib.CgicError(c, "decode_stream failed on the device")
        return syms[:c].cpu().tolist()

    def compress(self, info, output_path):
        data = self.encode_to_bytes(info)
        with open(output_path, "wb") as f:
            f.write(data)
        return output_path

    def decompress_string(self, path):
        with open(path, "rb") as f:
            data = f.read()
        return self.decode_bytes(data)


class HuffmanCoding(_StreamCoder):
    def __init__(self, frequency):
        order, freq = _frequency_items(frequency)
        self._table = _Table.from_freq(freq, order)

    @property
    def table(self):
        return self._table

    @property
    def codes(self):
        return self._table.code_strings()

    @property
    def reverse_mapping(self):
        return {v: k for k, v in self._table.code_strings().items()}
