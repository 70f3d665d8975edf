"""Batched counterpart of the entropy-coder half of CGIC.compress
(CGIC/models/model.py:217-260 encode side, :269-397 decode side).

The reference handles one image per call and talks to itself through five
fixed-name .bin files.  `GrainCodec` does a whole batch in two launches per
direction and keeps the streams in one device tensor; `write_legacy()` emits
exactly the reference's files for one image, byte for byte.
"""
import os
import threading

import torch

from . import _lib
from .indices_coding import HuffmanCoding

STREAM_NAMES = _lib.STREAM_NAMES


def mode_streams(mode):
    """which of the five streams compress() writes in this mode (model.py:225-260)"""
    m = _lib.call("cgic_mode_streams", int(mode))
    return tuple(bool(m >> i & 1) for i in range(_lib.NUM_STREAMS))


_DECODE_MODES = {"auto": 0, "latency": 1, "throughput": 2}
_tls = threading.local()


def set_default_decoder(mode):
    """process-wide default of calls that name no decoder (cgic_set_decode_mode); returns the previous one"""
    prev = _lib.call("cgic_set_decode_mode", _DECODE_MODES[mode])
    return {v: k for k, v in _DECODE_MODES.items()}[prev]


class decoder_mode:
    """`with decoder_mode("throughput"):` -- which prefix decoder `GrainCodec.decompress` launches (or captures) inside the
    block when the call itself names none: "latency" = the split-stream kernels (shortest time for one batch on an idle GPU),
    "throughput" = the self-synchronising one-workgroup-per-image kernel (small footprint: for several batches in flight,
    pipeline.LaneStream), "auto" = the process default (set_default_decoder; latency unless changed).  Results are identical.
    The choice is a property of each call (the `decoder` argument of cgic_decompress_streams): this context only supplies
    the calling THREAD's default, so two threads can decode in different modes at the same time."""

    def __init__(self, mode):
        if mode not in _DECODE_MODES:
            raise ValueError(f"decoder mode {mode!r}: expected one of {sorted(_DECODE_MODES)}")
        self.mode = mode
        self.prev = None

    def __enter__(self):
        self.prev = getattr(_tls, "mode", None)
        _tls.mode = self.mode
        return self

    def __exit__(self, *exc):
        _tls.mode = self.prev
        return False


def _decoder_flag(decoder):
    mode = decoder if decoder is not None else getattr(_tls, "mode", None)
    if mode is None:
        return 0
    if mode not in _DECODE_MODES:
        raise ValueError(f"decoder mode {mode!r}: expected one of {sorted(_DECODE_MODES)}")
    return _DECODE_MODES[mode]


class CompressedBatch:
    """streams of B images on the device: data [B, 5, slot] uint8, nbytes [B, 5] int32
    (-1 = not written in this mode, 0 = empty file)"""

    def __init__(self, data, nbytes, mode, h, w):
        self.data, self.nbytes, self.mode, self.h, self.w = data, nbytes, int(mode), int(h), int(w)

    @property
    def batch(self):
        return self.data.shape[0]

    def to_host(self):
        """list (per image) of {stream name: bytes} for the streams the mode writes.  Reads the device buffers on every
        call (nothing is cached: the buffers of a captured launch are rewritten by every replay)"""
        nb = self.nbytes.cpu()
        if int(nb.min()) < -1:
            bad = int(nb.min()) + 10
            if bad == _lib.ERR_INVALID:
                raise KeyError("a symbol is not in the code table")
            raise _lib.CgicError(bad, "compress_streams failed on the device")
        top = max(int(nb.max()), 0)
        blob = self.data[:, :, :top].cpu().numpy()
        return [{STREAM_NAMES[s]: blob[b, s, :int(nb[b, s])].tobytes()
                 for s in range(_lib.NUM_STREAMS) if int(nb[b, s]) >= 0}
                for b in range(self.batch)]

    def total_bytes(self):
        """[B] int64 on the device: sum of stream sizes per image"""
        return self.nbytes.clamp(min=0).sum(dim=1, dtype=torch.int64)

    def bpp(self, num_pixels=None):
        """per-image bits per pixel, `sum(os.path.getsize) * 8 / num_pixels` (model.py:233)"""
        num_pixels = 16 * self.h * self.w if num_pixels is None else num_pixels
        return [int(t) * 8 / num_pixels for t in self.total_bytes().cpu().tolist()]

    def write_legacy(self, path, b=0, host=None):
        """write image b's streams under the reference's fixed file names; returns the paths (`host`: a to_host() result
        to reuse when several images of the batch are written)"""
        out = []
        for name, data in (self.to_host() if host is None else host)[b].items():
            p = os.path.join(path, name + ".bin")
            with open(p, "wb") as f:
                f.write(data)
            out.append(p)
        return out

    @classmethod
    def from_host(cls, images, mode, h, w, slot, device):
        """inverse of to_host(): images = list of {name: bytes}"""
        B = len(images)
        data = torch.zeros((B, _lib.NUM_STREAMS, slot), dtype=torch.uint8)
        nbytes = torch.full((B, _lib.NUM_STREAMS), -1, dtype=torch.int32)
        for b, im in enumerate(images):
            for s, name in enumerate(STREAM_NAMES):
                if name in im:
                    d = im[name]
                    if len(d) + 8 > slot:
                        raise ValueError(f"stream {name} of image {b} ({len(d)} B) does not fit slot {slot}")
                    data[b, s, :len(d)] = torch.frombuffer(bytearray(d), dtype=torch.uint8) if d else data[b, s, :0]
                    nbytes[b, s] = len(d)
        return cls(data.to(device), nbytes.to(device), mode, h, w)

    @classmethod
    def read_legacy(cls, path, mode, h, w, slot, device):
        on = mode_streams(mode)
        im = {}
        for s, name in enumerate(STREAM_NAMES):
            if on[s]:
                with open(os.path.join(path, name + ".bin"), "rb") as f:
                    im[name] = f.read()
        return cls.from_host([im], mode, h, w, slot, device)


class GrainCodec:
    """encode/decode the three index streams + two mask streams of a batch.

    frequency: the reference's `model.quantize.embedding_counter` mapping (or an existing
    HuffmanCoding); codebook: [K,4] embedding weight for the fused gather on decode."""

    def __init__(self, frequency, codebook=None):
        self.huffman = frequency if isinstance(frequency, HuffmanCoding) else HuffmanCoding(frequency)
        self.codebook = codebook

    def slot_bytes(self, h, w):
        return int(_lib.lib().cgic_compress_slot_bytes(self.huffman.table.handle, h, w))

    def compress(self, ind, masks, mode, hist=None):
        """ind [B,h,w] (or flat [B*h*w]) int64; masks = [mask_c, mask_m, mask_f] int32 -> CompressedBatch.
        hist (int64 [n_e], optional) accumulates the usage histogram of `ind` in the same launch."""
        mc, mm, mf = (m.contiguous() for m in masks)
        _lib.require_device(ind, mc, mm, mf)
        B, h, w = mf.shape[0], mf.shape[-2], mf.shape[-1]
        ind = ind.contiguous()
        if ind.numel() != B * h * w or ind.dtype != torch.int64:
            raise ValueError("ind must be int64 with B*h*w elements")
        for m in (mc, mm, mf):
            if m.dtype != torch.int32:
                raise TypeError("masks must be int32 like the router's (RouterTriple.py:92)")
        l = _lib.lib()
        dev = ind.device
        slot = self.slot_bytes(h, w)
        data = torch.empty((B, _lib.NUM_STREAMS, slot), dtype=torch.uint8, device=dev)
        nbytes = torch.empty((B, _lib.NUM_STREAMS), dtype=torch.int32, device=dev)
        wsb = l.cgic_compress_workspace_bytes(B, h, w)
        ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
        with _lib.on_device(dev):
            _lib.call("cgic_compress_streams", self.huffman.table.handle, _lib.ptr(ind), _lib.ptr(mc), _lib.ptr(mm),
                      _lib.ptr(mf), B, h, w, int(mode), _lib.ptr(data), slot, _lib.ptr(nbytes), _lib.ptr(hist),
                      _lib.ptr(ws), _lib.current_stream(dev))
        return CompressedBatch(data, nbytes, mode, h, w)

    def post_conv_table(self, post_quant_conv, bias_first=False):
        """post_quant_conv applied to the codebook rows (model.py:52,115): gathering from it == convolving the gathered latent"""
        cbk = self.codebook.detach().contiguous()
        out = torch.empty_like(cbk)
        qc, keep = _lib.conv_arg(post_quant_conv, bias_first)
        with _lib.on_device(cbk.device):
            _lib.call("cgic_conv1x1_rows_f32", _lib.ptr(cbk), cbk.shape[0], qc, _lib.ptr(out), _lib.current_stream(cbk.device))
        del keep
        return out

    def decompress(self, cb, want_masks=True, want_zq=True, post_quant_conv=None, conv_bias_first=False, decoder=None, status=None):
        """CompressedBatch -> (ind [B,h,w] int64, [mask_c, mask_m, mask_f] int32 [B,1,.,.] or None,
        z_q [B,4,h,w] fp32 or None, status [B] int32 on the device (0 = ok)).
        With post_quant_conv (a Conv2d(4, 4, 1) or (weight, bias)) the third element is the pair
        (z_q, post_quant_conv(z_q)) -- what CGIC.decode feeds the decoder (model.py:114-116) -- from the same pass.
        decoder: "latency" / "throughput" / "auto" for THIS call (None: the enclosing decoder_mode block, else the process default).
        status: where to write the [B] status words (e.g. a slice of one buffer shared by several calls) instead of a new tensor."""
        B, h, w, dev = cb.batch, cb.h, cb.w, cb.data.device
        l = _lib.lib()
        ind = torch.empty((B, h, w), dtype=torch.int64, device=dev)
        masks = None
        if want_masks:
            masks = [torch.empty((B, 1, h // 4, w // 4), dtype=torch.int32, device=dev),
                     torch.empty((B, 1, h // 2, w // 2), dtype=torch.int32, device=dev),
                     torch.empty((B, 1, h, w), dtype=torch.int32, device=dev)]
        zq = None
        cbk = None
        if want_zq:
            if self.codebook is None:
                raise ValueError("GrainCodec was built without a codebook")
            cbk = self.codebook.detach().contiguous()
            zq = torch.empty((B, cbk.shape[1], h, w), dtype=torch.float32, device=dev)
        cbk2 = zq2 = None
        if post_quant_conv is not None:
            if not want_zq:
                raise ValueError("post_quant_conv needs want_zq")
            cbk2 = self.post_conv_table(post_quant_conv, conv_bias_first)
            zq2 = torch.empty_like(zq)
        if status is None:
            status = torch.empty(B, dtype=torch.int32, device=dev)
        elif status.dtype != torch.int32 or status.numel() != B or not status.is_contiguous() or status.device != dev:
            raise ValueError("status must be a contiguous int32 tensor with one element per image on the streams' device")
        ws = torch.empty(l.cgic_decompress_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
        with _lib.on_device(dev):
            _lib.call("cgic_decompress_streams", self.huffman.table.handle, _lib.ptr(cb.data), cb.data.shape[2],
                      _lib.ptr(cb.nbytes), B, h, w, cb.mode, _lib.ptr(ind),
                      _lib.ptr(masks[0]) if masks else None, _lib.ptr(masks[1]) if masks else None,
                      _lib.ptr(masks[2]) if masks else None, _lib.ptr(cbk),
                      cbk.shape[0] if cbk is not None else 0, cbk.shape[1] if cbk is not None else 0,
                      _lib.ptr(zq), _lib.ptr(cbk2), _lib.ptr(zq2), _lib.ptr(status), _lib.ptr(ws), _decoder_flag(decoder),
                      _lib.current_stream(dev))
        return ind, masks, (zq, zq2) if post_quant_conv is not None else zq, status
