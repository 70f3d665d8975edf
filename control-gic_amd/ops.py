"""`torch.ops.cgic.*` -- the hot-path kernels as PyTorch custom ops (torch.library), so that they carry a schema, shape
inference under FakeTensor / torch.compile, and (for the quantiser) an autograd formula.  The implementations are the same
ctypes calls into libcgic_hip.so the module classes use; CPU tensors raise (there is no CPU fallback).

    z_q, loss, idx = torch.ops.cgic.vq_forward(z, codebook, 0.25, True)
    e8, e16        = torch.ops.cgic.entropy_maps(x)
    mc, mm, mf     = torch.ops.cgic.router(e16, e8, 0.1, 0.8, True)
    data, nbytes   = torch.ops.cgic.compress_streams(idx, mc, mm, mf, mode, table, None)          # table = HuffmanCoding(...).table.handle.value
    ind, dc, dm, df, z_q, status = torch.ops.cgic.decompress_streams(data, nbytes, h, w, mode, table, codebook, "auto")
    h = torch.ops.cgic.grain_merge(h_c, h_m, h_f, mc, mm, mf)                                     # differentiable (vqvae_blocks.py:361-366)

A code table travels through an op as an integer: the `cgic_table*` handle of include/cgic_hip.h (ops take tensors and
scalars; the table is host-side state of the library, built once per frequency table).
"""
import ctypes
from typing import Optional, Tuple

import torch

from . import _lib
from .entropy import entropy_maps as _entropy_maps, entropy_maps_u8 as _entropy_maps_u8
from .quantize import _vq_forward, vq_backward as _vq_backward, vq_forward_route as _vq_forward_route
from .router import TripleGrainFixedEntropyRouter

_DEV = "cuda"


@torch.library.custom_op("cgic::vq_forward", mutates_args=(), device_types=_DEV)
def vq_forward(z: torch.Tensor, codebook: torch.Tensor, beta: float, legacy: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """VectorQuantize2.forward (quantize.py:69-97): (z_q [B,C,h,w], loss [], indices [B*h*w] int64)"""
    z_q, loss, idx = _vq_forward(z, codebook, beta, legacy, None)
    return z_q, loss, idx


@vq_forward.register_fake
def _(z, codebook, beta, legacy):
    B, C, h, w = z.shape
    return z.new_empty(z.shape), z.new_empty(()), z.new_empty((B * h * w,), dtype=torch.int64)


@torch.library.custom_op("cgic::vq_backward", mutates_args=(), device_types=_DEV)
def vq_backward(z: torch.Tensor, codebook: torch.Tensor, indices: torch.Tensor, g_zq: torch.Tensor, g_loss: torch.Tensor,
                beta: float, legacy: bool) -> Tuple[torch.Tensor, torch.Tensor]:
    """gradients of (z_q, loss) w.r.t. (z, codebook) -- quantize.py:85-93 under autograd; deterministic"""
    gz, gw = _vq_backward(z, codebook, indices, g_zq, g_loss, beta, legacy)
    return gz, gw


@vq_backward.register_fake
def _(z, codebook, indices, g_zq, g_loss, beta, legacy):
    return z.new_empty(z.shape), codebook.new_empty(codebook.shape)


def _vq_setup(ctx, inputs, output):
    z, codebook, beta, legacy = inputs
    ctx.save_for_backward(z, codebook, output[2])
    ctx.beta, ctx.legacy = beta, legacy


def _vq_bwd(ctx, g_zq, g_loss, _g_idx):
    z, codebook, idx = ctx.saved_tensors
    if g_zq is None:
        g_zq = torch.zeros_like(z)
    if g_loss is None:
        g_loss = torch.zeros((), dtype=torch.float32, device=z.device)
    gz, gw = torch.ops.cgic.vq_backward(z, codebook, idx, g_zq, g_loss, ctx.beta, ctx.legacy)
    return gz, gw, None, None


vq_forward.register_autograd(_vq_bwd, setup_context=_vq_setup)


@torch.library.custom_op("cgic::entropy_maps", mutates_args=(), device_types=_DEV)
def entropy_maps(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Entropy(8)(x), Entropy(16)(x) in one pass (model.py:100-101,433-483): ([B,H/8,W/8], [B,H/16,W/16])"""
    e8, e16 = _entropy_maps(x)
    return e8, e16


@entropy_maps.register_fake
def _(x):
    B, _, H, W = x.shape
    return x.new_empty((B, H // 8, W // 8)), x.new_empty((B, H // 16, W // 16))


@torch.library.custom_op("cgic::entropy_maps_reference_order", mutates_args=(), device_types=_DEV)
def entropy_maps_reference_order(x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """entropy_maps in the reference's own arithmetic (cgic_entropy_maps_ref_f32): torch's CPU summation order, exp / log
    correctly rounded -- the masks of tie-heavy content then agree with the CPU reference's"""
    e8, e16 = _entropy_maps(x, reference_order=True)
    return e8, e16


@entropy_maps_reference_order.register_fake
def _(x):
    B, _, H, W = x.shape
    return x.new_empty((B, H // 8, W // 8)), x.new_empty((B, H // 16, W // 16))


@torch.library.custom_op("cgic::entropy_maps_u8", mutates_args=(), device_types=_DEV)
def entropy_maps_u8(frames: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """T.ToTensor() and both Entropy maps of uint8 [B,H,W,3] frames in one pass (inference.py:50-59 + model.py:99-101):
    (x [B,3,H,W] fp32, e8, e16)"""
    x, e8, e16 = _entropy_maps_u8(frames)
    return x, e8, e16


@entropy_maps_u8.register_fake
def _(frames):
    B, H, W, _ = frames.shape
    f = lambda *s: frames.new_empty(s, dtype=torch.float32)
    return f(B, 3, H, W), f(B, H // 8, W // 8), f(B, H // 16, W // 16)


@torch.library.custom_op("cgic::router", mutates_args=(), device_types=_DEV)
def router(e16: torch.Tensor, e8: torch.Tensor, coarse_ratio: float, medium_ratio: float, per_image: bool,
           pixels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """TripleGrainFixedEntropyRouter.forward masks (RouterTriple.py:15-95): int32 [B,1,h16,w16], [B,1,2h16,2w16], [B,1,4h16,4w16];
    the mode is a function of the ratios alone: control_gic_amd.TripleGrainFixedEntropyRouter(c, m).mode.
    pixels: the image batch behind the maps -> threshold-band refinement (masks equal to the CPU reference's from pixels)"""
    mask, _, _, _ = TripleGrainFixedEntropyRouter(coarse_ratio, medium_ratio, per_image=per_image)(e16, e8, want_gate=False, pixels=pixels)
    return mask[0], mask[1], mask[2]


@router.register_fake
def _(e16, e8, coarse_ratio, medium_ratio, per_image, pixels=None):
    B, h16, w16 = e16.shape
    mk = lambda s: e16.new_empty((B, 1, s * h16, s * w16), dtype=torch.int32)
    return mk(1), mk(2), mk(4)


@torch.library.custom_op("cgic::vq_forward_route", mutates_args=(), device_types=_DEV)
def vq_forward_route(z: torch.Tensor, codebook: torch.Tensor, beta: float, legacy: bool, e16: torch.Tensor, e8: torch.Tensor,
                     coarse_ratio: float, medium_ratio: float, per_image: bool,
                     pixels: Optional[torch.Tensor] = None) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """vq_forward and router in ONE launch: (z_q, loss, indices, mask_c, mask_m, mask_f); pixels: see router"""
    z_q, loss, idx, mask, _, _ = _vq_forward_route(z, codebook, beta, legacy, e16, e8, coarse_ratio, medium_ratio, per_image=per_image,
                                                   pixels=pixels)
    return z_q, loss, idx, mask[0], mask[1], mask[2]


@vq_forward_route.register_fake
def _(z, codebook, beta, legacy, e16, e8, coarse_ratio, medium_ratio, per_image, pixels=None):
    B, C, h, w = z.shape
    _, h16, w16 = e16.shape
    mk = lambda s: e16.new_empty((B, 1, s * h16, s * w16), dtype=torch.int32)
    return z.new_empty(z.shape), z.new_empty(()), z.new_empty((B * h * w,), dtype=torch.int64), mk(1), mk(2), mk(4)


# ------------------------------------------------------------------------------------------------------------------
# the codec (CGIC.compress, model.py:217-260 / :269-397) and the single-stream coders (indices_coding.py, mask_coding.py)
def _table(handle: int):
    if not handle:
        raise ValueError("cgic ops: the code table handle is NULL (pass HuffmanCoding(...).table.handle.value)")
    return ctypes.c_void_p(int(handle))


def _slot_bytes(table: int, h: int, w: int) -> int:
    return int(_lib.lib().cgic_compress_slot_bytes(_table(table), int(h), int(w)))


@torch.library.custom_op("cgic::compress_streams", mutates_args=("hist",), device_types=_DEV)
def compress_streams(ind: torch.Tensor, mask_c: torch.Tensor, mask_m: torch.Tensor, mask_f: torch.Tensor, mode: int, table: int,
                     hist: Optional[torch.Tensor]) -> Tuple[torch.Tensor, torch.Tensor]:
    """masked select + Huffman + mask packing of a batch (model.py:217-260): (data uint8 [B,5,slot], nbytes int32 [B,5];
    -1 = stream not written in this mode); `hist` (int64 [n_e]) accumulates the usage histogram of `ind` in the same launch"""
    mc, mm, mf = (m.contiguous() for m in (mask_c, mask_m, mask_f))
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
    slot = _slot_bytes(table, h, w)
    data = torch.empty((B, _lib.NUM_STREAMS, slot), dtype=torch.uint8, device=dev)
    nbytes = torch.empty((B, _lib.NUM_STREAMS), dtype=torch.int32, device=dev)
    wsb = l.cgic_compress_workspace_bytes(B, h, w)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    with torch.cuda.device(dev):
        _lib.call("cgic_compress_streams", _table(table), _lib.ptr(ind), _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), B, h, w, int(mode),
                  _lib.ptr(data), slot, _lib.ptr(nbytes), _lib.ptr(hist), _lib.ptr(ws), _lib.current_stream(dev))
    return data, nbytes


@compress_streams.register_fake
def _(ind, mask_c, mask_m, mask_f, mode, table, hist):
    B, h, w = mask_f.shape[0], mask_f.shape[-2], mask_f.shape[-1]
    return (ind.new_empty((B, _lib.NUM_STREAMS, _slot_bytes(table, h, w)), dtype=torch.uint8),
            ind.new_empty((B, _lib.NUM_STREAMS), dtype=torch.int32))


_DECODERS = {"auto": 0, "latency": 1, "throughput": 2}


@torch.library.custom_op("cgic::decompress_streams", mutates_args=(), device_types=_DEV)
def decompress_streams(data: torch.Tensor, nbytes: torch.Tensor, h: int, w: int, mode: int, table: int, codebook: torch.Tensor,
                       decoder: str) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """prefix decode + mask rebuild + x2/x4 merge + embedding gather (model.py:269-397):
    (ind int64 [B,h,w], mask_c, mask_m, mask_f int32 [B,1,.,.], z_q fp32 [B,4,h,w], status int32 [B]);
    decoder: "auto" / "latency" / "throughput" -- a property of this call"""
    _lib.require_device(data, nbytes, codebook)
    if decoder not in _DECODERS:
        raise ValueError(f"decoder {decoder!r}: expected one of {sorted(_DECODERS)}")
    B, dev = data.shape[0], data.device
    data, nbytes = data.contiguous(), nbytes.contiguous()
    cbk = codebook.detach().contiguous()
    ind = torch.empty((B, h, w), dtype=torch.int64, device=dev)
    mc = torch.empty((B, 1, h // 4, w // 4), dtype=torch.int32, device=dev)
    mm = torch.empty((B, 1, h // 2, w // 2), dtype=torch.int32, device=dev)
    mf = torch.empty((B, 1, h, w), dtype=torch.int32, device=dev)
    zq = torch.empty((B, cbk.shape[1], h, w), dtype=torch.float32, device=dev)
    status = torch.empty(B, dtype=torch.int32, device=dev)
    ws = torch.empty(_lib.lib().cgic_decompress_workspace_bytes(B, h, w), dtype=torch.uint8, device=dev)
    with torch.cuda.device(dev):
        _lib.call("cgic_decompress_streams", _table(table), _lib.ptr(data), data.shape[2], _lib.ptr(nbytes), B, h, w, int(mode),
                  _lib.ptr(ind), _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), _lib.ptr(cbk), cbk.shape[0], cbk.shape[1], _lib.ptr(zq),
                  None, None, _lib.ptr(status), _lib.ptr(ws), _DECODERS[decoder], _lib.current_stream(dev))
    return ind, mc, mm, mf, zq, status


@decompress_streams.register_fake
def _(data, nbytes, h, w, mode, table, codebook, decoder):
    B = data.shape[0]
    i32 = lambda *s: data.new_empty(s, dtype=torch.int32)
    return (data.new_empty((B, h, w), dtype=torch.int64), i32(B, 1, h // 4, w // 4), i32(B, 1, h // 2, w // 2), i32(B, 1, h, w),
            codebook.new_empty((B, codebook.shape[1], h, w)), i32(B))


@torch.library.custom_op("cgic::encode_stream", mutates_args=(), device_types=_DEV)
def encode_stream(symbols: torch.Tensor, table: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """HuffmanCoding.compress / BinaryCoding.compress without the file (indices_coding.py:113-126, mask_coding.py:40-55):
    (bytes uint8 [capacity], nbytes int32 [1]); symbols: 1-D int64 / int32, at least one"""
    _lib.require_device(symbols)
    info = symbols.reshape(-1).contiguous()
    if info.dtype not in (torch.int64, torch.int32):
        raise TypeError("encode_stream: int64 / int32 symbols")
    n = info.numel()
    if n == 0:
        raise ValueError("encode_stream: an empty input is an empty FILE in the reference (indices_coding.py:116-118), not a stream")
    l, dev = _lib.lib(), info.device
    cap = l.cgic_stream_capacity(_table(table), n)
    out = torch.empty(cap, dtype=torch.uint8, device=dev)
    nbytes = torch.empty(1, dtype=torch.int32, device=dev)
    wsb = l.cgic_stream_workspace_bytes(n)
    ws = torch.empty(wsb, dtype=torch.uint8, device=dev) if wsb else None
    with torch.cuda.device(dev):
        _lib.call("cgic_encode_stream", _table(table), _lib.ptr(info), info.element_size(), n, _lib.ptr(out), cap, _lib.ptr(nbytes),
                  _lib.ptr(ws), _lib.current_stream(dev))
    return out, nbytes


@encode_stream.register_fake
def _(symbols, table):
    cap = int(_lib.lib().cgic_stream_capacity(_table(table), symbols.numel()))
    return symbols.new_empty((cap,), dtype=torch.uint8), symbols.new_empty((1,), dtype=torch.int32)


@torch.library.custom_op("cgic::decode_stream", mutates_args=(), device_types=_DEV)
def decode_stream(stream: torch.Tensor, nbytes: int, table: int) -> Tuple[torch.Tensor, torch.Tensor]:
    """HuffmanCoding.decompress_string / BinaryCoding.decompress_string without the file (indices_coding.py:153-168):
    (symbols int64 [(nbytes - 1) * 8] of which the first `count` are valid, count int64 [1]; -1 = the empty file's None).
    stream: uint8, at least nbytes + 16 readable bytes"""
    _lib.require_device(stream)
    if stream.dtype != torch.uint8 or stream.numel() < nbytes + 16:
        raise ValueError("decode_stream: uint8 buffer of at least nbytes + 16 bytes")
    dev = stream.device
    cap = max(1, (int(nbytes) - 1) * 8)
    syms = torch.empty(cap, dtype=torch.int64, device=dev)
    count = torch.empty(1, dtype=torch.int64, device=dev)
    with torch.cuda.device(dev):
        _lib.call("cgic_decode_stream", _table(table), _lib.ptr(stream.contiguous()), int(nbytes), _lib.ptr(syms), cap, _lib.ptr(count),
                  _lib.current_stream(dev))
    return syms, count


@decode_stream.register_fake
def _(stream, nbytes, table):
    return stream.new_empty((max(1, (nbytes - 1) * 8),), dtype=torch.int64), stream.new_empty((1,), dtype=torch.int64)


@torch.library.custom_op("cgic::index_histogram", mutates_args=("hist",), device_types=_DEV)
def index_histogram(indices: torch.Tensor, hist: torch.Tensor) -> None:
    """hist[indices[i]] += 1 (quantize.py:79-81), exact int64"""
    _lib.require_device(indices, hist)
    if indices.dtype != torch.int64 or hist.dtype != torch.int64:
        raise TypeError("index_histogram: int64 indices and histogram")
    idx = indices.contiguous()
    with torch.cuda.device(idx.device):
        _lib.call("cgic_index_histogram", _lib.ptr(idx), idx.numel(), hist.numel(), _lib.ptr(hist), _lib.current_stream(idx.device))


# ------------------------------------------------------------------------------------------------------------------
# the mask-merge kernels either side of the quantiser (vqvae_blocks.py:361-366, decoder.py:304-305,366-378): they sit inside the
# reference's training graph, so they carry autograd formulas.  The masks are constants of the graph (int32, from the router).
def _up(m, k):
    return m.to(torch.float32).repeat_interleave(k, dim=-2).repeat_interleave(k, dim=-1)


@torch.library.custom_op("cgic::grain_merge", mutates_args=(), device_types=_DEV)
def grain_merge(h_coarse: torch.Tensor, h_medium: torch.Tensor, h_fine: torch.Tensor, mask_c: torch.Tensor, mask_m: torch.Tensor,
                mask_f: torch.Tensor) -> torch.Tensor:
    """up4(h_coarse)*up4(mask_c) + up2(h_medium)*up2(mask_m) + h_fine*mask_f in one pass (vqvae_blocks.py:361-366), bit-identical"""
    _lib.require_device(h_coarse, h_medium, h_fine, mask_c, mask_m, mask_f)
    hc, hm, hf = (t.contiguous().float() for t in (h_coarse, h_medium, h_fine))
    mc, mm, mf = (m.contiguous() for m in (mask_c, mask_m, mask_f))
    B, C, h, w = hf.shape
    if tuple(hc.shape) != (B, C, h // 4, w // 4) or tuple(hm.shape) != (B, C, h // 2, w // 2):
        raise ValueError("h_coarse / h_medium must be the fine map's shape divided by 4 / 2")
    out = torch.empty_like(hf)
    with torch.cuda.device(hf.device):
        _lib.call("cgic_grain_merge_f32", _lib.ptr(hc), _lib.ptr(hm), _lib.ptr(hf), _lib.ptr(mc), _lib.ptr(mm),
                  _lib.ptr(mf), B, C, h, w, _lib.ptr(out), _lib.current_stream(hf.device))
    return out


@grain_merge.register_fake
def _(h_coarse, h_medium, h_fine, mask_c, mask_m, mask_f):
    return h_fine.new_empty(h_fine.shape, dtype=torch.float32)


def _grain_merge_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[3:])


def _grain_merge_bwd(ctx, g):
    mc, mm, mf = ctx.saved_tensors
    g = g.contiguous()
    B, _, h, w = g.shape
    # masks arrive as [B,1,.,.] or squeezed [B,.,.] (the forward only needs their element count): broadcast over channels
    # explicitly -- a squeezed [B,h,w] mask with B == C would otherwise line up with the CHANNEL axis and scale silently wrong
    mc = mc.reshape(B, 1, h // 4, w // 4).to(g.dtype)
    mm = mm.reshape(B, 1, h // 2, w // 2).to(g.dtype)
    mf = mf.reshape(B, 1, h, w).to(g.dtype)
    # d/dh_coarse = mask_c * (sum of g over the 4x4 cell): the window sum is the library's average pool x 16 (exact)
    g_c = torch.ops.cgic.avg_pool(g, 4) * 16.0 * mc
    g_m = torch.ops.cgic.avg_pool(g, 2) * 4.0 * mm
    return g_c, g_m, g * mf, None, None, None


grain_merge.register_autograd(_grain_merge_bwd, setup_context=_grain_merge_setup)


@torch.library.custom_op("cgic::avg_pool", mutates_args=(), device_types=_DEV)
def avg_pool(x: torch.Tensor, k: int) -> torch.Tensor:
    """torch.nn.AvgPool2d(k, k, 0) for k in (2, 4) (decoder.py:304-305,366-367): row-major window sum / k^2, bit-identical to the CPU kernel"""
    _lib.require_device(x)
    x = x.contiguous().float()
    B, C, H, W = x.shape
    out = torch.empty((B, C, H // k, W // k), dtype=torch.float32, device=x.device)
    with torch.cuda.device(x.device):
        _lib.call("cgic_avgpool_f32", _lib.ptr(x), B * C, H, W, int(k), _lib.ptr(out), _lib.current_stream(x.device))
    return out


@avg_pool.register_fake
def _(x, k):
    B, C, H, W = x.shape
    return x.new_empty((B, C, H // k, W // k), dtype=torch.float32)


def _avg_pool_setup(ctx, inputs, output):
    ctx.k = inputs[1]
    ctx.hw = tuple(inputs[0].shape[-2:])


def _avg_pool_bwd(ctx, g):
    k = ctx.k
    gx = (g * (1.0 / (k * k))).repeat_interleave(k, dim=-2).repeat_interleave(k, dim=-1)
    H, W = ctx.hw
    if gx.shape[-2] != H or gx.shape[-1] != W:                           # rows / columns the pool dropped get no gradient
        gx = torch.nn.functional.pad(gx, (0, W - gx.shape[-1], 0, H - gx.shape[-2]))
    return gx, None


avg_pool.register_autograd(_avg_pool_bwd, setup_context=_avg_pool_setup)


@torch.library.custom_op("cgic::decoder_blend_medium", mutates_args=(), device_types=_DEV)
def decoder_blend_medium(h: torch.Tensor, h_medium: torch.Tensor, mask_c: torch.Tensor, mask_m: torch.Tensor) -> torch.Tensor:
    """h * up2(mask_c) + h_medium * mask_m on the medium grid (decoder.py:372-374)"""
    _lib.require_device(h, h_medium, mask_c, mask_m)
    h, hm = h.contiguous().float(), h_medium.contiguous().float()
    mc, mm = mask_c.contiguous(), mask_m.contiguous()
    B, C, hh, ww = h.shape
    if tuple(hm.shape) != (B, C, hh, ww) or mc.numel() != B * (hh // 2) * (ww // 2) or mm.numel() != B * hh * ww:
        raise ValueError("decoder_blend_medium: h, h_medium on the medium grid; mask_c at half of it, mask_m on it")
    out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        _lib.call("cgic_decoder_blend_medium_f32", _lib.ptr(h), _lib.ptr(hm), _lib.ptr(mc), _lib.ptr(mm), B, C, hh, ww,
                  _lib.ptr(out), _lib.current_stream(h.device))
    return out


@decoder_blend_medium.register_fake
def _(h, h_medium, mask_c, mask_m):
    return h.new_empty(h.shape, dtype=torch.float32)


def _blend_m_setup(ctx, inputs, output):
    ctx.save_for_backward(inputs[2], inputs[3])


def _blend_m_bwd(ctx, g):
    mc, mm = ctx.saved_tensors
    B = g.shape[0]
    return g * _up(mc.reshape(B, 1, g.shape[-2] // 2, g.shape[-1] // 2), 2), g * mm.reshape(B, 1, g.shape[-2], g.shape[-1]).to(g.dtype), None, None


decoder_blend_medium.register_autograd(_blend_m_bwd, setup_context=_blend_m_setup)


@torch.library.custom_op("cgic::decoder_blend_fine", mutates_args=(), device_types=_DEV)
def decoder_blend_fine(h: torch.Tensor, h_fine: torch.Tensor, mask_c: torch.Tensor, mask_m: torch.Tensor, mask_f: torch.Tensor) -> torch.Tensor:
    """h * up4(mask_c) + h * up2(mask_m) + h_fine * mask_f on the fine grid (decoder.py:375-378)"""
    _lib.require_device(h, h_fine, mask_c, mask_m, mask_f)
    h, hf = h.contiguous().float(), h_fine.contiguous().float()
    mc, mm, mf = (m.contiguous() for m in (mask_c, mask_m, mask_f))
    B, C, hh, ww = h.shape
    if tuple(hf.shape) != (B, C, hh, ww) or mc.numel() != B * (hh // 4) * (ww // 4) or mm.numel() != B * (hh // 2) * (ww // 2) \
            or mf.numel() != B * hh * ww:
        raise ValueError("decoder_blend_fine: h, h_fine on the fine grid; masks at 1/4, 1/2, 1/1 of it")
    out = torch.empty_like(h)
    with torch.cuda.device(h.device):
        _lib.call("cgic_decoder_blend_fine_f32", _lib.ptr(h), _lib.ptr(hf), _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), B, C, hh, ww,
                  _lib.ptr(out), _lib.current_stream(h.device))
    return out


@decoder_blend_fine.register_fake
def _(h, h_fine, mask_c, mask_m, mask_f):
    return h.new_empty(h.shape, dtype=torch.float32)


def _blend_f_setup(ctx, inputs, output):
    ctx.save_for_backward(*inputs[2:])


def _blend_f_bwd(ctx, g):
    mc, mm, mf = ctx.saved_tensors
    B, _, hh, ww = g.shape
    # (the reference's expression adds h twice where both coarser masks are set; they never overlap in a router's output)
    wh = _up(mc.reshape(B, 1, hh // 4, ww // 4), 4) + _up(mm.reshape(B, 1, hh // 2, ww // 2), 2)
    return g * wh, g * mf.reshape(B, 1, hh, ww).to(g.dtype), None, None, None


decoder_blend_fine.register_autograd(_blend_f_bwd, setup_context=_blend_f_setup)
