"""Drop-in for the hot-path parts of the CGIC LightningModule (CGIC/models/model.py).

`install(model)` swaps the hot-path submodules of an existing reference `CGIC` instance for the MI355X
ones in place (same state_dict keys, so the published checkpoint keeps loading) and rebinds
`model.compress`.  `compress` keeps the reference's signature and return value; `compress_batch` is the
batched form the reference does not have (its `compress` raises IndexError for B > 1, model.py:219).
The conv encoder / decoder stay whatever the model already has (stock PyTorch; out of scope here).
"""
import types

import torch

from . import _lib
from .codec import GrainCodec
from .entropy import Entropy
from .indices_coding import HuffmanCoding
from .quantize import FusedQuantConv, VectorQuantize2

ROUTER_TARGET = "control_gic_amd.router.TripleGrainFixedEntropyRouter"


def grain_merge(h_coarse, h_medium, h_fine, mask):
    """h = up4(h_coarse)*up4(mask[0]) + up2(h_medium)*up2(mask[1]) + h_fine*mask[2]
    (vqvae_blocks.py:361-366) in one pass; mask = the router's three int32 tensors.  Differentiable w.r.t. the three
    latents (torch.ops.cgic.grain_merge: the op sits inside the reference's training graph)."""
    return torch.ops.cgic.grain_merge(h_coarse, h_medium, h_fine, mask[0], mask[1], mask[2])


def avg_pool(x, k):
    """torch.nn.AvgPool2d(k, k, 0) for k in (2, 4) -- decoder.py:304-305,366-367; bit-identical to the CPU kernel
    (row-major running sum of the window, divided by k*k); differentiable (torch.ops.cgic.avg_pool)"""
    return torch.ops.cgic.avg_pool(x, int(k))


def decoder_blend_medium(h, h_medium, mask, out=None):
    """h * up2(mask[0]) + h_medium * mask[1] on the medium grid (decoder.py:372-374).  Differentiable
    (torch.ops.cgic.decoder_blend_medium); with `out` (which may be `h`: in place) the raw kernel call, no autograd."""
    if out is None:
        return torch.ops.cgic.decoder_blend_medium(h, h_medium, mask[0], mask[1])
    _lib.require_device(h, h_medium, mask[0], mask[1])
    h, hm = h.contiguous().float(), h_medium.contiguous().float()
    mc, mm = mask[0].contiguous(), mask[1].contiguous()
    B, C, hh, ww = h.shape
    if tuple(hm.shape) != (B, C, hh, ww) or mc.numel() != B * (hh // 2) * (ww // 2) or mm.numel() != B * hh * ww:
        raise ValueError("decoder_blend_medium: h, h_medium on the medium grid; mask[0] at half of it, mask[1] on it")
    with torch.cuda.device(h.device):
        _lib.call("cgic_decoder_blend_medium_f32", _lib.ptr(h), _lib.ptr(hm), _lib.ptr(mc), _lib.ptr(mm), B, C, hh, ww,
                  _lib.ptr(out), _lib.current_stream(h.device))
    return out


def decoder_blend_fine(h, h_fine, mask, out=None):
    """h * up4(mask[0]) + h * up2(mask[1]) + h_fine * mask[2] on the fine grid (decoder.py:375-378).  Differentiable
    (torch.ops.cgic.decoder_blend_fine); with `out` the raw kernel call (in place if `out is h`), no autograd."""
    if out is None:
        return torch.ops.cgic.decoder_blend_fine(h, h_fine, mask[0], mask[1], mask[2])
    _lib.require_device(h, h_fine, *mask)
    h, hf = h.contiguous().float(), h_fine.contiguous().float()
    mc, mm, mf = (m.contiguous() for m in mask)
    B, C, hh, ww = h.shape
    if tuple(hf.shape) != (B, C, hh, ww) or mc.numel() != B * (hh // 4) * (ww // 4) or mm.numel() != B * (hh // 2) * (ww // 2) \
            or mf.numel() != B * hh * ww:
        raise ValueError("decoder_blend_fine: h, h_fine on the fine grid; masks at 1/4, 1/2, 1/1 of it")
    with torch.cuda.device(h.device):
        _lib.call("cgic_decoder_blend_fine_f32", _lib.ptr(h), _lib.ptr(hf), _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), B, C, hh, ww,
                  _lib.ptr(out), _lib.current_stream(h.device))
    return out


def _codec_for(model, h_indices=None):
    c = getattr(model, "_cgic_codec", None)
    if c is None or c.codebook is not model.quantize.embedding.weight:
        huff = h_indices if isinstance(h_indices, HuffmanCoding) else HuffmanCoding(model.quantize.embedding_counter)
        c = GrainCodec(huff, model.quantize.embedding.weight)
        model._cgic_codec = c
    return c


def compress_batch(model, input, h_indices=None, decode=True):
    """batched CGIC.compress (model.py:206-401): -> (dec [B,3,H,W] or None, bpp list[B], CompressedBatch).
    Every image is routed on its own thresholds (what B independent B=1 calls of the reference give)."""
    assert len(input.shape) == 4                                         # model.py:207
    codec = _codec_for(model, h_indices)
    # "B independent B=1 calls": every image routed on its own thresholds, whatever the router's batch semantics
    # for encode()/forward() are (the reference flattens the batch, RouterTriple.py:21-31)
    rc = getattr(model.encoder, "router_config", None)
    params = rc.get("params") if isinstance(rc, dict) or hasattr(rc, "get") else None
    saved = params.get("per_image", None) if params is not None else None
    if params is not None:
        params["per_image"] = True
    try:
        quant, diff, grain_indices, grain_mask, ind, _, mode = model.encode(input)
    finally:
        if params is not None:
            if saved is None:
                params.pop("per_image", None)
            else:
                params["per_image"] = saved
    comp = codec.compress(ind, grain_mask, mode)
    bpp = comp.bpp(input.shape[2] * input.shape[3])                      # model.py:223,233
    dec = None
    if decode:
        pqc = getattr(model, "post_quant_conv", None)
        fuse = (getattr(model, "_cgic_fuse_post_quant_conv", False) and isinstance(pqc, torch.nn.Conv2d)
                and tuple(pqc.weight.shape) == (4, 4, 1, 1) and hasattr(model, "decoder"))
        ind_d, mask_d, quant_d, status = codec.decompress(comp, post_quant_conv=pqc if fuse else None)
        if int(status.abs().max()) != 0:
            raise RuntimeError("decoded symbol count does not match its mask")   # shape mismatch in the reference
        if fuse:
            quant, quant2 = quant_d                                      # post_quant_conv came out of the gather
            dec = model.decoder(quant2, quant, mask_d)                   # model.py:115-116
        else:
            dec = model.decode(quant_d, mask_d)                          # model.py:399
    return dec, bpp, comp


def compress(self, input, path, h_indices=None, h_mask=None, save_img=False):
    """CGIC.compress with the reference's signature and return value (dec, bpp, partition_map); also
    leaves the reference's five .bin files for the image in `path` (model.py:226-249).  B must be 1 like
    the reference; use compress_batch for more."""
    if save_img:
        raise NotImplementedError("partition-map drawing (CGIC/modules/draw.py) is outside the hot path")
    if input.shape[0] != 1:
        raise IndexError("compress() takes one image like the reference (model.py:219); use compress_batch")
    dec, bpp, comp = compress_batch(self, input, h_indices)
    comp.write_legacy(path, 0)
    return dec, bpp[0], None


def install(model, per_image=False, fuse_convs=True):
    """swap VectorQuantize2 / Entropy / router target / compress of a reference CGIC instance in place.
    per_image=False keeps the reference's routing for encode() / forward() / training (thresholds over the flattened
    batch, RouterTriple.py:21-31); compress_batch / compress / the tiling driver always route per image.
    fuse_convs: move quant_conv into the VQ kernel and post_quant_conv into the decode-side gather (under no_grad).  The
    hand-off is explicit: under no_grad `model.quant_conv(h)` returns a quantize.PendingQuantConv (its input, tagged) that
    `model.quantize` convolves inside its kernel; used anywhere else it behaves as the convolved latent, and a latent that
    reaches `model.quantize` as an ordinary tensor is quantised as it is."""
    old = model.quantize
    dev = old.embedding.weight.device
    q = VectorQuantize2(old.n_e, old.e_dim, beta=old.beta, legacy=getattr(old, "legacy", True))
    q.load_state_dict(old.state_dict(), strict=False)
    q.to(dev).train(old.training)
    model.quantize = q
    for name, p in (("entropy_calculation_p8", 8), ("entropy_calculation_p16", 16)):
        if hasattr(model, name):
            setattr(model, name, Entropy(p))
    rc = getattr(model.encoder, "router_config", None)
    if rc is not None:
        rc["target"] = ROUTER_TARGET
        rc["params"]["per_image"] = bool(per_image)
    # the two 1x1 convolutions either side of the quantiser (model.py:51-52): quant_conv runs inside the VQ kernel,
    # post_quant_conv becomes a second gather table of the decode-side merge kernel (inference only; autograd sees Conv2d)
    if fuse_convs and isinstance(getattr(model, "quant_conv", None), torch.nn.Conv2d) \
            and tuple(model.quant_conv.weight.shape) == (4, 4, 1, 1) and q.n_e % 64 == 0 and q.n_e <= 1024:
        model.quant_conv = FusedQuantConv.adopt(model.quant_conv)          # hands its input to the quantiser as a PendingQuantConv
    model._cgic_fuse_post_quant_conv = bool(fuse_convs)
    model.compress = types.MethodType(compress, model)
    model.compress_batch = types.MethodType(compress_batch, model)
    model._cgic_codec = None
    return model
