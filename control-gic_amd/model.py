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
    """the model's GrainCodec, rebuilt when what it was built from changed: the codebook tensor, the usage counters (the
    Huffman table IS the counters: a train() step between two compress() calls changes it -- round 3 kept a stale table), or the
    coder object the caller hands in.  `h_indices` (model.py:206): a control_gic_amd HuffmanCoding is used as it is; a FOREIGN
    coder (the reference's own class, whose codes the GPU coder cannot take over) must describe the same code as the counters
    do, otherwise its files would not be what this compress() writes: that raises instead of being ignored."""
    q = model.quantize
    own = h_indices if isinstance(h_indices, HuffmanCoding) else None
    counter = q.usage_counter if hasattr(q, "usage_counter") else None
    c = getattr(model, "_cgic_codec", None)
    key = getattr(model, "_cgic_codec_key", None)
    fresh = (c is not None and key is not None and c.codebook is q.embedding.weight and key[0] is own
             and (counter is None or (key[1].device == counter.device and torch.equal(key[1], counter))))
    if not fresh:
        huff = own if own is not None else HuffmanCoding(q.embedding_counter)
        c = GrainCodec(huff, q.embedding.weight)
        model._cgic_codec = c
        model._cgic_codec_key = (own, None if counter is None else counter.detach().clone())
    if h_indices is not None and own is None:
        theirs = getattr(h_indices, "codes", None)
        if not isinstance(theirs, dict) or {int(k): v for k, v in theirs.items()} != c.huffman.codes:
            raise ValueError("compress(h_indices=...): the coder handed in does not carry the code table of this model's "
                             "embedding_counter (or is not a Huffman coder with .codes); build it from "
                             "model.quantize.embedding_counter like inference.py:150, or pass a control_gic_amd.HuffmanCoding")
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


class AvgPool(torch.nn.Module):
    """torch.nn.AvgPool2d(k, k, 0) on the library's kernel (decoder.py:304-305): bit-identical to the CPU kernel, differentiable"""

    def __init__(self, k):
        super().__init__()
        self.k = int(k)

    def forward(self, x):
        if not x.is_cuda or x.dtype != torch.float32 or x.dim() != 4 or x.shape[2] % self.k or x.shape[3] % self.k:
            return torch.nn.functional.avg_pool2d(x, self.k, self.k, 0)
        return torch.ops.cgic.avg_pool(x, self.k)


def install(model, per_image=False, fuse_convs=True, patch_pools=True):
    """swap VectorQuantize2 / Entropy / router target / compress of a reference CGIC instance in place.
    patch_pools: the decoder's two average pools in front of its masked blends (decoder.avgpool_layer1 / _layer2,
    decoder.py:304-305,366-367) become control_gic_amd.model.AvgPool -- they are modules, so they can be swapped.  The three
    masked-blend EXPRESSIONS (vqvae_blocks.py:364-366, decoder.py:372-378) are inline arithmetic of the reference's forward
    methods: using grain_merge / decoder_blend_medium / decoder_blend_fine there takes the two source edits INTEGRATION.md shows.
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
    dec = getattr(model, "decoder", None)
    if patch_pools and dec is not None:
        for name, k in (("avgpool_layer1", 4), ("avgpool_layer2", 2)):
            m = getattr(dec, name, None)
            if isinstance(m, torch.nn.AvgPool2d) and m.kernel_size in (k, (k, k)) and m.stride in (k, (k, k)) and m.padding in (0, (0, 0)):
                setattr(dec, name, AvgPool(k))
    model.compress = types.MethodType(compress, model)
    model.compress_batch = types.MethodType(compress_batch, model)
    model._cgic_codec = None
    model._cgic_codec_key = None
    return model
