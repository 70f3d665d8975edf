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
from .quantize import VectorQuantize2

ROUTER_TARGET = "control_gic_amd.router.TripleGrainFixedEntropyRouter"


def grain_merge(h_coarse, h_medium, h_fine, mask):
    """h = up4(h_coarse)*up4(mask[0]) + up2(h_medium)*up2(mask[1]) + h_fine*mask[2]
    (vqvae_blocks.py:361-366) in one pass; mask = the router's three int32 tensors"""
    _lib.require_device(h_coarse, h_medium, h_fine, *mask)
    hc, hm, hf = (t.contiguous().float() for t in (h_coarse, h_medium, h_fine))
    mc, mm, mf = (m.contiguous() for m in mask)
    B, C, h, w = hf.shape
    if tuple(hc.shape) != (B, C, h // 4, w // 4) or tuple(hm.shape) != (B, C, h // 2, w // 2):
        raise ValueError("h_coarse / h_medium must be the fine map's shape divided by 4 / 2")
    out = torch.empty_like(hf)
    with torch.cuda.device(hf.device):
        _lib.call("cgic_grain_merge_f32", _lib.ptr(hc), _lib.ptr(hm), _lib.ptr(hf), _lib.ptr(mc), _lib.ptr(mm),
                  _lib.ptr(mf), B, C, h, w, _lib.ptr(out), _lib.current_stream(hf.device))
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
    Every image is routed on its own thresholds (install() configures the router that way)."""
    assert len(input.shape) == 4                                         # model.py:207
    codec = _codec_for(model, h_indices)
    quant, diff, grain_indices, grain_mask, ind, _, mode = model.encode(input)
    comp = codec.compress(ind, grain_mask, mode)
    bpp = comp.bpp(input.shape[2] * input.shape[3])                      # model.py:223,233
    dec = None
    if decode:
        ind_d, mask_d, quant_d, status = codec.decompress(comp)
        if int(status.abs().max()) != 0:
            raise RuntimeError("decoded symbol count does not match its mask")   # shape mismatch in the reference
        dec = model.decode(quant_d, mask_d)                              # model.py:399
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


def install(model, per_image=True):
    """swap VectorQuantize2 / Entropy / router target / compress of a reference CGIC instance in place"""
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
    model.compress = types.MethodType(compress, model)
    model.compress_batch = types.MethodType(compress_batch, model)
    model._cgic_codec = None
    return model
