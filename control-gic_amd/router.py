"""TripleGrainFixedEntropyRouter -- drop-in for CGIC/modules/vqvae/RouterTriple.py:7-95.

Constructed from `router_config` on every encoder forward in the reference
(CGIC/modules/vqvae/vqvae_blocks.py:354-355), so pointing
`router_config.target` at this class is the whole integration.
"""
import ctypes

import torch
from torch import nn

from . import _lib


def _flat_of(pixels, *maps):
    """the constant-patch map entropy_maps left on one of `maps`, if it was made from these very pixels"""
    for e in maps:
        if getattr(e, "_cgic_pixels", None) is pixels and getattr(e, "_cgic_flat8", None) is not None:
            return e._cgic_flat8
    return None


class TripleGrainFixedEntropyRouter(nn.Module):
    #: False = thresholds over the flattened batch (the reference's behaviour);
    #: True = one threshold set per image (what B independent B=1 calls give).
    per_image = False

    def __init__(self, coarse_grain_ratio, medium_grain_ratio, per_image=None):
        super().__init__()
        self.coarse_grain_ratio = coarse_grain_ratio
        self.medium_grain_ratio = medium_grain_ratio
        self.fine_grain_ratio = 1 - coarse_grain_ratio - medium_grain_ratio     # float64, RouterTriple.py:13
        if per_image is not None:
            self.per_image = per_image

    @property
    def mode(self):
        return _lib.lib().cgic_router_mode(float(self.coarse_grain_ratio), float(self.medium_grain_ratio))

    #: True (default): when the pixels behind the maps are known -- `pixels=`, or maps that come from control_gic_amd.Entropy
    #: and carry them -- patches whose entropy lies within the entropy kernel's error of a threshold are re-evaluated in the
    #: reference's own arithmetic inside the router (cgic_router_f32's `refine`): masks equal to the CPU reference's from pixels
    refine = True

    def forward(self, x_entropy_p16, x_entropy_p8, want_gate=True, pixels=None, flat8=None):
        _lib.require_device(x_entropy_p16, x_entropy_p8)
        explicit = pixels is not None        # asked for by the caller: a segment that cannot be refined raises (from the maps' tags: warns)
        if pixels is None and self.refine:
            p16, p8 = getattr(x_entropy_p16, "_cgic_pixels", None), getattr(x_entropy_p8, "_cgic_pixels", None)
            pixels = p16 if (p16 is not None and p16 is p8) else None        # both maps from the same image batch
        if flat8 is None and pixels is not None:
            flat8 = _flat_of(pixels, x_entropy_p8, x_entropy_p16)
        e16 = x_entropy_p16.contiguous().float()
        e8 = x_entropy_p8.contiguous().float()
        B, h16, w16 = e16.shape
        if tuple(e8.shape) != (B, 2 * h16, 2 * w16):
            raise ValueError(f"x_entropy_p8 {tuple(e8.shape)} must be [B, 2*h16, 2*w16] of {tuple(e16.shape)}")
        dev = e16.device
        mc = torch.empty((B, 1, h16, w16), dtype=torch.int32, device=dev)
        mm = torch.empty((B, 1, 2 * h16, 2 * w16), dtype=torch.int32, device=dev)
        mf = torch.empty((B, 1, 4 * h16, 4 * w16), dtype=torch.int32, device=dev)
        gate = torch.empty((B, 1, 4 * h16, 12 * w16), dtype=torch.float32, device=dev) if want_gate else None
        mode = ctypes.c_int(0)
        px, keep = _lib.pixels_arg(pixels if self.refine else None, B, h16, w16, self.per_image, flat8=flat8, queues=True, explicit=explicit)
        with torch.cuda.device(dev):
            _lib.call("cgic_router_f32", _lib.ptr(e16), _lib.ptr(e8), B, h16, w16,
                      float(self.coarse_grain_ratio), float(self.medium_grain_ratio), int(bool(self.per_image)),
                      _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), _lib.ptr(gate), ctypes.byref(mode), px,
                      _lib.current_stream(dev))
        del keep
        return [mc, mm, mf], gate, [self.coarse_grain_ratio, self.medium_grain_ratio, self.fine_grain_ratio], mode.value
