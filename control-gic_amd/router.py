"""TripleGrainFixedEntropyRouter -- drop-in for CGIC/modules/vqvae/RouterTriple.py:7-95.

Constructed from `router_config` on every encoder forward in the reference
(CGIC/modules/vqvae/vqvae_blocks.py:354-355), so pointing
`router_config.target` at this class is the whole integration.
"""
import ctypes

import torch
from torch import nn

from . import _lib


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

    def forward(self, x_entropy_p16, x_entropy_p8, want_gate=True):
        _lib.require_device(x_entropy_p16, x_entropy_p8)
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
        with torch.cuda.device(dev):
            _lib.call("cgic_router_f32", _lib.ptr(e16), _lib.ptr(e8), B, h16, w16,
                      float(self.coarse_grain_ratio), float(self.medium_grain_ratio), int(bool(self.per_image)),
                      _lib.ptr(mc), _lib.ptr(mm), _lib.ptr(mf), _lib.ptr(gate), ctypes.byref(mode),
                      _lib.current_stream(dev))
        return [mc, mm, mf], gate, [self.coarse_grain_ratio, self.medium_grain_ratio, self.fine_grain_ratio], mode.value
