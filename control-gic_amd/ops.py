"""`torch.ops.cgic.*` -- the hot-path kernels as PyTorch custom ops (torch.library), so that they carry a schema, shape
inference under FakeTensor / torch.compile, and (for the quantiser) an autograd formula.  The implementations are the same
ctypes calls into libcgic_hip.so the module classes use; CPU tensors raise (there is no CPU fallback).

    z_q, loss, idx = torch.ops.cgic.vq_forward(z, codebook, 0.25, True)
    e8, e16        = torch.ops.cgic.entropy_maps(x)
    mc, mm, mf     = torch.ops.cgic.router(e16, e8, 0.1, 0.8, True)
"""
from typing import Tuple

import torch

from . import _lib
from .entropy import entropy_maps as _entropy_maps
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


@torch.library.custom_op("cgic::router", mutates_args=(), device_types=_DEV)
def router(e16: torch.Tensor, e8: torch.Tensor, coarse_ratio: float, medium_ratio: float, per_image: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """TripleGrainFixedEntropyRouter.forward masks (RouterTriple.py:15-95): int32 [B,1,h16,w16], [B,1,2h16,2w16], [B,1,4h16,4w16];
    the mode is a function of the ratios alone: control_gic_amd.TripleGrainFixedEntropyRouter(c, m).mode"""
    mask, _, _, _ = TripleGrainFixedEntropyRouter(coarse_ratio, medium_ratio, per_image=per_image)(e16, e8, want_gate=False)
    return mask[0], mask[1], mask[2]


@router.register_fake
def _(e16, e8, coarse_ratio, medium_ratio, per_image):
    B, h16, w16 = e16.shape
    mk = lambda s: e16.new_empty((B, 1, s * h16, s * w16), dtype=torch.int32)
    return mk(1), mk(2), mk(4)


@torch.library.custom_op("cgic::vq_forward_route", mutates_args=(), device_types=_DEV)
def vq_forward_route(z: torch.Tensor, codebook: torch.Tensor, beta: float, legacy: bool, e16: torch.Tensor, e8: torch.Tensor,
                     coarse_ratio: float, medium_ratio: float, per_image: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
    """vq_forward and router in ONE launch: (z_q, loss, indices, mask_c, mask_m, mask_f)"""
    z_q, loss, idx, mask, _, _ = _vq_forward_route(z, codebook, beta, legacy, e16, e8, coarse_ratio, medium_ratio, per_image=per_image)
    return z_q, loss, idx, mask[0], mask[1], mask[2]


@vq_forward_route.register_fake
def _(z, codebook, beta, legacy, e16, e8, coarse_ratio, medium_ratio, per_image):
    B, C, h, w = z.shape
    _, h16, w16 = e16.shape
    mk = lambda s: e16.new_empty((B, 1, s * h16, s * w16), dtype=torch.int32)
    return z.new_empty(z.shape), z.new_empty(()), z.new_empty((B * h * w,), dtype=torch.int64), mk(1), mk(2), mk(4)
