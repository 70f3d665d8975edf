"""Entropy -- drop-in for CGIC/models/model.py:433-483 (the router's input maps).

`entropy_maps(x)` produces the patch-8 and patch-16 maps in ONE pass over the
image; `Entropy(p)` keeps the reference's per-patch-size module API on top of it.
"""
import ctypes

import torch
from torch import nn

from . import _lib

_bins = _lib.linspace_bins         # torch.linspace(-1, 1, 32) evaluated on the CPU, like the CPU reference path (model.py:480)


class _MadeMaps:
    """what entropy_maps_tiles leaves on the tile batch it wrote: the maps made in the same pass -- by WEAK reference (the maps point
    back at the batch for the router's refinement: strong references both ways would be a cycle of ~40 MB of device memory per tiled
    image that only the cyclic collector frees) and with the batch's version counter (a batch modified in place has no maps)"""

    def __init__(self, batch, e8, e16, flat8):
        import weakref
        self.refs = tuple(weakref.ref(t) for t in (e8, e16, flat8))
        self.version = batch._version

    def get(self, batch):
        maps = tuple(r() for r in self.refs)
        if batch._version != self.version or any(m is None for m in maps):
            return None
        return maps


def _tag(maps, pixels, flat8):
    """the maps carry what the router's refinement wants (plain attributes: gone after any arithmetic on a map)"""
    for e in maps:
        if e is not None:
            e._cgic_pixels, e._cgic_flat8 = pixels, flat8


def entropy_maps(x, want8=True, want16=True, sigma=0.01, reference_order=False, want_flat=True):
    """x [B,3,H,W] fp32 on the device, H and W multiples of 16 -> (e8 [B,H/8,W/8], e16 [B,H/16,W/16]).
    The maps come back tagged with the pixels they were made from and (want_flat) the by-product map of constant 8x8 patches:
    control_gic_amd's router finds both there (TripleGrainFixedEntropyRouter.forward) for its threshold-band refinement.
    reference_order=True: the reference's own fp32 operation sequence and summation order with correctly rounded exp / log
    (cgic_entropy_maps_ref_f32): bit-identical to the CPU reference in ~99 % of the values and no flipped mask element on
    tie-heavy content, at ~8x the instructions -- for when the masks must agree with the CPU reference's to the bit."""
    _lib.require_device(x)
    if x.dim() != 4 or x.shape[1] != 3:
        raise ValueError(f"expected [B,3,H,W], got {tuple(x.shape)}")
    made = getattr(x, "_cgic_maps", None)
    made = made.get(x) if made is not None else None
    if made is not None and not reference_order and float(sigma) == 0.01:
        # a tile batch that entropy_maps_tiles wrote (and nobody has touched since): its maps were made in the same pass
        return (made[0] if want8 else None), (made[1] if want16 else None)
    x = x.contiguous().float()
    B, _, H, W = x.shape
    e8 = torch.empty((B, H // 8, W // 8), dtype=torch.float32, device=x.device) if want8 else None
    e16 = torch.empty((B, H // 16, W // 16), dtype=torch.float32, device=x.device) if want16 else None
    with _lib.on_device(x.device):
        if reference_order:
            _lib.call("cgic_entropy_maps_ref_f32", _lib.ptr(x), B, H, W, _bins(), 32, float(sigma), _lib.ptr(e8), _lib.ptr(e16),
                      _lib.current_stream(x.device))
        else:
            flat8 = torch.empty((B, H // 8, W // 8), dtype=torch.float32, device=x.device) if want_flat else None
            _lib.call("cgic_entropy_maps_f32", _lib.ptr(x), B, H, W, _bins(), 32, float(sigma), _lib.ptr(e8), _lib.ptr(e16), _lib.ptr(flat8),
                      _lib.current_stream(x.device))
            _tag((e8, e16), x, flat8)
    return e8, e16


def entropy_maps_tiles(src, origins, th, tw, sigma=0.01):
    """pad + crop of the tiling driver (inference_high_resolution.py:145-173, :236-244) and both entropy maps in ONE pass
    (cgic_entropy_maps_tiles).  src: fp32 [N,3,H,W] unpadded images (or uint8 frames [N,H,W,3]); origins: [(y0, x0)] of the T tiles of ONE
    shape th x tw in unpadded coordinates -> (tiles [N*T,3,th,tw] fp32 image-major, e8, e16).  The tile batch comes back tagged with its
    maps (weakly: keep e8 / e16 alive for as long as they are wanted): entropy_maps(tiles) returns them without another pass, and the
    router finds pixels + flat8 on the maps as usual.  The tag is void once the batch has been modified in place."""
    import ctypes
    _lib.require_device(src)
    u8 = src.dtype == torch.uint8
    if src.dim() != 4 or (src.shape[3] if u8 else src.shape[1]) != 3 or (not u8 and src.dtype != torch.float32):
        raise ValueError(f"expected fp32 [N,3,H,W] or uint8 [N,H,W,3], got {src.dtype} {tuple(src.shape)}")
    src = src.contiguous()
    N = src.shape[0]
    H, W = (src.shape[1], src.shape[2]) if u8 else (src.shape[2], src.shape[3])
    T = len(origins)
    dev = src.device
    tiles = torch.empty((N * T, 3, th, tw), dtype=torch.float32, device=dev)
    e8 = torch.empty((N * T, th // 8, tw // 8), dtype=torch.float32, device=dev)
    e16 = torch.empty((N * T, th // 16, tw // 16), dtype=torch.float32, device=dev)
    flat8 = torch.empty((N * T, th // 8, tw // 8), dtype=torch.float32, device=dev)
    org = (ctypes.c_int * (2 * T))(*[int(v) for yx in origins for v in yx])
    with _lib.on_device(dev):
        _lib.call("cgic_entropy_maps_tiles", _lib.ptr(src), int(u8), N, H, W, T, org, th, tw, _bins(), 32, float(sigma), _lib.ptr(tiles),
                  _lib.ptr(e8), _lib.ptr(e16), _lib.ptr(flat8), _lib.current_stream(dev))
    _tag((e8, e16), tiles, flat8)
    tiles._cgic_maps = _MadeMaps(tiles, e8, e16, flat8)
    return tiles, e8, e16


def entropy_maps_u8(frames, want_x=True, want8=True, want16=True, sigma=0.01, want_flat=True):
    """frames [B,H,W,3] uint8 on the device (PIL / decoder layout), H and W multiples of 16 -> (x, e8, e16): ToTensor and both Entropy
    maps in ONE pass (inference.py:50-59 + model.py:99-101).  x [B,3,H,W] fp32 = frames / 255 exactly as T.ToTensor() rounds it
    (None with want_x=False); the maps are bit-identical to entropy_maps(x)."""
    _lib.require_device(frames)
    if frames.dim() != 4 or frames.shape[3] != 3 or frames.dtype != torch.uint8:
        raise ValueError(f"expected uint8 [B,H,W,3], got {frames.dtype} {tuple(frames.shape)}")
    frames = frames.contiguous()
    B, H, W, _ = frames.shape
    dev = frames.device
    x = torch.empty((B, 3, H, W), dtype=torch.float32, device=dev) if want_x else None
    e8 = torch.empty((B, H // 8, W // 8), dtype=torch.float32, device=dev) if want8 else None
    e16 = torch.empty((B, H // 16, W // 16), dtype=torch.float32, device=dev) if want16 else None
    flat8 = torch.empty((B, H // 8, W // 8), dtype=torch.float32, device=dev) if want_flat else None
    with _lib.on_device(dev):
        _lib.call("cgic_entropy_maps_u8", _lib.ptr(frames), B, H, W, _bins(), 32, float(sigma), _lib.ptr(x), _lib.ptr(e8), _lib.ptr(e16),
                  _lib.ptr(flat8), _lib.current_stream(dev))
    _tag((e8, e16), frames, flat8)
    return x, e8, e16


class Entropy(nn.Module):
    def __init__(self, patch_size, reference_order=False):
        super().__init__()
        self.reference_order = bool(reference_order)          # see entropy_maps
        if patch_size not in (8, 16):
            raise NotImplementedError(
                f"Entropy(patch_size={patch_size}): Control-GIC uses (8, 16) (config_inference.yaml:11-13); "
                "other sizes are not built")
        self.psize = patch_size

    def forward(self, inputs):
        e8, e16 = entropy_maps(inputs, want8=self.psize == 8, want16=self.psize == 16, reference_order=self.reference_order)
        # The router is reached through the reference's own Encoder.forward(x, x_entropy_p16, x_entropy_p8)
        # (vqvae_blocks.py:303,355), which hands it the maps and nothing else: entropy_maps tags the map with the pixels it was
        # made from, and control_gic_amd's router uses them for its threshold-band refinement (masks equal to the CPU
        # reference's from PIXELS).  A plain attribute: it does not survive arithmetic on the map, and then the maps decide.
        return e8 if self.psize == 8 else e16
