"""Tiling driver for high-resolution images -- the hot-path side of inference_high_resolution.py.

The reference pads the image to a multiple of 16 (centred zero pad, :145-173,:227-228), cuts it into a
non-overlapping 768-px grid with ragged last row/column (:112-125) and calls `model.compress` once per tile
at B=1 (:246), overwriting the same five .bin files every time; bpp is accumulated as
sum(bpp_tile * tile_w * tile_h) / (W * H) of the UNPADDED image (:250,:256).

Here tiles of equal shape are batched through the kernels together (router thresholds stay per tile =
per image of the batch), the streams stay on the device, and the accounting is the reference's.
"""
import math

import torch

from . import _lib
from .codec import GrainCodec

TILE = 768


def compute_padding(in_h, in_w, min_div=16):
    """(left, right, top, bottom) pad and the matching unpad -- inference_high_resolution.py:145-173"""
    out_h = (in_h + min_div - 1) // min_div * min_div
    out_w = (in_w + min_div - 1) // min_div * min_div
    left = (out_w - in_w) // 2
    right = out_w - in_w - left
    top = (out_h - in_h) // 2
    bottom = out_h - in_h - top
    return (left, right, top, bottom), (-left, -right, -top, -bottom)


def center_crop16(x, div=16):
    """[..., H, W] -> the centred [..., div*(H//div), div*(W//div)] window: the dataset transform of both reference scripts
    (`_resize_and_crop`, inference.py:62-68 = inference_high_resolution.py:62-68: torchvision's center_crop, whose offsets are
    int(round((H - H') / 2.0)) with Python's round-half-to-even).  A view, no copy"""
    H, W = x.shape[-2:]
    th, tw = div * (H // div), div * (W // div)
    top, left = int(round((H - th) / 2.0)), int(round((W - tw) / 2.0))
    return x[..., top:top + th, left:left + tw]


def tile_grid(h, w, tile=TILE):
    """row-major list of (y, x, tile_h, tile_w) -- nonoverlapping_grid_indices, :112-125 + the loop at :236-244"""
    ys = list(range(0, h, tile))
    xs = list(range(0, w, tile))
    ths = [tile] * (h // tile) + ([h % tile] if h % tile else [])
    tws = [tile] * (w // tile) + ([w % tile] if w % tile else [])
    return [(y, x, th, tw) for y, th in zip(ys, ths) for x, tw in zip(xs, tws)]


def gaussian_weights(tile_width, tile_height, device=None):
    """[1,3,tile_h,tile_w] float64 blend weights -- _gaussian_weights, :127-143 (note the reference's
    asymmetric midpoints: (w-1)/2 for x, h/2 for y)"""
    var = 0.01
    mx = (tile_width - 1) / 2
    xp = [math.exp(-(x - mx) * (x - mx) / (tile_width * tile_width) / (2 * var)) / math.sqrt(2 * math.pi * var)
          for x in range(tile_width)]
    my = tile_height / 2
    yp = [math.exp(-(y - my) * (y - my) / (tile_height * tile_height) / (2 * var)) / math.sqrt(2 * math.pi * var)
          for y in range(tile_height)]
    wts = torch.tensor(yp, dtype=torch.float64)[:, None] * torch.tensor(xp, dtype=torch.float64)[None, :]
    return wts.to(device).expand(1, 3, tile_height, tile_width)


class TiledImage:
    """result of compress_tiled: per shape-group CompressedBatch + where each tile sits"""

    def __init__(self, image_hw, pad, tiles, groups):
        self.image_hw = image_hw          # (H, W) unpadded
        self.pad = pad                    # (left, right, top, bottom)
        self.tiles = tiles                # [(y, x, th, tw)] row-major, padded coordinates
        self.groups = groups              # [(tile indices, CompressedBatch, encode outputs)]

    def tile_bpp(self):
        bpp = [None] * len(self.tiles)
        for idxs, comp, _ in self.groups:
            for i, v in zip(idxs, comp.bpp()):
                bpp[i] = v
        return bpp

    def bpp(self):
        """bit_sum / (W * H) of the unpadded image -- inference_high_resolution.py:250,256"""
        bits = sum(b * tw * th for b, (_, _, th, tw) in zip(self.tile_bpp(), self.tiles))
        return bits / self.image_hw[1] / self.image_hw[0]

    def streams(self):
        """per tile (row-major) {stream name: bytes}"""
        out = [None] * len(self.tiles)
        for idxs, comp, _ in self.groups:
            for i, s in zip(idxs, comp.to_host()):
                out[i] = s
        return out


# ---- shape groups side by side.  A 2040x1356 image is 6 tiles in 4 shape groups of one or two tiles: each group's launches
# keep a handful of CUs busy, so the groups cost what the slowest one costs when they run on parallel streams (forked from
# and joined back into the caller's stream by events -- which is also what a hipGraph capture of the whole image records).
_side_streams = {}


def _tensors_of(obj):
    if torch.is_tensor(obj):
        yield obj
    elif isinstance(obj, (list, tuple)):
        for o in obj:
            yield from _tensors_of(o)
    elif hasattr(obj, "data") and hasattr(obj, "nbytes") and torch.is_tensor(getattr(obj, "data", None)):
        yield obj.data
        yield obj.nbytes


class _Fork:
    """run the bodies of `for k, st in fork.lanes(n)` on n streams: lane 0 is the current stream, the others wait for what was
    queued on it so far; join() makes the current stream wait for all of them and tells the allocator about the hand-over"""

    def __init__(self, device, enabled):
        self.cur = torch.cuda.current_stream(device)
        self.enabled = enabled
        self.used = []
        self.device = device
        # ONE fork point, recorded before lane 0's body is queued: a side lane waits for what precedes the fork, not for the
        # (largest) group that lane 0 runs -- recorded lazily, the side lanes only started once that group was over
        self.fork_ev = None
        if enabled:
            self.fork_ev = torch.cuda.Event()
            self.fork_ev.record(self.cur)

    def on(self, k):
        """context of lane k's stream (no context switch at all when the fork is off: ~10 us of torch per use)"""
        return torch.cuda.stream(self.lane(k)) if self.enabled else _lib._null_ctx

    def lane(self, k):
        if not self.enabled or k == 0:
            return self.cur
        pool = _side_streams.setdefault(torch.device(self.device).index or 0, [])
        while len(pool) < k:
            pool.append(torch.cuda.Stream(self.device))
        st = pool[k - 1]
        if st not in self.used:
            st.wait_event(self.fork_ev)              # everything before the fork
            self.used.append(st)
        return st

    def join(self, produced=()):
        for st in self.used:
            ev = torch.cuda.Event()
            ev.record(st)
            self.cur.wait_event(ev)
        if self.used:
            for t in _tensors_of(list(produced)):
                t.record_stream(self.cur)            # allocated on a side stream, consumed (and eventually freed) on this one
        self.used = []


def _cut_all(x, frames, N, H, W, top, left, tiles, order):
    """the tile batches of all shape groups from the unpadded image(s) in one launch -> [batch per group], batch
    [N, T, 3, th, tw] fp32 or [N, T, th, tw, 3] uint8 (inference_high_resolution.py:145-173 pad + :236-244 crop)"""
    _lib.require_device(x)
    n = sum(len(idxs) for _, idxs in order)
    desc = (_lib.Tile * n)()
    batches, j = [], 0
    for (th, tw), idxs in order:
        T = len(idxs)
        batch = torch.empty((N, T, th, tw, 3) if frames else (N, T, 3, th, tw), dtype=x.dtype, device=x.device)
        per = 3 * th * tw
        for k, i in enumerate(idxs):
            desc[j] = _lib.Tile(batch.data_ptr() + k * per * batch.element_size(), T * per, tiles[i][0] - top, tiles[i][1] - left, th, tw)
            j += 1
        batches.append(batch)
    with torch.cuda.device(x.device):
        _lib.call("cgic_cut_tiles", _lib.ptr(x), int(frames), N, H, W, n, desc, _lib.current_stream(x.device))
    return batches


def _compress_groups(x, encode, codec, tile, concurrent, chain=False, fuse_maps=True, frames_fp32=False):
    """x [N,3,H,W] fp32 -- or uint8 frames [N,H,W,3], the tiles then reach `encode` as uint8 [T,th,tw,3] (entropy_maps_u8 makes the
    fp32 tiles and the maps in one pass) --: the shape groups of N images of one size, each group ONE batch of N * T tiles (image-major)
    -> (H, W), pad, tiles, [(tile indices, CompressedBatch, (ind, masks, mode))]"""
    frames = x.dtype == torch.uint8
    if frames and x.shape[-1] != 3 or not frames and x.shape[1] != 3:
        raise ValueError(f"expected [N,3,H,W] (or uint8 [N,H,W,3]), got {x.dtype} {tuple(x.shape)}")
    N = x.shape[0]
    H, W = (x.shape[1], x.shape[2]) if frames else (x.shape[2], x.shape[3])
    pad, _ = compute_padding(H, W)
    left, right, top, bottom = pad
    tiles = tile_grid(H + top + bottom, W + left + right, tile)
    by_shape = {}
    for i, (_, _, th, tw) in enumerate(tiles):
        by_shape.setdefault((th, tw), []).append(i)
    groups = []
    # the largest group first: it is the long pole, and lane 0 (no fork latency) is its stream
    order = sorted(by_shape.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1])
    chain = chain and 1 < len(order) <= _lib.lib().cgic_group_max()
    fork = _Fork(x.device, concurrent and not chain)
    xv = x.permute(0, 3, 1, 2) if frames else x              # [N,3,H,W] view either way
    # chain: ONE launch chain for all shape groups (cgic_group_begin / _launch): every group's calls are recorded, then issued as
    # one launch per kernel whose grid is the concatenation of the groups' grids
    grp = _lib.launch_group(len(order), [N * len(idxs) * th * tw for (th, tw), idxs in order], x.device) if chain else None
    batches = []
    cut, made = None, []
    if chain and (not frames or frames_fp32) and fuse_maps and max(len(idxs) for _, idxs in order) <= 48:
        # pad + crop + BOTH entropy maps of all shape groups in ONE launch (cgic_entropy_maps_tiles, grouped): the tile batches come
        # back tagged with their maps -- encode's entropy_maps(tiles) returns them without another pass over the pixels
        from .entropy import entropy_maps_tiles
        xc = x.contiguous()
        cut, made = [], []
        with _lib.launch_group(len(order), [N * len(idxs) * th * tw for (th, tw), idxs in order], x.device) as g:
            for k, ((th, tw), idxs) in enumerate(order):
                g.select(k)
                t, e8_, e16_ = entropy_maps_tiles(xc, [(tiles[i][0] - top, tiles[i][1] - left) for i in idxs], th, tw)
                made.append((e8_, e16_, e8_._cgic_flat8))           # (strong references for the length of this call)
                cut.append(t.view(N, len(idxs), 3, th, tw))
    elif not fork.enabled and len(tiles) <= 96:
        # pad + crop of ALL tiles as one launch (cgic_cut_tiles): every tile written straight from the unpadded image
        cut = _cut_all(x.contiguous(), frames, N, H, W, top, left, tiles, order)
    for lane, ((th, tw), idxs) in enumerate(order):
        with fork.on(lane):
            # pad + cut in ONE copy per tile (F.pad of the whole image and a stack of views would move every pixel twice): a tile
            # is the part of the image it covers, zeros where it reaches into the centred pad
            batch = cut[lane] if cut is not None else \
                torch.empty((N, len(idxs), th, tw, 3) if frames else (N, len(idxs), 3, th, tw), dtype=x.dtype, device=x.device)
            bv = batch.permute(0, 1, 4, 2, 3) if frames and not made else batch
            for k, i in enumerate(idxs if cut is None else ()):
                y0, x0 = tiles[i][0] - top, tiles[i][1] - left                  # in unpadded coordinates
                sy0, sy1, sx0, sx1 = max(y0, 0), min(y0 + th, H), max(x0, 0), min(x0 + tw, W)
                dst = bv[:, k]
                dst[:, :, sy0 - y0:sy1 - y0, sx0 - x0:sx1 - x0] = xv[:, :, sy0:sy1, sx0:sx1]
                # only the strips that reach into the pad are zeroed (a few rows / columns, not the whole batch)
                for strip in (dst[:, :, :sy0 - y0], dst[:, :, sy1 - y0:], dst[:, :, :, :sx0 - x0], dst[:, :, :, sx1 - x0:]):
                    if strip.numel():
                        strip.zero_()
            batch = batch.view(-1, th, tw, 3) if frames and not made else batch.view(-1, 3, th, tw)
            if made:                                     # (tags do not survive a view: this is the object `encode` sees)
                from .entropy import _MadeMaps
                e8, e16, flat8 = made[lane]
                batch._cgic_maps = _MadeMaps(batch, e8, e16, flat8)       # (weak: the maps point back at the batch)
                e8._cgic_pixels = e16._cgic_pixels = batch
            if chain:
                batches.append(batch)
                continue
            ind, masks, mode = encode(batch)
            groups.append((idxs, codec.compress(ind, masks, mode), (ind, masks, mode)))
    if chain:
        with grp as g:
            for k, (((th, tw), idxs), batch) in enumerate(zip(order, batches)):
                g.select(k)
                ind, masks, mode = encode(batch)
                groups.append((idxs, codec.compress(ind, masks, mode), (ind, masks, mode)))
    fork.join([(c, e) for _, c, e in groups])
    return (H, W), pad, tiles, groups


def compress_tiled(x, encode, codec, tile=TILE, concurrent=False, chain=False, fuse_maps=True, frames_fp32=False):
    """x [1,3,H,W] on the device; encode(tiles [T,3,th,tw]) -> (ind [T*h*w] int64, masks [3 x int32], mode)
    with per-tile routing (the reference's per-tile B=1 call); codec: GrainCodec.  -> TiledImage.
    concurrent: the shape groups run on parallel streams (same results; see _Fork).
    chain: the shape groups (four for a 2040x1356 image) go through ONE launch chain -- entropy maps, VQ + router, compress: three
    launches for all six tiles instead of three per group (cgic_group_begin / _launch; same bytes).  `encode` is then called
    inside a launch group: it may allocate and call control_gic_amd's entropy_maps / entropy_maps_u8 / vq_forward_route (whose
    launches are recorded and issued when all groups are in), but must not enqueue torch work that READS their outputs --
    a conv encoder that consumes the router's gate cannot run under chain=True.
    fuse_maps (with chain, fp32 input): the tiles are cut AND their entropy maps made in one pass over the image
    (cgic_entropy_maps_tiles); the tile batches `encode` receives carry their maps, entropy_maps(tiles) returns them as they are.
    frames_fp32 (uint8 frames with chain + fuse_maps): `encode` receives the fp32 tiles T.ToTensor() would have produced (tagged with their
    maps) instead of uint8 tiles -- ToTensor, pad, crop and both maps are then one pass over the frames (3 B read + 12 B written per pixel)."""
    if x.dim() != 4 or x.shape[0] != 1:
        raise ValueError("compress_tiled takes one image [1,3,H,W] (or one uint8 frame [1,H,W,3]; the reference script uses batch 1); "
                         "compress_tiled_batch takes several of one size")
    return TiledImage(*_compress_groups(x, encode, codec, tile, concurrent, chain, fuse_maps, frames_fp32))


def compress_tiled_batch(x, encode, codec, tile=TILE, concurrent=False, chain=False, fuse_maps=True, frames_fp32=False):
    """x [N,3,H,W] (or uint8 frames [N,H,W,3]: `encode` then gets uint8 tiles [T,th,tw,3] for entropy_maps_u8 -- a pad of zero
    bytes is the pad of zeros ToTensor would have produced): N images of ONE size (a folder of camera frames, a DIV2K bucket) -> list of N TiledImage, each what
    compress_tiled gives for that image alone (routing is per tile, so batching across images changes no byte).  The tiles
    of equal shape of ALL the images go through the kernels as one batch: a 2040x1356 image alone is four launch chains of one or
    two tiles each (a handful of workgroups per launch); eight images are chains of 8-16 tiles.  The TiledImages hold views
    of the shared per-group buffers"""
    if x.dim() != 4:
        raise ValueError("compress_tiled_batch takes [N,3,H,W] or uint8 [N,H,W,3]")
    N = x.shape[0]
    hw, pad, tiles, groups = _compress_groups(x, encode, codec, tile, concurrent, chain, fuse_maps, frames_fp32)
    out = []
    for n in range(N):
        mine = []
        for idxs, comp, (ind, masks, mode) in groups:
            T = len(idxs)
            sl = slice(n * T, (n + 1) * T)
            per = ind.numel() // (N * T)
            mine.append((idxs, type(comp)(comp.data[sl], comp.nbytes[sl], comp.mode, comp.h, comp.w),
                         (ind.view(N * T, per)[sl].reshape(-1), [m[sl] for m in masks], mode)))
        t = TiledImage(hw, pad, tiles, mine)
        t._whole = (groups, n, N)                 # decompress_tiled_batch of the whole list reads the shared buffers in place
        out.append(t)
    return out


class TiledCall:
    """The tiled driver (inference_high_resolution.py:236-257: one compress() per tile) for images of ONE size as ONE foreign call
    per image batch (cgic_compress_tiled): pad + crop + both entropy maps, VQ + per-tile router, stream coder and -- decode=True --
    decoder + merge, every link one launch for all shape groups, over buffers allocated once.  Hot path only: the latent of every
    shape group is the caller's (`zs`).  The results live in this object's buffers: valid until the next call.
    -> __call__(x, zs): x [N,3,H,W] fp32 or uint8 frames [N,H,W,3]; zs: per shape group (self.groups order: (th, tw), tile indices) the
    latent [N*T,4,th/4,tw/4]; returns a TiledImage (N == 1) or a list of N, .decoded = per group (ind, masks, z_q, status)."""

    def __init__(self, quantizer, coarse_ratio, medium_ratio, N, H, W, frequency=None, decode=True, frames=False, decoder=None, tile=TILE,
                 prepare=True):
        import ctypes
        from .codec import GrainCodec, _decoder_flag
        from .quantize import prepare_codebook
        w = quantizer.embedding.weight
        dev = w.device
        _lib.require_device(w)
        self.vq, self.dev, self.N, self.H, self.W, self.frames = quantizer, dev, N, H, W, bool(frames)
        self.ratios = (float(coarse_ratio), float(medium_ratio))
        self.codec = GrainCodec(frequency if frequency is not None else quantizer.embedding_counter, w)
        self.prepared = prepare_codebook(w) if prepare else None
        self._prepared_version = w._version
        self.decoder = _decoder_flag(decoder)
        self.pad, _ = compute_padding(H, W)
        left, right, top, bottom = self.pad
        self.tiles = tile_grid(H + top + bottom, W + left + right, tile)
        by_shape = {}
        for i, (_, _, th, tw) in enumerate(self.tiles):
            by_shape.setdefault((th, tw), []).append(i)
        self.groups = sorted(by_shape.items(), key=lambda kv: -len(kv[1]) * kv[0][0] * kv[0][1])
        l = _lib.lib()
        if len(self.groups) > l.cgic_group_max():
            raise ValueError(f"{len(self.groups)} tile shapes; cgic_compress_tiled takes at most {l.cgic_group_max()}")
        f32, i32, i64, u8t = torch.float32, torch.int32, torch.int64, torch.uint8
        E = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        total = sum(len(ix) * th * tw for (th, tw), ix in self.groups)
        self._arr = (_lib.TileGroup * len(self.groups))()
        self._buf, self._keep = [], []
        p = _lib.ptr
        for k, ((th, tw), idxs) in enumerate(self.groups):
            T, B, h, ww = len(idxs), N * len(idxs), th // 4, tw // 4
            slot = self.codec.slot_bytes(h, ww)
            t = {"x": E((B, 3, th, tw), f32), "e8": E((B, th // 8, tw // 8), f32), "e16": E((B, th // 16, tw // 16), f32),
                 "flat8": E((B, th // 8, tw // 8), f32), "ind": E((B * h * ww,), i64),
                 "mask": [E((B, 1, h // 4, ww // 4), i32), E((B, 1, h // 2, ww // 2), i32), E((B, 1, h, ww), i32)],
                 "data": E((B, _lib.NUM_STREAMS, slot), u8t), "nbytes": E((B, _lib.NUM_STREAMS), i32), "slot": slot, "h": h, "w": ww}
            if decode:
                t.update({"dind": E((B, h, ww), i64), "dz_q": E((B, 4, h, ww), f32), "status": E((B,), i32),
                          "dmask": [E((B, 1, h // 4, ww // 4), i32), E((B, 1, h // 2, ww // 2), i32), E((B, 1, h, ww), i32)]})
            ws_c = E((max(1, l.cgic_compress_workspace_bytes(B, h, ww)),), u8t)
            ws_d = E((l.cgic_decompress_workspace_bytes(B, h, ww),), u8t) if decode else None
            if not l.cgic_router_refine_in_lds(B, th // 16, tw // 16, 1):
                raise ValueError(f"TiledCall: tiles of {th}x{tw} are routed as one segment beyond the router workgroup's LDS; their refinement is a chain "
                                 "of launches that a launch group cannot record (cgic_router_refine_in_lds): use tiles of at most 768x768 "
                                 "(the reference's), or compress_tiled(..., chain=False)")
            nref = l.cgic_router_refine_scratch_bytes(B, th // 16, tw // 16, 1) if ((th // 16) * (tw // 16) >= _lib.REFINE_SPLIT_MIN_PATCHES or _lib.REFINE_FUSED_QUEUES) and _lib.REFINE_QUEUES else 0
            ws_r = E((nref,), u8t) if nref else None                 # (large tiles: their row bands split a threshold band between them)
            org = (ctypes.c_int * (2 * T))(*[v for i in idxs for v in (self.tiles[i][0] - top, self.tiles[i][1] - left)])
            g = self._arr[k]
            g.ntiles, g.th, g.tw, g.origins, g.share = T, th, tw, org, len(idxs) * th * tw / total
            io = g.io
            io.x_out, io.e8, io.e16, io.flat8, io.ind = p(t["x"]), p(t["e8"]), p(t["e16"]), p(t["flat8"]), p(t["ind"])
            io.mask_c, io.mask_m, io.mask_f = (p(m) for m in t["mask"])
            io.streams, io.slot, io.nbytes = p(t["data"]), slot, p(t["nbytes"])
            if decode:
                io.dind, io.dz_q, io.status = p(t["dind"]), p(t["dz_q"]), p(t["status"])
                io.dmask_c, io.dmask_m, io.dmask_f = (p(m) for m in t["dmask"])
            io.ws_compress, io.ws_decompress = p(ws_c), p(ws_d)
            io.ws_refine, io.ws_refine_bytes = p(ws_r), nref
            self._buf.append(t)
            self._keep += [org, ws_c, ws_d, ws_r]
        self._mode = ctypes.c_int(0)
        self._fn = l.cgic_compress_tiled
        self._bins = _lib.linspace_bins()
        self._decode = bool(decode)
        self.decoded = None

    def refresh_codebook(self):
        """after changing embedding.weight: rewrite the codebook image in place (HotCall / HotPathPipeline have the same method; __call__
        does it by itself when the weight tensor's version counter moved)"""
        from .quantize import prepare_codebook
        if self.prepared is not None:
            self.prepared = prepare_codebook(self.vq.embedding.weight, out=self.prepared)
            self._prepared_version = self.vq.embedding.weight._version

    def __call__(self, x, zs):
        import ctypes
        from .codec import CompressedBatch
        N, H, W = self.N, self.H, self.W
        if tuple(x.shape) != ((N, H, W, 3) if self.frames else (N, 3, H, W)) or x.dtype != (torch.uint8 if self.frames else torch.float32) \
                or not x.is_contiguous() or x.device != self.dev or len(zs) != len(self.groups):
            raise ValueError("TiledCall: x / zs do not have the shape, dtype, device or layout this object was built for")
        for k, (((th, tw), idxs), z) in enumerate(zip(self.groups, zs)):
            if tuple(z.shape) != (N * len(idxs), 4, th // 4, tw // 4) or z.dtype != torch.float32 or not z.is_contiguous() or z.device != self.dev:
                raise ValueError(f"TiledCall: latent of shape group {k} must be fp32 [{N * len(idxs)},4,{th // 4},{tw // 4}]")
            self._arr[k].io.z = z.data_ptr()
        w = self.vq.embedding.weight
        if self.prepared is not None and w._version != self._prepared_version:
            self.refresh_codebook()                 # the weights were modified in place since the image was made: a stale image gives wrong indices

        def go():
            _lib.check(self._fn(self.codec.huffman.table.handle, w.data_ptr(), w.shape[0], w.shape[1], _lib.ptr(self.prepared), x.data_ptr(),
                                int(self.frames), N, H, W, len(self.groups), self._arr, self.ratios[0], self.ratios[1], float(self.vq.beta),
                                int(bool(self.vq.legacy)), self._bins, 32, 0.01, self.decoder, ctypes.byref(self._mode),
                                torch.cuda.current_stream(self.dev).cuda_stream))
        if torch.cuda.current_device() == self.dev.index:
            go()
        else:
            with torch.cuda.device(self.dev):
                go()
        mode = self._mode.value
        groups = [(idxs, CompressedBatch(t["data"], t["nbytes"], mode, t["h"], t["w"]), (t["ind"], t["mask"], mode))
                  for (_, idxs), t in zip(self.groups, self._buf)]
        self.decoded = [(t["dind"], t["dmask"], t["dz_q"], t["status"]) for t in self._buf] if self._decode else None
        if N == 1:
            return TiledImage((H, W), self.pad, self.tiles, groups)
        out = []
        for n in range(N):
            mine = []
            for idxs, comp, (ind, masks, mode_) in groups:
                T = len(idxs)
                sl = slice(n * T, (n + 1) * T)
                per = ind.numel() // (N * T)
                mine.append((idxs, CompressedBatch(comp.data[sl], comp.nbytes[sl], comp.mode, comp.h, comp.w),
                             (ind.view(N * T, per)[sl].reshape(-1), [m[sl] for m in masks], mode_)))
            out.append(TiledImage((H, W), self.pad, self.tiles, mine))
        return out


def decompress_tiled_batch(tiled_list, codec, concurrent=False, check=True, chain=False, decoder=None):
    """inverse of compress_tiled_batch for TiledImages of one geometry (from it, or from N compress_tiled calls on images
    of one size, or rebuilt from containers): ONE decompress per shape group over all the images
    -> list (per image) of per-tile (ind, masks, z_q); check=False: (that, [N * tiles] status tensor)"""
    if not tiled_list:
        return []
    first = tiled_list[0]
    for t in tiled_list[1:]:
        if t.tiles != first.tiles or [g[0] for g in t.groups] != [g[0] for g in first.groups] or \
                any(a[1].mode != b[1].mode for a, b in zip(t.groups, first.groups)):
            raise ValueError("decompress_tiled_batch: the images differ in geometry or routing mode")
    N = len(tiled_list)
    per_image = [[None] * len(first.tiles) for _ in range(N)]
    dev = first.groups[0][1].data.device
    chain = chain and 1 < len(first.groups) <= _lib.lib().cgic_group_max()
    fork = _Fork(dev, concurrent and not chain)
    outs, statuses = [], []
    pending = []
    whole = getattr(first, "_whole", None)
    if whole is not None and not (whole[2] == N and all(
            getattr(t, "_whole", (None,))[0] is whole[0] and t._whole[1] == n for n, t in enumerate(tiled_list))):
        whole = None
    for lane, (idxs, c0, _) in enumerate(first.groups):
        with fork.on(lane):
            if whole is not None:
                comp = whole[0][lane][1]
            else:
                comps = [t.groups[lane][1] for t in tiled_list]
                slot = max(c.data.shape[-1] for c in comps)
                data = torch.cat([c.data if c.data.shape[-1] == slot else
                                  torch.nn.functional.pad(c.data, (0, slot - c.data.shape[-1])) for c in comps])
                comp = type(c0)(data, torch.cat([c.nbytes for c in comps]), c0.mode, c0.h, c0.w)
            if chain:
                pending.append(comp)
                continue
            outs.append(codec.decompress(comp, decoder=decoder))
    status_all = None
    if chain:
        # the decoder and the merge of all shape groups as ONE launch each (see compress_tiled); one status buffer for all of them
        status_all = torch.empty(sum(c.batch for c in pending), dtype=torch.int32, device=dev)
        at = 0
        with _lib.launch_group(len(pending), [c.batch * c.h * c.w for c in pending], dev) as g:
            for k, comp in enumerate(pending):
                g.select(k)
                outs.append(codec.decompress(comp, status=status_all[at:at + comp.batch], decoder=decoder))
                at += comp.batch
    for (idxs, _, _), (ind, masks, zq, status) in zip(first.groups, outs):
        statuses.append(status)
        T = len(idxs)
        for n in range(N):
            for k, i in enumerate(idxs):
                j = n * T + k
                per_image[n][i] = (ind[j:j + 1], [m[j:j + 1] for m in masks], zq[j:j + 1])
    fork.join(outs)
    all_status = status_all if status_all is not None else torch.cat(statuses)
    if not check:
        return per_image, all_status
    if int(all_status.abs().max()) != 0:
        raise RuntimeError("corrupt tile stream")
    return per_image


def decompress_tiled(tiled, codec, decode=None, concurrent=False, check=True, chain=False, decoder=None):
    """-> per-tile (ind, masks, z_q) in row-major order; with decode(z_q, masks) -> pixels also the blended,
    clamped, unpadded reconstruction (:248-255; tiles do not overlap, so the weights cancel).
    concurrent: shape groups on parallel streams; check=False skips the host synchronisation on the decoder status (for
    stream capture): the second result is then the [tiles] status tensor to look at later.
    decoder: "latency" / "throughput" / "auto" (GrainCodec.decompress).  "latency" = this call has the GPU to itself: with chain the
    decoder and the merge of ALL shape groups are then one launch; the default keeps that form to launches of at most half the
    chip (several decode chains in flight on other streams must never leave a merge band spinning on a decoder without a CU)."""
    per_tile = [None] * len(tiled.tiles)
    statuses = []
    dev = tiled.groups[0][1].data.device if tiled.groups else None
    chain = chain and dev is not None and 1 < len(tiled.groups) <= _lib.lib().cgic_group_max()
    fork = _Fork(dev, concurrent and dev is not None and not chain)
    outs = []
    # chain: the decoder and the merge of all shape groups as ONE launch each (see compress_tiled)
    status_all = None
    if chain:
        # (one status buffer for all groups: no concatenation kernel afterwards)
        status_all = torch.empty(sum(c.batch for _, c, _ in tiled.groups), dtype=torch.int32, device=dev)
        at = 0
        with _lib.launch_group(len(tiled.groups), [c.batch * c.h * c.w for _, c, _ in tiled.groups], dev) as g:
            for lane, (_, comp, _) in enumerate(tiled.groups):
                g.select(lane)
                outs.append(codec.decompress(comp, status=status_all[at:at + comp.batch], decoder=decoder))
                at += comp.batch
    for lane, (idxs, comp, _) in enumerate(tiled.groups):
        if not chain:
            with fork.on(lane):
                outs.append(codec.decompress(comp, decoder=decoder))
        ind, masks, zq, status = outs[lane]
        statuses.append(status)
        for k, i in enumerate(idxs):
            per_tile[i] = (ind[k:k + 1], [m[k:k + 1] for m in masks], zq[k:k + 1])
    fork.join(outs)
    all_status = status_all if status_all is not None else (torch.cat(statuses) if statuses else None)
    if not check:
        return per_tile, all_status
    # ONE host synchronisation for the whole image (not one per shape group)
    if all_status is not None and int(all_status.abs().max()) != 0:
        raise RuntimeError("corrupt tile stream")
    if decode is None:
        return per_tile, None
    H, W = tiled.image_hw
    left, right, top, bottom = tiled.pad
    dev = per_tile[0][2].device
    rec = torch.zeros((1, 3, H + top + bottom, W + left + right), device=dev)
    contrib = torch.zeros_like(rec)
    for (y, x, th, tw), (ind, masks, zq) in zip(tiled.tiles, per_tile):
        wts = gaussian_weights(tw, th, dev)
        rec[:, :, y:y + th, x:x + tw] += decode(zq, masks) * wts                 # :248 (float32 += float64 product)
        contrib[:, :, y:y + th, x:x + tw] += wts
    rec = (rec / contrib).clamp(0, 1)
    return per_tile, rec[:, :, top:top + H, left:left + W]
