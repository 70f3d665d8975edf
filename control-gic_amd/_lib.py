"""ctypes binding of libcgic_hip.so (include/cgic_hip.h).

There is no CPU fallback anywhere in this package: if the shared library is
missing or a call fails, the caller gets an exception, never a silently
different code path.
"""
import ctypes as C
import os
import threading

_HERE = os.path.dirname(os.path.abspath(__file__))
# CGIC_LIB lets dev tools load an instrumented build of the SAME sources (make -C csrc dbg)
LIB_PATH = os.environ.get("CGIC_LIB") or os.path.join(_HERE, "libcgic_hip.so")

OK, ERR_INVALID, ERR_UNSUPPORTED, ERR_HIP, ERR_NOMEM, ERR_CAPACITY = 0, -1, -2, -3, -4, -5
NUM_STREAMS = 5
STREAM_NAMES = ("indices_coarse", "indices_medium", "indices_fine", "mask_coarse", "mask_medium")


class CgicError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libcgic_hip error {code}: {msg}")
        self.code = code


_vp, _i64, _i32, _int, _f32, _f64, _sz = C.c_void_p, C.c_int64, C.c_int32, C.c_int, C.c_float, C.c_double, C.c_size_t


class Conv1x1(C.Structure):
    """struct cgic_conv1x1 (include/cgic_hip.h): a Conv2d(4, 4, 1) fused into a kernel"""
    _fields_ = [("weight", _vp), ("bias", _vp), ("bias_first", _int)]


_cv = C.POINTER(Conv1x1)


class Pixels(C.Structure):
    """struct cgic_pixels (include/cgic_hip.h): the image batch behind a pair of entropy maps, for the router's refinement"""
    _fields_ = [("x", _vp), ("is_u8", _int), ("bins", C.POINTER(_f32)), ("nbins", _int), ("sigma", _f32), ("flat8", _vp),
                ("scratch", _vp), ("scratch_bytes", _sz)]


_px = C.POINTER(Pixels)


class ImageIO(C.Structure):
    """struct cgic_image_io (include/cgic_hip.h): the buffers of one cgic_compress_image call"""
    _fields_ = [("x", _vp), ("x_is_u8", _int), ("z", _vp), ("x_out", _vp), ("e8", _vp), ("e16", _vp), ("flat8", _vp),
                ("ind", _vp), ("z_q", _vp), ("loss", _vp), ("mask_c", _vp), ("mask_m", _vp), ("mask_f", _vp),
                ("streams", _vp), ("slot", _i64), ("nbytes", _vp), ("hist", _vp),
                ("dind", _vp), ("dmask_c", _vp), ("dmask_m", _vp), ("dmask_f", _vp), ("dz_q", _vp), ("status", _vp),
                ("ws_vq", _vp), ("ws_compress", _vp), ("ws_decompress", _vp), ("ws_refine", _vp), ("ws_refine_bytes", _sz)]


class TileGroup(C.Structure):
    """struct cgic_tile_group (include/cgic_hip.h): one shape group of cgic_compress_tiled"""
    _fields_ = [("ntiles", _int), ("th", _int), ("tw", _int), ("origins", C.POINTER(_int)), ("share", _f64), ("io", ImageIO)]


class Tile(C.Structure):
    """struct cgic_tile (include/cgic_hip.h): one destination tile of cgic_cut_tiles"""
    _fields_ = [("dst", _vp), ("image_stride", _i64), ("y0", _int), ("x0", _int), ("th", _int), ("tw", _int)]


# name -> (restype, argtypes); every function include/cgic_hip.h declares
PROTOTYPES = {
    "cgic_last_error": (C.c_char_p, []),
    "cgic_abi_version": (_int, []),
    "cgic_set_decode_mode": (_int, [_int]),
    "cgic_decode_stats": (_int, [_vp]),
    "cgic_device_count": (_int, []),
    "cgic_launch_graphs": (_int, [C.POINTER(_vp), C.POINTER(_vp), _int]),
    "cgic_group_max": (_int, []),
    "cgic_group_begin": (_int, [_int, C.POINTER(C.c_double)]),
    "cgic_group_select": (_int, [_int]),
    "cgic_group_launch": (_int, [_vp]),
    "cgic_group_abort": (None, []),
    "cgic_ticket_scope_begin": (_int, []),
    "cgic_ticket_scope_end": (_int, []),
    "cgic_ticket_scope_release": (_int, [_int]),
    "cgic_ticket_slots_in_use": (_int, []),
    "cgic_ticket_pool_dirty_words": (C.c_longlong, []),
    "cgic_ticket_pool_dirty_dump": (_int, [_vp, _int]),
    "cgic_vq_stats": (_int, [_vp]),
    "cgic_vq_filter_probe_f32": (_int, [_vp, _i64, _i64, _vp, _int, _vp, _vp, _vp, _vp]),
    "cgic_vq_workspace_bytes": (_sz, [_i64]),
    "cgic_conv1x1_rows_f32": (_int, [_vp, _i64, _cv, _vp, _vp]),
    "cgic_vq_prepared_bytes": (_sz, [_int]),
    "cgic_vq_cluster_permutation_host": (_int, [_vp, _int, _vp]),
    "cgic_vq_prepare_f32": (_int, [_vp, _int, _int, _vp, _vp]),
    "cgic_vq_forward_f32": (_int, [_vp, _i64, _i64, _vp, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _cv, _vp, _vp]),
    "cgic_vq_forward_valu_f32": (_int, [_vp, _i64, _i64, _vp, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _cv, _vp]),
    "cgic_vq_forward_route_f32": (_int, [_vp, _i64, _i64, _vp, _int, _int, _f32, _int, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i64,
                                         _f64, _f64, _int, _vp, _vp, _vp, _vp, C.POINTER(_int), _cv, _vp, _px, _vp]),
    "cgic_vq_backward_workspace_bytes": (_sz, [_i64, _int]),
    "cgic_vq_backward_f32": (_int, [_vp, _i64, _i64, _vp, _int, _int, _vp, _vp, _vp, _f32, _int, _vp, _vp, _vp, _vp]),
    "cgic_index_histogram": (_int, [_vp, _i64, _int, _vp, _vp]),
    "cgic_entropy_maps_f32": (_int, [_vp, _i64, _i64, _i64, C.POINTER(_f32), _int, _f32, _vp, _vp, _vp, _vp]),
    "cgic_entropy_maps_ref_f32": (_int, [_vp, _i64, _i64, _i64, C.POINTER(_f32), _int, _f32, _vp, _vp, _vp]),
    "cgic_entropy_maps_u8": (_int, [_vp, _i64, _i64, _i64, C.POINTER(_f32), _int, _f32, _vp, _vp, _vp, _vp, _vp]),
    "cgic_router_mode": (_int, [_f64, _f64]),
    "cgic_compress_image": (_int, [_vp, _vp, _int, _int, _vp, _i64, _i64, _i64, _f64, _f64, _f32, _int, C.POINTER(_f32), _int, _f32, _int,
                                  C.POINTER(ImageIO), C.POINTER(_int), _vp]),
    "cgic_compress_tiled": (_int, [_vp, _vp, _int, _int, _vp, _vp, _int, _i64, _i64, _i64, _int, C.POINTER(TileGroup), _f64, _f64, _f32, _int,
                                  C.POINTER(_f32), _int, _f32, _int, C.POINTER(_int), _vp]),
    "cgic_router_refine_supported": (_int, [_i64, _i64, _i64, _int]),
    "cgic_router_refine_in_lds": (_int, [_i64, _i64, _i64, _int]),
    "cgic_router_refine_scratch_bytes": (_sz, [_i64, _i64, _i64, _int]),
    "cgic_router_f32": (_int, [_vp, _vp, _i64, _i64, _i64, _f64, _f64, _int, _vp, _vp, _vp, _vp, C.POINTER(_int), _px, _vp]),
    "cgic_table_create": (_int, [C.POINTER(_i64), C.POINTER(_i32), _int, C.POINTER(_vp)]),
    "cgic_table_binary": (_int, [C.POINTER(_vp)]),
    "cgic_table_destroy": (None, [_vp]),
    "cgic_table_num_symbols": (_int, [_vp]),
    "cgic_table_max_len": (_int, [_vp]),
    "cgic_table_words": (_int, [_vp]),
    "cgic_table_get": (_int, [_vp, C.POINTER(_i32), C.POINTER(C.c_uint32)]),
    "cgic_stream_capacity": (_sz, [_vp, _i64]),
    "cgic_stream_workspace_bytes": (_sz, [_i64]),
    "cgic_encode_stream": (_int, [_vp, _vp, _int, _i64, _vp, _i64, _vp, _vp, _vp]),
    "cgic_decode_stream": (_int, [_vp, _vp, _i64, _vp, _i64, _vp, _vp]),
    "cgic_compress_slot_bytes": (_sz, [_vp, _i64, _i64]),
    "cgic_compress_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cgic_mode_streams": (_int, [_int]),
    "cgic_compress_streams": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _i64, _i64, _int, _vp, _i64, _vp, _vp, _vp, _vp]),
    "cgic_decompress_workspace_bytes": (_sz, [_i64, _i64, _i64]),
    "cgic_decompress_streams": (_int, [_vp, _vp, _i64, _vp, _i64, _i64, _i64, _int, _vp, _vp, _vp, _vp, _vp, _int,
                                       _int, _vp, _vp, _vp, _vp, _vp, _int, _vp]),
    "cgic_grain_merge_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _vp, _vp]),
    "cgic_avgpool_f32": (_int, [_vp, _i64, _i64, _i64, _int, _vp, _vp]),
    "cgic_cut_tiles": (_int, [_vp, _int, _i64, _i64, _i64, _int, C.POINTER(Tile), _vp]),
    "cgic_entropy_maps_tiles": (_int, [_vp, _int, _i64, _i64, _i64, _int, C.POINTER(_int), _i64, _i64, C.POINTER(_f32), _int, _f32, _vp, _vp, _vp,
                                       _vp, _vp]),
    "cgic_decoder_blend_medium_f32": (_int, [_vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _vp, _vp]),
    "cgic_decoder_blend_fine_f32": (_int, [_vp, _vp, _vp, _vp, _vp, _i64, _int, _i64, _i64, _vp, _vp]),
    "cgic_embedding_gather_f32": (_int, [_vp, _i64, _i64, _vp, _int, _int, _vp, _vp, _vp]),
}

_lib = None


def lib():
    """Load libcgic_hip.so (once).  Raises if it has not been built -- run
    `python -c "import __graft_entry__ as g; g.build()"` or `make -C control-gic_amd/csrc`."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: the HIP extension has not been built "
                "(make -C control-gic_amd/csrc).  There is no CPU fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(l, name)
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc):
    if rc < 0:
        raise CgicError(rc, lib().cgic_last_error().decode(errors="replace"))
    return rc


def call(name, *args):
    return check(getattr(lib(), name)(*args))


class launch_group:
    """`with launch_group(n, shares) as g:` -- independent sub-batches of different shapes through ONE launch per kernel
    (cgic_group_begin / _select / _launch, include/cgic_hip.h): inside the block the entropy / VQ+router / compress / decompress
    calls of this thread record their launches; `g.select(k)` names the group the following calls belong to; leaving the block
    issues the recorded launches position by position on `device`'s current stream.  Outputs are valid (in stream order) after the
    block; no torch work may consume them inside it."""

    def __init__(self, n, shares=None, device=None):
        self.n, self.shares, self.device, self.launches = int(n), shares, device, None

    def __enter__(self):
        arr = None
        if self.shares is not None:
            tot = float(sum(self.shares))
            arr = (C.c_double * self.n)(*[max(float(v) / tot, 1e-6) for v in self.shares])
        import torch
        self._dev_ctx = torch.cuda.device(self.device)
        self._dev_ctx.__enter__()
        try:
            call("cgic_group_begin", self.n, arr)
            _group_tls.stream = torch.cuda.current_stream(self.device).cuda_stream
            _group_tls.device_index = torch.cuda.current_device()
        except BaseException:
            self._dev_ctx.__exit__(None, None, None)
            raise
        _group_tls.keep = []
        return self

    def select(self, k):
        call("cgic_group_select", int(k))

    def __exit__(self, et, ev, tb):
        try:
            if et is not None:
                lib().cgic_group_abort()
                return False
            self.launches = call("cgic_group_launch", _group_tls.stream)
        except BaseException:
            lib().cgic_group_abort()
            raise
        finally:
            _group_tls.keep = None          # (released in stream order: after the launches just enqueued)
            _group_tls.stream = None
            self._dev_ctx.__exit__(None, None, None)
        return False


_group_tls = threading.local()


def ptr(t):
    """device/host pointer of a torch tensor (None -> NULL).  Inside a launch_group block the tensor is also kept alive until the
    group has been launched: the launches are deferred, and a temporary released before them (a workspace, an output the caller
    drops) would be handed out again by the caching allocator while a recorded launch still points at it"""
    if t is None:
        return None
    keep = getattr(_group_tls, "keep", None)
    if keep is not None:
        keep.append(t)
    return t.data_ptr()


def conv_arg(conv, bias_first=False):
    """(ctypes pointer or None, keep-alive tuple) for a fused 1x1 convolution.  conv: None, a torch.nn.Conv2d(4, 4, 1) or a
    (weight, bias) pair; the tensors must stay alive until the call has been enqueued (the returned tuple holds them)."""
    if conv is None:
        return None, ()
    import torch
    weight, bias = (conv.weight, conv.bias) if hasattr(conv, "weight") else conv
    w = weight.detach().reshape(weight.shape[0], -1).contiguous().float()
    if tuple(w.shape) != (4, 4):
        raise NotImplementedError(f"fused 1x1 convolution: weight {tuple(weight.shape)}; Control-GIC's quant_conv / post_quant_conv are 4 -> 4 (model.py:51-52)")
    b = None if bias is None else bias.detach().contiguous().float()
    require_device(w, b)
    st = Conv1x1(w.data_ptr(), None if b is None else b.data_ptr(), int(bool(bias_first)))
    return C.byref(st), (st, w, b)


_BINS = None


def linspace_bins():
    """torch.linspace(-1, 1, 32) evaluated on the CPU like the reference's CPU path (model.py:480), as a ctypes float[32]"""
    global _BINS
    if _BINS is None:
        import torch
        _BINS = (C.c_float * 32)(*torch.linspace(-1, 1, 32, dtype=torch.float32).tolist())
    return _BINS


#: False: bands are evaluated inside the image's own router workgroup only (no refinement queues: the pre-ABI-7 behaviour; tests, A/B)
REFINE_QUEUES = True
#: the fused VQ + router launch: tiles of at least this many 16x16 patches (1024 = where a per-image segment gets row bands) get the
#: scratch that lets their row bands SPLIT a long threshold band between them (a smooth 768x768 tile 251 -> 82 us).  The launch then
#: runs the router in two attempts (plain first, the split instantiation only when a band is long: DESIGN.md 4.3): the ordinary
#: tile pays nothing measurable for it
#: the fused VQ + router launch with one workgroup per image (batches of 256x256 ...): images whose threshold band is long start over
#: with the launch's refinement queues and the routers that are done help (round 6: a smooth 8-bit batch no longer waits for the
#: image with the densest cluster).  False (the default for single calls): every image evaluates its band in its own workgroup -- the
#: kernel variant with the queues costs the ORDINARY launch ~1.5 us by itself (two more router instantiations in one kernel), so
#: it is chosen per stream of batches (pipeline.HotPathPipeline(refine_queues="auto"): from the content of the batches at capture)
REFINE_FUSED_QUEUES = os.environ.get("CGIC_REFINE_FUSED_QUEUES", "0") != "0"
REFINE_SPLIT_MIN_PATCHES = int(os.environ.get("CGIC_REFINE_SPLIT_MIN_PATCHES", "1024"))      # (a huge value: never)


def pixels_arg(pixels, B, h16, w16, per_image, sigma=0.01, flat8=None, queues=False, explicit=False):
    """(ctypes pointer or None, keep-alive) for the router's `refine` argument.  pixels: None, the fp32 [B,3,16 h16,16 w16] image
    batch the maps were made from, or the uint8 [B,16 h16,16 w16,3] frames; flat8: the constant-patch map the same entropy call
    made (entropy_maps(...) attaches it to its maps as `_cgic_flat8`), optional.  A routing segment that does not fit the router
    workgroup's LDS (flattened batches -- the reference's encode() semantics --, images beyond ~768x768 routed as one segment)
    is refined too (ABI 8): through patched copies of the maps in a scratch this function allocates.  queues: also hand over the scratch of the launch's refinement
    queues (cgic_pixels.scratch: the stand-alone router launch evaluates long bands with every idle wave of the launch)."""
    if pixels is None:
        return None, ()
    import torch
    require_device(pixels)
    u8 = pixels.dtype == torch.uint8
    want = (B, 16 * h16, 16 * w16, 3) if u8 else (B, 3, 16 * h16, 16 * w16)
    if tuple(pixels.shape) != want or (not u8 and pixels.dtype != torch.float32):
        raise ValueError(f"pixels {pixels.dtype} {tuple(pixels.shape)} do not belong to entropy maps of {B} x {h16} x {w16} "
                         f"(expected fp32 [B,3,16 h16,16 w16] or uint8 [B,16 h16,16 w16,3])")
    if not lib().cgic_router_refine_supported(B, h16, w16, int(bool(per_image))):
        # (a segment of 2^31 patches: not a practical case since ABI 8 -- segments beyond the LDS are refined through patched copies)
        msg = (f"threshold-band refinement is not available for a routing segment of {B if not per_image else 1} x {16 * h16}x{16 * w16} "
               "pixels (cgic_router_refine_supported): masks are made from the default entropy maps as given (within 2e-6 of the reference's "
               "arithmetic; under a strict '<' a tie-heavy image can get a few other mask elements).  Route per image "
               "(per_image=True, tiles of at most 768x768), or make the maps with entropy_maps(..., reference_order=True)")
        if explicit:
            raise ValueError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=3)
        return None, ()
    px = pixels.contiguous()
    if flat8 is not None:
        require_device(flat8)
        if tuple(flat8.shape) != (B, 2 * h16, 2 * w16) or flat8.dtype != torch.float32:
            raise ValueError(f"flat8 {flat8.dtype} {tuple(flat8.shape)} does not belong to these maps")
        flat8 = flat8.contiguous()
    # scratch of the launch's refinement queues (long bands are evaluated by every idle workgroup of the launch): an ordinary
    # temporary of the call -- the caching allocator keeps it alive for the stream, a captured graph keeps its own
    # ... and, for a segment that does not fit the router workgroup's LDS (the reference's flattened-batch routing, an untiled large
    # image), the patched copies of the maps its refinement works on: required there
    big = not lib().cgic_router_refine_in_lds(B, h16, w16, int(bool(per_image)))
    nbytes = lib().cgic_router_refine_scratch_bytes(B, h16, w16, int(bool(per_image))) if ((queues and REFINE_QUEUES) or big) else 0
    scratch = torch.empty(nbytes, dtype=torch.uint8, device=px.device) if nbytes else None
    st = Pixels(ptr(px), int(u8), linspace_bins(), 32, float(sigma), ptr(flat8), ptr(scratch), nbytes)
    return C.byref(st), (st, px, flat8, scratch)


class ticket_scope:
    """`with ticket_scope() as sc:` around the capture of a hipGraph (on this thread): the ticket slots its launches take are
    tagged; `sc.release()` -- or `sc.release_with(obj)`: when `obj`, the graph object, is garbage-collected -- returns them
    to the library's pool (cgic_ticket_scope_begin / _end / _release in include/cgic_hip.h)"""

    def __init__(self):
        self.id = 0

    def __enter__(self):
        self.id = call("cgic_ticket_scope_begin")
        return self

    def __exit__(self, *exc):
        call("cgic_ticket_scope_end")
        return False

    def release(self):
        if self.id:
            lib().cgic_ticket_scope_release(self.id)
            self.id = 0

    def release_with(self, obj):
        """give the slots back once `obj` (the graph object) is gone AND its last replay has finished.  Destroying a graph does
        not wait for a replay that is still in flight, and a slot handed to a new launch while the old graph's kernels still
        use it corrupts the shared ticket: the finalizer therefore only QUEUES the scope; flush_released() -- called by
        pipeline.capture_graph before every capture, the only consumer of pool slots -- synchronises the device and
        releases what is queued."""
        import weakref
        f = weakref.finalize(obj, _released.append, self.id)
        f.atexit = False
        return obj


_released = []          # scope ids whose graph object died (list.append is atomic; drained by flush_released)


def flush_released(device=None):
    """return the ticket slots of graphs that were garbage-collected since the last call.  Synchronises `device` first if there
    is anything to return (a replay of a dead graph may still be running); must not be called while a stream is capturing."""
    if not _released:
        return 0
    import torch
    torch.cuda.synchronize(device)
    n = 0
    while _released:
        r = lib().cgic_ticket_scope_release(_released.pop())
        n += max(r, 0)
    return n


def current_stream(device=None):
    import torch
    cached = getattr(_group_tls, "stream", None)          # inside a launch_group block: looked up once (the calls only record)
    if cached is not None:
        return cached
    return torch.cuda.current_stream(device).cuda_stream


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


_null_ctx = _NullCtx()


def on_device(device):
    """`with on_device(dev):` == `with torch.cuda.device(dev):`, except inside a launch_group block, which holds its device for
    the whole block (the ~10 us of the torch context manager per recorded call were a third of the host time of a tiled image)"""
    if getattr(_group_tls, "stream", None) is not None:
        want, have = getattr(device, "index", None), getattr(_group_tls, "device_index", None)
        if want is not None and have is not None and want != have:
            raise RuntimeError(f"a launch group is open on cuda:{have}; tensors on cuda:{want} cannot be recorded into it")
        return _null_ctx
    import torch
    return torch.cuda.device(device)


def require_device(*tensors):
    """The product path is HIP-only: refuse CPU tensors loudly."""
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "control_gic_amd ops run on an MI355X (HIP) device only; got a CPU tensor. "
                "There is deliberately no CPU fallback -- the CPU oracle lives in oracle/ and is test-only.")
