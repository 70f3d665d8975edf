"""VectorQuantize2 / VectorQuantizer -- drop-in for the reference's quantizer
(CGIC/modules/vqvae/quantize.py:9-97; aliased VectorQuantizer at CGIC/models/model.py:14).

Same constructor, attributes, forward signature and state_dict keys; the
forward body is one fused HIP kernel (csrc/cgic_vq.hip) instead of an N x K
distance matrix in HBM.
"""
from collections.abc import MutableMapping

import torch
from torch import nn

from . import _lib


class _CounterView(MutableMapping):
    """`embedding_counter` as the reference exposes it (quantize.py:28): a mapping
    str(i) -> 1-element fp32 tensor (``.item()``, ``+= 1``, ``.data.fill_()`` all work),
    iterating its keys in nn.ParameterDict order, i.e. SORTED AS STRINGS
    ('0','1','10','100',...) -- the order HuffmanCoding's heap is filled in
    (indices_coding.py:46-49).  Backed by one [n_e] tensor instead of n_e Parameters."""

    def __init__(self, owner):
        self._owner = owner
        self._keys = sorted(str(i) for i in range(owner.n_e))

    def __getitem__(self, k):
        i = int(k)
        if not 0 <= i < self._owner.n_e or str(i) != str(k):
            raise KeyError(k)
        return self._owner.usage_counter[i:i + 1]

    def __setitem__(self, k, v):   # `counter[k] += 1` re-assigns the (already updated) view
        view = self[k]
        if v is not view:
            view.copy_(torch.as_tensor(v).reshape(1))

    def __delitem__(self, k):
        raise TypeError("embedding_counter entries cannot be deleted")

    def __iter__(self):
        return iter(self._keys)

    def __len__(self):
        return len(self._keys)

    def as_int_list(self):
        """[int(counter[str(i)].item()) for i in range(n_e)] with ONE device->host copy"""
        return [int(v) for v in self._owner.usage_counter.detach().cpu().tolist()]


class _VQFunction(torch.autograd.Function):
    """Forward = the HIP kernel.  Backward = csrc/cgic_vq_bwd.hip, restating quantize.py:85-93 analytically:
    z_q = z + (e - z).detach()  => dz += g_zq;  loss = m1 + beta*m2 (legacy) with
    m1 = mean((e.detach() - z)^2), m2 = mean((e - z.detach())^2)  => dz += g_loss * (-2/n * w_z) * (e - z) and a
    deterministic scatter-add of g_loss * (2/n * w_e) * (e - z) into the codebook rows."""

    @staticmethod
    def forward(ctx, z, weight, beta, legacy, hist):
        z_q, loss, idx = _vq_forward(z, weight, beta, legacy, hist)
        ctx.save_for_backward(z, weight, idx)
        ctx.beta, ctx.legacy = beta, legacy
        ctx.mark_non_differentiable(idx)
        return z_q, loss, idx

    @staticmethod
    def backward(ctx, g_zq, g_loss, _):
        z, weight, idx = ctx.saved_tensors
        gz, gw = vq_backward(z, weight, idx, g_zq, g_loss, ctx.beta, ctx.legacy,
                             want_gz=ctx.needs_input_grad[0], want_gw=ctx.needs_input_grad[1])
        return gz, gw, None, None, None


def vq_backward(z, weight, idx, g_zq, g_loss, beta, legacy, want_gz=True, want_gw=True):
    """(g_z [B,C,h,w] or None, g_codebook [K,C] or None) -- cgic_vq_backward_f32"""
    _lib.require_device(z, weight, idx, g_zq, g_loss)
    B, C, h, w = z.shape
    z = z.contiguous()
    wt = weight.detach().contiguous()
    dev = z.device
    g_zq = None if g_zq is None else g_zq.contiguous().float()
    g_loss = None if g_loss is None else g_loss.reshape(1).contiguous().float()
    gz = torch.empty_like(z) if want_gz else None
    gw = torch.empty_like(wt) if want_gw else None
    N = B * h * w
    ws = torch.empty(_lib.lib().cgic_vq_backward_workspace_bytes(N, wt.shape[0]), dtype=torch.uint8, device=dev) if want_gw else None
    with _lib.on_device(dev):
        _lib.call("cgic_vq_backward_f32", _lib.ptr(z), B, h * w, _lib.ptr(wt), wt.shape[0], C, _lib.ptr(idx.contiguous()),
                  _lib.ptr(g_zq), _lib.ptr(g_loss), float(beta), int(bool(legacy)), _lib.ptr(gz), _lib.ptr(gw), _lib.ptr(ws),
                  _lib.current_stream(dev))
    return gz, gw


def prepare_codebook(weight, out=None):
    """The codebook image cgic_vq_prepare_f32 makes for the filter path (uint8 tensor; None if this K has no filter path):
    row norms, the maxima that fix the fp16 scaling and the split MFMA operands -- what every workgroup of every launch
    otherwise derives from `weight` (torch.sum(embedding.weight**2) of quantize.py:73-75 is per call in the reference too).
    A SNAPSHOT: pass it only to launches against the same, unchanged weights.  `out`: an image made earlier for a codebook of
    the same K -- it is rewritten IN PLACE (same device pointer), so hipGraphs captured with it see the new weights."""
    _lib.require_device(weight)
    wt = weight.detach().contiguous()
    if wt.dtype != torch.float32 or wt.dim() != 2:
        raise TypeError("prepare_codebook: fp32 [K, e_dim] weight expected")
    nbytes = int(_lib.lib().cgic_vq_prepared_bytes(wt.shape[0]))
    if nbytes == 0:
        return None
    if out is not None and (out.numel() != nbytes or out.device != wt.device or out.dtype != torch.uint8):
        raise ValueError("prepare_codebook: `out` is not an image of a codebook of this size on this device")
    img = out if out is not None else torch.empty(nbytes, dtype=torch.uint8, device=wt.device)
    with _lib.on_device(wt.device):
        _lib.call("cgic_vq_prepare_f32", _lib.ptr(wt), wt.shape[0], wt.shape[1], _lib.ptr(img), _lib.current_stream(wt.device))
    return img


def _vq_forward(z, weight, beta, legacy, hist, want_zq=True, want_loss=True, kernel="mfma", quant_conv=None, conv_bias_first=False,
                prepared=None):
    """quant_conv: optional Conv2d(4, 4, 1) (or (weight, bias)) applied to z inside the kernel -- CGIC.quant_conv
    (model.py:51,110); conv_bias_first selects which of the CPU reference's two rounding sequences to reproduce;
    prepared: prepare_codebook(weight) of the same, unchanged weight (MFMA kernels only)"""
    _lib.require_device(z, weight)
    if z.dtype != torch.float32 or weight.dtype != torch.float32:
        raise TypeError("VectorQuantizer computes in fp32 like the reference; got "
                        f"{z.dtype}/{weight.dtype}")
    B, C, h, w = z.shape
    z = z.contiguous()
    weight = weight.detach().contiguous()
    N = B * h * w
    idx = torch.empty(N, dtype=torch.int64, device=z.device)
    z_q = torch.empty_like(z) if want_zq else None
    loss = torch.empty((), dtype=torch.float32, device=z.device) if want_loss else None
    ws = None
    if want_loss:
        ws = torch.empty(_lib.lib().cgic_vq_workspace_bytes(N), dtype=torch.uint8, device=z.device)
    fn = "cgic_vq_forward_f32" if kernel == "mfma" else "cgic_vq_forward_valu_f32"
    qc, keep = _lib.conv_arg(quant_conv, conv_bias_first)
    extra = (_lib.ptr(prepared),) if kernel == "mfma" else ()
    with _lib.on_device(z.device):
        _lib.call(fn, _lib.ptr(z), B, h * w, _lib.ptr(weight), weight.shape[0], C, float(beta), int(bool(legacy)),
                  _lib.ptr(idx), _lib.ptr(z_q), _lib.ptr(loss), _lib.ptr(hist), _lib.ptr(ws), qc, *extra,
                  _lib.current_stream(z.device))
    del keep
    return z_q, loss, idx


def vq_forward_route(z, weight, beta, legacy, e16, e8, coarse_ratio, medium_ratio, per_image=True, want_gate=False,
                     want_zq=True, want_loss=True, quant_conv=None, conv_bias_first=False, prepared=None, pixels=None, flat8=None,
                     refine_queues=None):
    """VectorQuantize2.forward and TripleGrainFixedEntropyRouter.forward in ONE launch (the router's per-image
    workgroups ride behind the VQ workgroups; see cgic_vq_forward_route_f32).  Returns
    (z_q, loss, indices, [mask_c, mask_m, mask_f], gate, mode) -- identical to the two separate calls.
    pixels: the image batch the maps were made from (fp32 [B,3,H,W] or uint8 [B,H,W,3]) -> the router's threshold-band
    refinement (router.TripleGrainFixedEntropyRouter.forward); flat8: its constant-patch map (default: the one entropy_maps
    left on the maps).  refine_queues (images with a router workgroup of their own: per_image, up to 32x32 patches): True -- an image
    whose threshold band is long starts over with the launch's refinement queues and the routers that are done help (tie-heavy
    batches: 71 -> 50 us on smooth 8-bit content; that kernel variant costs the ordinary launch ~2 us alone); None: the process
    default (_lib.REFINE_FUSED_QUEUES, off); pipeline.HotPathPipeline decides per stream of batches.  Same masks either way."""
    import ctypes
    _lib.require_device(z, weight, e16, e8)
    if refine_queues is None:
        refine_queues = _lib.REFINE_FUSED_QUEUES
    if flat8 is None and pixels is not None:
        from .router import _flat_of
        flat8 = _flat_of(pixels, e8, e16)
    B, C, h, w = z.shape
    z = z.contiguous()
    weight = weight.detach().contiguous()
    e16 = e16.contiguous().float()
    e8 = e8.contiguous().float()
    _, h16, w16 = e16.shape
    if tuple(e8.shape) != (B, 2 * h16, 2 * w16) or e16.shape[0] != B:
        raise ValueError("entropy maps do not match the latent batch")
    dev = z.device
    N = B * h * w
    idx = torch.empty(N, dtype=torch.int64, device=dev)
    z_q = torch.empty_like(z) if want_zq else None
    loss = torch.empty((), dtype=torch.float32, device=dev) if want_loss else None
    ws = torch.empty(_lib.lib().cgic_vq_workspace_bytes(N), dtype=torch.uint8, device=dev) if want_loss else None
    mc = torch.empty((B, 1, h16, w16), dtype=torch.int32, device=dev)
    mm = torch.empty((B, 1, 2 * h16, 2 * w16), dtype=torch.int32, device=dev)
    mf = torch.empty((B, 1, 4 * h16, 4 * w16), dtype=torch.int32, device=dev)
    gate = torch.empty((B, 1, 4 * h16, 12 * w16), dtype=torch.float32, device=dev) if want_gate else None
    mode = ctypes.c_int(0)
    qc, keep = _lib.conv_arg(quant_conv, conv_bias_first)
    # (the scratch of the launch's refinement queues: images whose threshold band is long publish it and the router workgroups
    # that are done evaluate patches for them; the row bands of a large tile split a band between them)
    px, keep_px = _lib.pixels_arg(pixels, B, h16, w16, per_image, flat8=flat8, explicit=True,
                                   queues=bool(per_image) and (bool(refine_queues) or h16 * w16 >= _lib.REFINE_SPLIT_MIN_PATCHES))
    with _lib.on_device(dev):
        _lib.call("cgic_vq_forward_route_f32", _lib.ptr(z), B, h * w, _lib.ptr(weight), weight.shape[0], C, float(beta),
                  int(bool(legacy)), _lib.ptr(idx), _lib.ptr(z_q), _lib.ptr(loss), _lib.ptr(ws), _lib.ptr(e16), _lib.ptr(e8),
                  h16, w16, float(coarse_ratio), float(medium_ratio), int(bool(per_image)), _lib.ptr(mc), _lib.ptr(mm),
                  _lib.ptr(mf), _lib.ptr(gate), ctypes.byref(mode), qc, _lib.ptr(prepared), px, _lib.current_stream(dev))
    del keep, keep_px
    return z_q, loss, idx, [mc, mm, mf], gate, mode.value


class PendingQuantConv(torch.Tensor):
    """What FusedQuantConv returns under no_grad: the convolution's INPUT, tagged with the convolution that is still due.

    The hand-off to the quantiser is explicit: VectorQuantize2.forward recognises the type and applies W h + b inside its
    kernel (on the four channel values each lane holds anyway: one HBM round trip of the latent less).  Anybody else who
    touches the tensor -- arithmetic, another module, .cpu(), printing -- gets the convolved latent: every torch function
    first materialises the pending convolution (a plain F.conv2d) and runs on its result, so `model.quant_conv(h)` used on
    its own is still the convolution, and a latent that was convolved some other way is an ordinary tensor that the quantiser
    takes as it is.  Only shape / dtype / device queries are answered without materialising (a 1x1 convolution keeps them)."""

    _METADATA = None

    @staticmethod
    def __new__(cls, h, conv):
        t = torch.Tensor._make_subclass(cls, h.detach(), False)
        t._cgic_conv = conv
        return t

    def plain(self):
        """the untouched convolution input as an ordinary tensor (same storage)"""
        with torch._C.DisableTorchFunctionSubclass():
            return self.as_subclass(torch.Tensor)

    def materialize(self):
        conv = self._cgic_conv
        return torch.nn.functional.conv2d(self.plain(), conv.weight, conv.bias)

    @classmethod
    def __torch_function__(cls, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if cls._METADATA is None:
            T = torch.Tensor
            cls._METADATA = {T.shape.__get__, T.dtype.__get__, T.device.__get__, T.is_cuda.__get__, T.ndim.__get__, T.layout.__get__,
                             T.requires_grad.__get__, T.dim, T.size, T.numel, T.is_contiguous, T.is_floating_point, T.get_device,
                             T.stride, T.storage_offset, T.element_size}
        if func in cls._METADATA:
            with torch._C.DisableTorchFunctionSubclass():
                return func(*args, **kwargs)
        from torch.utils._pytree import tree_map
        done = lambda a: a.materialize() if isinstance(a, PendingQuantConv) else a
        return func(*tree_map(done, args), **tree_map(done, kwargs))


class FusedQuantConv(nn.Conv2d):
    """CGIC.quant_conv (model.py:51) with its work moved into the quantiser's kernel: under no_grad, on the GPU, it returns a
    PendingQuantConv -- its input, tagged -- and VectorQuantize2 applies the convolution in its kernel; any other consumer
    of that tensor gets the convolved latent (see PendingQuantConv).  With autograd on it is a plain Conv2d.  Same
    parameters, same state_dict keys."""

    bias_first = False      # which of the CPU reference's two rounding sequences the fused kernel reproduces (cgic_hip.h)

    @classmethod
    def adopt(cls, conv):
        if tuple(conv.weight.shape) != (4, 4, 1, 1):
            raise NotImplementedError("FusedQuantConv: Control-GIC's quant_conv is Conv2d(4, 4, 1)")
        m = cls(4, 4, 1, bias=conv.bias is not None)
        m.weight, m.bias = conv.weight, conv.bias
        m.train(conv.training)
        return m

    def defers(self, x):
        return (not torch.is_grad_enabled()) and self.weight.is_cuda and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 \
            and not isinstance(x, PendingQuantConv)

    def forward(self, x):
        return PendingQuantConv(x, self) if self.defers(x) else super().forward(x)


class VectorQuantize2(nn.Module):
    def __init__(self, n_e, e_dim, beta, remap=None, unknown_index="random", sane_index_shape=False,
                 legacy=True):
        super().__init__()
        self.n_e = n_e
        self.e_dim = e_dim
        self.beta = beta
        self.legacy = legacy
        self.embedding = nn.Embedding(n_e, e_dim)
        self.embedding.weight.data.uniform_(-1.0 / n_e, 1.0 / n_e)          # quantize.py:26
        # usage counter (quantize.py:28): n_e fp32 counters, saved as embedding_counter.<i>
        self.register_buffer("usage_counter", torch.zeros(n_e), persistent=False)
        # exact integer histogram accumulated by the kernel in training mode
        self.register_buffer("usage_hist", torch.zeros(n_e, dtype=torch.int64), persistent=False)
        if remap is not None:
            raise NotImplementedError("remap is unused by Control-GIC (remap=None, model.py:49-50) and not built here")
        self.remap = None
        self.re_embed = n_e
        self.sane_index_shape = sane_index_shape
        self._register_state_dict_hook(self._save_counter)
        self._register_load_state_dict_pre_hook(self._load_counter)

    # ---- state_dict compatibility: quantize.embedding_counter.<i>, each of shape [1]
    @staticmethod
    def _save_counter(module, state, prefix, local_metadata):
        c = module.usage_counter.detach()
        for k in module.embedding_counter:
            state[f"{prefix}embedding_counter.{k}"] = c[int(k):int(k) + 1].clone()

    def _load_counter(self, state, prefix, local_metadata, strict, missing, unexpected, errors):
        self._prepared = None                       # new weights are coming in: the snapshot is stale
        loaded = False
        for k in list(state.keys()):
            if k.startswith(prefix + "embedding_counter."):
                i = int(k[len(prefix) + len("embedding_counter."):])
                if 0 <= i < self.n_e:
                    with torch.no_grad():
                        self.usage_counter[i] = state[k].reshape(-1)[0].to(self.usage_counter)
                    loaded = True
                del state[k]
        if loaded:
            # counters that come out of a checkpoint are the whole job's (every rank loads the same file): they are the base
            # that sync_usage_counter_now() must not add up again over the ranks
            self._usage_synced = self.usage_counter.detach().to(torch.float64).round().to(torch.int64)

    @property
    def embedding_counter(self):
        return _CounterView(self)

    #: False (default) = the reference: every rank counts its own shard (its counters are requires_grad=False Parameters that DDP
    #: never reduces, quantize.py:28,79-81) and no collective sits inside forward() -- an asymmetric training-mode forward
    #: (a rank-0-only calibration pass, uneven step counts) cannot hang the job.  True: all-reduce the step's histogram over
    #: the process group before it is folded in, so that every rank's counter -- the Huffman frequency table that ends up in
    #: the checkpoint -- counts the whole batch (one blocking 8 KB collective per training-mode forward on EVERY rank).  The
    #: cheaper way to the same table: leave this off and call `sync_usage_counter_now()` once at checkpoint / epoch end.
    sync_usage_counter = False

    def fold_usage_hist(self):
        """Add the kernel's exact int64 histogram into the fp32 counters and clear it; with a process group, all-reduce it
        first (one 8 KB int64 all-reduce: RCCL on the GPUs, exact).
        The reference adds 1.0 per vector in fp32 (quantize.py:79-81), which stops
        counting at 2**24; below that both give the same integers."""
        with torch.no_grad():
            if self.sync_usage_counter:
                from . import dist as cdist
                cdist.all_reduce_histogram(self.usage_hist)          # no-op without a process group / with one rank
            self.usage_counter += self.usage_hist.to(self.usage_counter.dtype)
            self.usage_hist.zero_()

    def prepare(self):
        """Inference: snapshot the codebook image of the HIP kernel once (prepare_codebook) instead of deriving it at the head
        of every launch.  Used by forward() / indices() while the module is in eval mode and autograd is off; dropped by
        train() and load_state_dict().  The snapshot is keyed on the weight tensor and its autograd version counter: an
        in-place update of embedding.weight that autograd sees (optimizer step, `with no_grad(): w.copy_()`, EMA) is noticed by
        the next forward(), which rewrites the image in place.  Writes through `weight.data` bypass the version counter: call
        prepare() again after those."""
        w = self.embedding.weight
        old = getattr(self, "_prepared", None)
        if old is not None and (old.device != w.device or self._prepared_key[0] is not w):
            old = None
        self._prepared = prepare_codebook(w, out=old)
        self._prepared_key = (w, w._version)
        return self

    def train(self, mode=True):
        if mode:
            self._prepared = None
        return super().train(mode)

    def _prepared_image(self):
        if self.training or torch.is_grad_enabled():
            return None
        img = getattr(self, "_prepared", None)
        if img is not None:
            w = self.embedding.weight
            if self._prepared_key[0] is not w or self._prepared_key[1] != w._version:
                self.prepare()                      # the weights changed under the snapshot
                img = self._prepared
        return img

    def sync_usage_counter_now(self):
        """Sum over the process group what every rank has counted SINCE THE LAST SYNC (or since the counters were loaded from a
        checkpoint: those are the whole job's already) and add it to that common base: every rank then holds the counts of the
        whole data set, like sync_usage_counter=True would have accumulated step by step (exact while the totals stay below
        2^24, the fp32 counter's own limit).  Safe to call at every checkpoint / epoch end: only the per-rank delta is reduced,
        so earlier counts are never multiplied by the world size.  Collective: call it on every rank.  Counters filled by hand
        identically on every rank are a common base too: say so with mark_usage_counter_synced()."""
        from . import dist as cdist
        with torch.no_grad():
            total = self.usage_counter.to(torch.float64).round().to(torch.int64)
            base = getattr(self, "_usage_synced", None)
            if base is None or base.shape != total.shape:
                base = torch.zeros_like(total)
            elif base.device != total.device:
                base = base.to(total.device)        # load_state_dict on CPU, then .cuda(): the buffer moved, the base follows
            delta = total - base
            cdist.all_reduce_histogram(delta)
            total = base + delta
            self.usage_counter.copy_(total.to(self.usage_counter.dtype))
            self._usage_synced = total

    def mark_usage_counter_synced(self):
        """declare the current counters common to all ranks (e.g. after filling them by hand on every rank)"""
        self._usage_synced = self.usage_counter.detach().to(torch.float64).round().to(torch.int64)

    def forward(self, z):
        if not isinstance(z, PendingQuantConv) and z.dtype != torch.float32:
            z = z.float()                   # (autocast regions hand over fp16 / bf16: the reference quantises in fp32)
        hist = self.usage_hist if self.training else None
        conv = None
        if isinstance(z, PendingQuantConv):                   # FusedQuantConv handed its INPUT over: the convolution is due here
            if torch.is_grad_enabled() or self.n_e % 64 or self.n_e > 1024:
                z = z.materialize()                           # (autograd switched on in between / no fused kernel for this K)
            else:
                conv, z = z._cgic_conv, z.plain()
        if torch.is_grad_enabled() and (z.requires_grad or self.embedding.weight.requires_grad):
            z_q, loss, idx = _VQFunction.apply(z, self.embedding.weight, self.beta, self.legacy, hist)
        else:
            z_q, loss, idx = _vq_forward(z, self.embedding.weight, self.beta, self.legacy, hist, quant_conv=conv,
                                         conv_bias_first=conv.bias_first if conv is not None else False,
                                         prepared=self._prepared_image())
        if self.training:
            self.fold_usage_hist()
        return z_q, loss, idx

    def indices(self, z, kernel="mfma"):
        """argmin only (no z_q / loss): what compress() consumes (model.py:216)."""
        return _vq_forward(z, self.embedding.weight, self.beta, self.legacy, None, False, False, kernel,
                           prepared=self._prepared_image() if kernel == "mfma" else None)[2]


VectorQuantizer = VectorQuantize2
