"""The hot path over batches: HotPathPipeline (one batch, five dependent launches), LaneStream (successive batches on
independent hardware queues, hipGraphs), GraphLanes (arbitrary captured work on independent queues).
"""
import torch

from . import _lib
from .codec import GrainCodec, decoder_mode
from .entropy import entropy_maps
from .quantize import _vq_forward, vq_forward_route, prepare_codebook
from .router import TripleGrainFixedEntropyRouter


class HotPathPipeline:
    """One batch through the hot path on the current stream: entropy maps -> [VQ + per-image router, one launch] -> stream
    coder (+ usage histogram) -> prefix decoder -> scatter / merge / gather: five launches, each depending on the one before.
    (Cutting a batch into chunks on several streams and forking VQ next to entropy -> router were measured slower, NOTES.md;
    what overlaps well is WHOLE batches on independent queues: LaneStream.)"""

    def __init__(self, quantizer, coarse_ratio, medium_ratio, frequency=None, fuse_router=True, prepare=False, refine=True,
                 refine_queues="auto"):
        self.vq = quantizer
        # refine_queues: True / False / "auto".  The fused launch has a variant in which an image whose threshold band is long starts
        # over with the launch's refinement queues and the router workgroups that are done help (quantize.vq_forward_route): worth
        # ~17 % on batches of smooth 8-bit content, ~1.5 us of latency on the ordinary launch.  "auto": decide() looks at a batch
        # of the stream once (LaneStream.capture does; a bare pipeline stays on the plain kernel until decide() is called) -- tie-heavy
        # content comes by the stream.  The masks are the same either way.
        self.refine_queues = refine_queues
        self._queues = refine_queues is True
        # prepare=True: inference against a codebook that does not change -- the VQ kernel's codebook image is made once
        # (quantize.prepare_codebook, a snapshot of the weights NOW; refresh_codebook() after changing them) instead of by
        # every workgroup of every launch
        self.prepared = prepare_codebook(quantizer.embedding.weight) if prepare else None
        self.router = TripleGrainFixedEntropyRouter(coarse_ratio, medium_ratio, per_image=True)
        self.codec = GrainCodec(frequency if frequency is not None else quantizer.embedding_counter,
                                quantizer.embedding.weight)
        # True: the per-image router workgroups ride in the VQ launch (best for ONE batch at a time).  False: a launch
        # of its own in front of the VQ kernel.
        self.fuse_router = bool(fuse_router)
        # refine: the router re-evaluates, in the reference's own arithmetic, the patches whose entropy lies within the
        # default kernel's error of a threshold (cgic_router_refined_f32): masks -- hence bytes -- equal to the CPU
        # reference's from PIXELS, also on tie-heavy content
        self.refine = bool(refine)

    def refresh_codebook(self):
        """after changing embedding.weight: rewrite the codebook image IN PLACE (same device buffer), so that hipGraphs
        captured with it (LaneStream, GraphLanes) read the new weights on their next replay instead of a freed snapshot"""
        if self.prepared is not None:
            self.prepared = prepare_codebook(self.vq.embedding.weight, out=self.prepared)

    def decide(self, x):
        """refine_queues="auto": look at one batch of the stream -- how many patches per image lie inside the threshold bands AND are
        not constant (those are evaluated from their pixels) -- and choose the fused launch's variant for the batches to come.
        Host-side, once per stream (a few torch kernels and one synchronisation: never inside a timed or captured region)."""
        if self.refine_queues != "auto" or not self.refine or not self.fuse_router:
            return self._queues
        e8, e16 = entropy_maps(x)
        flat8 = getattr(e8, "_cgic_flat8", None)
        c, m = float(self.router.coarse_grain_ratio), float(self.router.medium_grain_ratio)
        units, mean_units = 0, 0.0
        # (the COARSE band decides: 16x16 patches are four units each, and a long coarse band -- the nearly constant patches of smooth
        # content -- is where the queues pay, 71 -> 50 us per launch; a long MEDIUM band of 8x8 edge patches (flat regions with edges)
        # is done sooner where it was found: 57 us against 62 with the restart and the queue's hand-offs)
        for e, ratio, per_patch in ((e16, c, 4),):
            B = e.shape[0]
            v = e.reshape(B, -1)
            k = max(int(round(v.shape[1] * ratio)), 1)
            if ratio <= 0.0 or k > v.shape[1]:
                continue
            thr = torch.kthvalue(v, k, dim=1).values[:, None]
            band = (v - thr).abs() <= 4e-6
            band_all = band
            if flat8 is not None:                       # constant patches cost one evaluation per gray, not one per patch
                f = flat8 if per_patch == 1 else torch.nn.functional.max_pool2d(torch.isnan(flat8).float()[:, None], 2)[:, 0]
                nonconst = torch.isnan(f) if per_patch == 1 else f > 0
                band = band & nonconst.reshape(B, -1)
            members = band_all.sum(dim=1)                  # (the kernel takes the queues only for bands of at most 64 members)
            per_image = band.sum(dim=1).float() * per_patch * (members <= 64).float()
            units, mean_units = max(units, int(per_image.max().item())), float(per_image.mean().item())
        # more than eight rounds of a router workgroup's eight waves in SOME image while the batch as a whole has routers to spare:
        # when every image carries about the same band, every router is busy with its own and the restart is pure cost (flat regions
        # with edges: 57 us plain, 62 with the queues)
        self._decide_stats = (units, mean_units)
        self._queues = units > 64 and units >= 1.6 * mean_units
        return self._queues

    def _chain(self, x, z, hist, decode, decoder=None):
        e8, e16 = entropy_maps(x)
        px = x if self.refine else None
        if not self.fuse_router:
            mask, _, _, mode = self.router(e16, e8, want_gate=False, pixels=px)
            zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None, prepared=self.prepared)
        else:
            # VQ and the per-image router share one launch (the router rides in the VQ kernel's shadow)
            zq, loss, ind, mask, _, mode = vq_forward_route(
                z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, e16, e8,
                self.router.coarse_grain_ratio, self.router.medium_grain_ratio, per_image=True, prepared=self.prepared,
                pixels=px, refine_queues=self._queues)
        comp = self.codec.compress(ind, mask, mode, hist=hist)      # usage histogram rides on the coder launch
        dec = self.codec.decompress(comp, decoder=decoder) if decode else None
        return {"e8": e8, "e16": e16, "mask": mask, "mode": mode, "z_q": zq, "loss": loss, "ind": ind,
                "comp": comp, "dec": dec}

    def run(self, x, z, hist=None, decode=True):
        """x [B,3,H,W], z [B,4,H/4,W/4] on the device -> [result dict] (a list for compatibility with round 1's chunked form).
        `hist` (int64 [n_e], optional) accumulates the usage histogram."""
        return [self._chain(x, z, hist, decode)]


class HotCall:
    """CGIC.compress (hot path: model.py:206-401 without the conv nets) for batches of ONE size as ONE foreign call per batch
    (cgic_compress_image: entropy maps -> [VQ + per-image router] -> stream coder (+ usage histogram) -> prefix decoder + merge)
    over buffers allocated ONCE -- the eager counterpart of a captured hipGraph: what a loop like inference.py:157-166 pays per
    image is one ctypes call and the launches themselves, no Python per kernel and no allocation.
    The results of a call (`out`: the same keys as HotPathPipeline.run) live in this object's buffers: valid until the next call."""

    def __init__(self, quantizer, coarse_ratio, medium_ratio, B, H, W, frequency=None, decode=True, u8=False, hist=None,
                 decoder=None, prepare=True, want_zq=True, want_loss=True):
        import ctypes
        from .codec import CompressedBatch, _decoder_flag
        if H % 16 or W % 16:
            raise ValueError("H and W must be multiples of 16")
        w = quantizer.embedding.weight
        dev = w.device
        _lib.require_device(w)
        self.vq, self.dev, self.shape, self.u8 = quantizer, dev, (B, H, W), bool(u8)
        self.ratios = (float(coarse_ratio), float(medium_ratio))
        self.codec = GrainCodec(frequency if frequency is not None else quantizer.embedding_counter, w)
        self.prepared = prepare_codebook(w) if prepare else None
        self.decoder = _decoder_flag(decoder)
        h, ww = H // 4, W // 4
        l = _lib.lib()
        f32, i32, i64, u8t = torch.float32, torch.int32, torch.int64, torch.uint8
        E = lambda shape, dt: torch.empty(shape, dtype=dt, device=dev)
        slot = self.codec.slot_bytes(h, ww)
        t = {"e8": E((B, H // 8, W // 8), f32), "e16": E((B, H // 16, W // 16), f32), "flat8": E((B, H // 8, W // 8), f32),
             "x_out": E((B, 3, H, W), f32) if u8 else None,
             "ind": E((B * h * ww,), i64), "z_q": E((B, 4, h, ww), f32) if want_zq else None, "loss": E((), f32) if want_loss else None,
             "mask": [E((B, 1, h // 4, ww // 4), i32), E((B, 1, h // 2, ww // 2), i32), E((B, 1, h, ww), i32)],
             "data": E((B, _lib.NUM_STREAMS, slot), u8t), "nbytes": E((B, _lib.NUM_STREAMS), i32), "hist": hist}
        if decode:
            t.update({"dind": E((B, h, ww), i64), "dmask": [E((B, 1, h // 4, ww // 4), i32), E((B, 1, h // 2, ww // 2), i32), E((B, 1, h, ww), i32)],
                      "dz_q": E((B, 4, h, ww), f32), "status": E((B,), i32)})
        ws = {"vq": E((l.cgic_vq_workspace_bytes(B * h * ww),), u8t) if want_loss else None,
              "c": E((max(1, l.cgic_compress_workspace_bytes(B, h, ww)),), u8t),
              "d": E((l.cgic_decompress_workspace_bytes(B, h, ww),), u8t) if decode else None}
        self._t, self._ws, self._decode, self._slot = t, ws, bool(decode), slot
        p = _lib.ptr
        io = _lib.ImageIO()
        io.x_is_u8 = int(self.u8)
        io.x_out, io.e8, io.e16, io.flat8 = p(t["x_out"]), p(t["e8"]), p(t["e16"]), p(t["flat8"])
        io.ind, io.z_q, io.loss = p(t["ind"]), p(t["z_q"]), p(t["loss"])
        io.mask_c, io.mask_m, io.mask_f = (p(m) for m in t["mask"])
        io.streams, io.slot, io.nbytes, io.hist = p(t["data"]), slot, p(t["nbytes"]), p(hist)
        if decode:
            io.dind, io.dz_q, io.status = p(t["dind"]), p(t["dz_q"]), p(t["status"])
            io.dmask_c, io.dmask_m, io.dmask_f = (p(m) for m in t["dmask"])
        io.ws_vq, io.ws_compress, io.ws_decompress = p(ws["vq"]), p(ws["c"]), p(ws["d"])
        # the refinement scratch: the launch's queues (images with long threshold bands publish them, the other routers help; large
        # tiles: their row bands split a band) -- or, for an image beyond ~768x768 routed as one segment, the patched copies of its maps
        nref = l.cgic_router_refine_scratch_bytes(B, H // 16, W // 16, 1)
        if nref and (_lib.REFINE_FUSED_QUEUES or not l.cgic_router_refine_in_lds(B, H // 16, W // 16, 1)):
            ws["r"] = E((nref,), u8t)
            io.ws_refine, io.ws_refine_bytes = p(ws["r"]), nref
        self._io = io
        self._mode = ctypes.c_int(0)
        self._CompressedBatch = CompressedBatch
        self._fn = l.cgic_compress_image
        self._bins = _lib.linspace_bins()
        self.out = None

    def refresh_codebook(self):
        if self.prepared is not None:
            self.prepared = prepare_codebook(self.vq.embedding.weight, out=self.prepared)

    def __call__(self, x, z):
        """x [B,3,H,W] fp32 (or uint8 frames [B,H,W,3] with u8=True), z [B,4,H/4,W/4] fp32, contiguous, on the device -> dict"""
        import ctypes
        B, H, W = self.shape
        if tuple(x.shape) != ((B, H, W, 3) if self.u8 else (B, 3, H, W)) or x.dtype != (torch.uint8 if self.u8 else torch.float32) \
                or tuple(z.shape) != (B, 4, H // 4, W // 4) or z.dtype != torch.float32 or not x.is_contiguous() or not z.is_contiguous() \
                or x.device != self.dev or z.device != self.dev:
            raise ValueError("HotCall: x / z do not have the shape, dtype, device or layout this object was built for")
        io, t, vq = self._io, self._t, self.vq
        io.x, io.z = x.data_ptr(), z.data_ptr()
        w = vq.embedding.weight

        def go():
            _lib.check(self._fn(self.codec.huffman.table.handle, w.data_ptr(), w.shape[0], w.shape[1], _lib.ptr(self.prepared), B, H, W,
                                self.ratios[0], self.ratios[1], float(vq.beta), int(bool(vq.legacy)), self._bins, 32, 0.01, self.decoder,
                                ctypes.byref(io), ctypes.byref(self._mode), torch.cuda.current_stream(self.dev).cuda_stream))
        if torch.cuda.current_device() == self.dev.index:      # (the torch device context manager costs ~10 us: only when it is needed)
            go()
        else:
            with torch.cuda.device(self.dev):
                go()
        mode = self._mode.value
        comp = self._CompressedBatch(t["data"], t["nbytes"], mode, H // 4, W // 4)
        self.out = {"e8": t["e8"], "e16": t["e16"], "mask": t["mask"], "mode": mode, "z_q": t["z_q"], "loss": t["loss"], "ind": t["ind"],
                    "comp": comp, "dec": (t["dind"], t["dmask"], t["dz_q"], t["status"]) if self._decode else None, "x": t["x_out"]}
        return self.out


class BatchSlot:
    """static buffers + two captured hipGraphs (encode side / decode side) for one batch of the stream"""

    def __init__(self, x, z):
        self.x, self.z = x, z
        self.enc = self.dec = None          # dict of encode outputs / tuple of decode outputs (graph-owned memory)
        self.g_enc = self.g_dec = None
        self.ev_enc = torch.cuda.Event()
        self.ev_dec = torch.cuda.Event()
        self.used = False


def capture_graph(fn, stream):
    """capture `fn()` (which only enqueues work on the current stream) into a hipGraph on `stream` -> (graph, fn's result).
    The ticket slots the captured launches take from the library's pool go back to it after the graph object has been
    garbage-collected -- at the next capture, behind a device synchronisation (_lib.flush_released) -- so a long-lived process
    can capture per image shape for as long as it likes."""
    _lib.flush_released(stream.device)              # slots of graphs that died since the last capture (waits for their last replay)
    g = torch.cuda.CUDAGraph()
    sc = _lib.ticket_scope()
    try:
        with sc:
            with torch.cuda.stream(stream):
                with torch.cuda.graph(g, stream=stream):
                    out = fn()
    except BaseException:
        sc.release()                                  # a failed capture keeps nothing
        raise
    sc.release_with(g)
    return g, out


def distinct_queue_streams(device, n, candidates=16, spin_cycles=400_000):
    """`n` HIP streams that sit on DIFFERENT hardware queues.

    ROCm multiplexes all HIP streams of a process onto a few hardware queues (GPU_MAX_HW_QUEUES, 4 by default; the null
    stream holds one of them), and which queue a stream lands on depends on how many streams the process created before it.
    Two streams on one queue run their kernels strictly one after the other: four lanes on three queues were measured at
    71 GPixel/s against 87 on four (rocprofv3 kernel trace, queue_id column).  The mapping is not exposed by the HIP API, so
    it is measured: a single-thread spin kernel (`torch.cuda._sleep`) on two streams takes one spin if the queues differ and
    two if they are the same.  Greedy choice over `candidates` pool streams; if fewer than `n` distinct queues exist the
    remaining lanes share queues (still correct, only less overlap)."""
    streams = [torch.cuda.Stream(device) for _ in range(max(n, candidates))]
    if n <= 1:
        return streams[:n]
    import time

    def spin(group):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for st in group:
            with torch.cuda.stream(st):
                torch.cuda._sleep(spin_cycles)
        torch.cuda.synchronize(device)
        return time.perf_counter() - t0

    spin(streams[:1])
    one = min(spin(streams[:1]) for _ in range(3))
    chosen = [streams[0]]
    for c in streams[1:]:
        if len(chosen) == n:
            break
        if all(min(spin([c, k]) for _ in range(2)) < 1.5 * one for k in chosen):
            chosen.append(c)
    for c in streams:                                   # not enough distinct queues: share
        if len(chosen) == n:
            break
        if c not in chosen:
            chosen.append(c)
    return chosen


class LaneStream:
    """Successive batches through the hot path on `lanes` INDEPENDENT HIP streams (batch t runs on lane t % lanes).

    Inside one batch every launch depends on the previous one, and three of the five (stream coder, prefix decoder,
    scatter/merge) plus the router are one-workgroup-per-image kernels: 64-256 workgroups that wait on dependent LDS /
    memory chains while three quarters of the chip idle.  Across batches nothing depends on anything, so the lanes
    carry no events and no cross-stream edges at all -- each lane replays its own hipGraphs on its own stream and the
    hardware schedules the workgroups of up to `lanes` batches side by side.  (The two-stream BatchStream above splits
    ONE batch into an encode and a decode graph tied by events; every event is a graph boundary, ~7 us on the device,
    and it was measured no faster than one stream.  Independent lanes: 53 -> 87 GPixel/s at B=64 of 256x256, 4 lanes.)

    Graphs.  A lane's work is a sequence of steps over its slots in rotation.  `submit(n)` cuts each lane's share into
    runs of at most `max_ring` consecutive steps and replays ONE hipGraph per run (successive graph launches on one
    stream are ~7 us apart on the device and cost the host tens of microseconds each; kernels inside a graph are
    back to back).  A graph is keyed by (first slot position, number of steps) and captured the first time it is
    needed -- `prepare(n)` captures what the next `submit(n)` will replay, so that no capture lands inside a timed or
    latency-critical region.  `ring=False` keeps one graph per step.
    Every graph owns the output buffers of the steps it holds.  After a replay, `slot.enc` / `slot.dec` name the
    buffers of the graph that ran the slot LAST (a slot refilled between two submits is therefore always read back
    from the launch that really processed the new input, whichever mix of graphs the submits took).
    Results are bit-identical to the one-stream order: same kernels, same inputs, no shared scratch between slots
    (per-launch tickets are library-owned, a captured launch keeps its own).

    slots: list of (x [B,3,H,W], z [B,4,H/4,W/4]) device tensors; the caller refills a slot's tensors in place after
    `join()` to feed new data.  `hist` (int64 [n_e], optional) accumulates the usage histogram of everything submitted.
    """

    def __init__(self, quantizer, coarse_ratio, medium_ratio, slots, lanes=4, frequency=None, hist=None, decode=True,
                 graph=True, ring=True, fuse_router=True, decoder=None, max_ring=8, prepare=True, native_launch=True,
                 refine=True, refine_queues="auto"):
        if not slots:
            raise ValueError("LaneStream needs at least one slot")
        # prepare: a stream of batches is inference against ONE codebook -- its image is made once, when capture() runs
        # (a captured launch holds a snapshot of the codebook either way: HotPathPipeline.prepare)
        self.pipe = HotPathPipeline(quantizer, coarse_ratio, medium_ratio, frequency=frequency, fuse_router=fuse_router, prepare=prepare,
                                    refine=refine, refine_queues=refine_queues)
        self.hist = hist
        self.decode = bool(decode)
        self.graph = bool(graph)
        self.ring = bool(ring) and self.graph
        self.max_ring = max(1, int(max_ring)) if self.ring else 1
        self.slots = [BatchSlot(x, z) for x, z in slots]
        self.device = self.slots[0].x.device
        nl = max(1, min(int(lanes), len(self.slots)))
        # several batches in flight: the small-footprint decoder; one at a time: the low-latency one (codec.decoder_mode)
        self.decoder = decoder if decoder is not None else ("throughput" if nl > 1 else "latency")
        with torch.cuda.device(self.device):
            streams = distinct_queue_streams(self.device, nl)          # one hardware queue per lane, measured
        self.lanes = [{"slots": self.slots[j::nl], "pos": 0, "stream": streams[j], "graphs": {}} for j in range(nl)]
        # native_launch (default): all graph launches of a submit in ONE C call (cgic_launch_graphs: no interpreter between
        # them: host time of a K=20 submit 115 -> 93 us, the window 777 -> 756 us); False = replay() from Python
        self.native_launch = bool(native_launch) and self.graph and hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec")
        self._t = 0
        self._captured = False

    def _step(self, s):
        """one batch through the hot path on the current stream -> (enc dict, dec tuple)"""
        enc = self.pipe._chain(s.x, s.z, self.hist, self.decode, decoder=self.decoder)
        return enc, enc["dec"]

    def capture(self, warmup=2):
        """run every slot eagerly (uploads tables, sets function attributes, creates the ticket pools) and capture the
        one-step graph of every slot"""
        cur = torch.cuda.current_stream(self.device)
        side = self.lanes[0]["stream"]
        side.wait_stream(cur)
        with torch.cuda.stream(side):
            self.pipe.refresh_codebook()                # the weights as they are now
            self.pipe.decide(self.slots[0].x)           # refine_queues="auto": the stream's first batch chooses the fused launch's variant
            for s in self.slots:
                for _ in range(warmup):
                    s.enc, s.dec = self._step(s)
        cur.wait_stream(side)
        self._captured = True
        if self.graph:
            for lane in self.lanes:
                for p in range(len(lane["slots"])):
                    self._graph(lane, p, 1)
        for lane in self.lanes:
            lane["stream"].wait_stream(cur)

    def _graph(self, lane, start, count):
        """the hipGraph of `count` consecutive steps of `lane` from slot position `start` (captured on first use):
        (graph, {slot position: (enc, dec) of the LAST step of that slot inside the graph})"""
        key = (start, count)
        hit = lane["graphs"].get(key)
        if hit is not None:
            return hit
        m = len(lane["slots"])
        cur = torch.cuda.current_stream(self.device)
        st = lane["stream"]
        st.wait_stream(cur)
        outs = {}

        def body():
            for i in range(count):
                p = (start + i) % m
                outs[p] = None                        # an earlier step's buffers of this slot go back to the graph's pool
                outs[p] = self._step(lane["slots"][p])
        g, _ = capture_graph(body, st)
        cur.wait_stream(st)
        lane["graphs"][key] = (g, outs)
        return g, outs

    def _plan(self, n, after=0):
        """[(lane, start, count), ...] per lane for the n batches that follow the next `after` ones, without advancing"""
        L = len(self.lanes)
        skip, todo = [0] * L, [0] * L
        for t in range(self._t, self._t + after):
            skip[t % L] += 1
        for t in range(self._t + after, self._t + after + n):
            todo[t % L] += 1
        plans = []
        for j, lane in enumerate(self.lanes):
            m, pos, runs = len(lane["slots"]), (lane["pos"] + skip[j]) % len(lane["slots"]), []
            c = todo[j]
            while c:
                k = min(c, self.max_ring)
                runs.append((pos, k))
                pos = (pos + k) % m
                c -= k
            plans.append(runs)
        return plans

    def prepare(self, n, after=0):
        """capture (outside any timed region) every graph submit(n) will replay once the next `after` batches have been
        submitted: prepare(W); prepare(K, after=W); submit(W); submit(K) leaves no capture between the warm-up and the run"""
        if not self._captured:
            self.capture()
        if self.graph:
            plans = self._plan(n, after)
            for lane, runs in zip(self.lanes, plans):
                for start, count in runs:
                    self._graph(lane, start, count)
            if self.native_launch:
                self._native_args_for(plans)

    def _native_args_for(self, plans):
        import ctypes
        key = tuple(tuple(r) for r in plans)
        cached = getattr(self, "_native_args", {}).get(key)
        if cached is None:
            execs, streams = [], []
            depth = max(len(r) for r in plans)
            for i in range(depth):                                       # one launch per lane and turn, like the Python loop
                for lane, runs in zip(self.lanes, plans):
                    if i < len(runs):
                        execs.append(self._graph(lane, *runs[i])[0].raw_cuda_graph_exec())
                        streams.append(lane["stream"].cuda_stream)
            n = len(execs)
            cached = ((ctypes.c_void_p * n)(*execs), (ctypes.c_void_p * n)(*streams), n)
            self.__dict__.setdefault("_native_args", {})[key] = cached
        return cached

    def _launch_native(self, plans):
        e, s, n = self._native_args_for(plans)
        with torch.cuda.device(self.device):
            _lib.call("cgic_launch_graphs", e, s, n)

    def submit(self, n=1):
        """enqueue the next n batches (batch t on lane t % lanes, slots of a lane in rotation); returns immediately"""
        if not self._captured:
            self.capture()
        plans = self._plan(n)
        self._t += n
        if self.graph:
            for lane, runs in zip(self.lanes, plans):                  # captures (if any are missing) before the first launch
                for start, count in runs:
                    self._graph(lane, start, count)
            if self.native_launch:
                self._launch_native(plans)
            else:
                depth = max(len(r) for r in plans)
                for i in range(depth):                                  # one launch per lane and turn keeps every queue fed
                    for lane, runs in zip(self.lanes, plans):
                        if i < len(runs):
                            with torch.cuda.stream(lane["stream"]):
                                self._graph(lane, *runs[i])[0].replay()
            for lane, runs in zip(self.lanes, plans):
                m = len(lane["slots"])
                for start, count in runs:
                    for p, (enc, dec) in self._graph(lane, start, count)[1].items():
                        lane["slots"][p].enc, lane["slots"][p].dec = enc, dec
                    lane["pos"] = (start + count) % m
        else:
            depth = max(sum(c for _, c in r) for r in plans)
            for i in range(depth):
                for lane, runs in zip(self.lanes, plans):
                    if i < sum(c for _, c in runs):
                        s = lane["slots"][lane["pos"]]
                        lane["pos"] = (lane["pos"] + 1) % len(lane["slots"])
                        with torch.cuda.stream(lane["stream"]):
                            s.enc, s.dec = self._step(s)

    def join(self, stream=None):
        """make `stream` (default: the current one) wait for everything submitted so far"""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        for lane in self.lanes:
            stream.wait_stream(lane["stream"])

    def fork(self, stream=None):
        """make every lane wait for `stream` (default: the current one), e.g. after refilling slots"""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        for lane in self.lanes:
            lane["stream"].wait_stream(stream)


class GraphLanes:
    """Arbitrary captured work on independent hardware queues: `fns` are callables that only enqueue work on the current
    stream (e.g. `highres.compress_tiled` + `decompress_tiled(check=False)` of one image); each is run once eagerly, captured into
    a hipGraph on its own stream (one per hardware queue, `distinct_queue_streams`) and `replay(n)` launches every graph n
    times, lane after lane, with no dependency between the lanes.  `results[k]` is what fns[k] returned during capture (its
    output tensors live in the graph's memory and are refreshed by every replay).  LaneStream is this plus slot rotation and
    ring graphs for the fixed five-launch step."""

    def __init__(self, device, fns, decoder="throughput"):
        self.device = device
        self.decoder = decoder
        with torch.cuda.device(device):
            self.streams = distinct_queue_streams(device, len(fns))
        self.graphs, self.results = [], []
        cur = torch.cuda.current_stream(device)
        with decoder_mode(decoder):
            for fn in fns:
                fn()                                        # eager once: tables, function attributes, ticket pools
            torch.cuda.synchronize(device)
            for fn, st in zip(fns, self.streams):
                st.wait_stream(cur)
                g, out = capture_graph(fn, st)
                self.results.append(out)
                self.graphs.append(g)
        torch.cuda.synchronize(device)

    def replay(self, n=1):
        for _ in range(n):
            for g, st in zip(self.graphs, self.streams):
                with torch.cuda.stream(st):
                    g.replay()

    def join(self, stream=None):
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        for st in self.streams:
            stream.wait_stream(st)
