"""HotPathPipeline -- the whole hot path over a batch, scheduled for throughput.

VQ and the entropy maps are throughput-bound kernels that fill all 256 CUs; the router, the
stream coder and the decoder are one-workgroup-per-image kernels that are latency-bound and
occupy a quarter of the chip.  Images are independent (per-image routing), so the batch is cut
into `chunks` and each chunk runs the full chain on its own HIP stream: the latency-bound
kernels of one chunk overlap the throughput-bound kernels of another.  Results are identical to
the unchunked path (same kernels, same per-image semantics); the fork/join uses stream events
only, so a step can be captured in a hipGraph and replayed.
"""
import torch

from . import _lib
from .codec import GrainCodec, decoder_mode
from .entropy import entropy_maps
from .quantize import _vq_forward, vq_forward_route, prepare_codebook
from .router import TripleGrainFixedEntropyRouter


class HotPathPipeline:
    def __init__(self, quantizer, coarse_ratio, medium_ratio, chunks=1, frequency=None, fork_vq=False, fuse_router=True,
                 prepare=False):
        self.vq = quantizer
        # prepare=True: inference against a codebook that does not change -- the VQ kernel's codebook image is made once
        # (quantize.prepare_codebook, a snapshot of the weights NOW; refresh_codebook() after changing them) instead of by
        # every workgroup of every launch
        self.prepared = prepare_codebook(quantizer.embedding.weight) if prepare else None
        self.router = TripleGrainFixedEntropyRouter(coarse_ratio, medium_ratio, per_image=True)
        self.codec = GrainCodec(frequency if frequency is not None else quantizer.embedding_counter,
                                quantizer.embedding.weight)
        self.chunks = max(1, int(chunks))
        self.fork_vq = int(fork_vq)       # 1: VQ(+hist) on a side stream next to entropy -> router; 2: router on a side stream next to VQ
        # True: the per-image router workgroups ride in the VQ launch (best for ONE batch at a time).  False: a launch
        # of its own in front of the VQ kernel -- a small-footprint kernel that shares CUs with the kernels of OTHER
        # batches when several independent streams of batches are in flight (bench.py --lanes).
        self.fuse_router = bool(fuse_router)
        self._streams = None
        self._side = None

    def refresh_codebook(self):
        if self.prepared is not None:
            self.prepared = prepare_codebook(self.vq.embedding.weight)

    def _get_streams(self, device):
        if self._streams is None or self._streams[0].device != device:
            self._streams = [torch.cuda.Stream(device) for _ in range(self.chunks)]
        return self._streams

    def _chain(self, x, z, hist, decode, decoder=None):
        if self.fork_vq == 2:
            # entropy, then the router on a side stream next to the VQ kernel
            cur = torch.cuda.current_stream(x.device)
            if self._side is None or self._side.device != x.device:
                self._side = torch.cuda.Stream(x.device)
            e8, e16 = entropy_maps(x)
            fork = torch.cuda.Event()
            fork.record(cur)
            self._side.wait_event(fork)
            with torch.cuda.stream(self._side):
                mask, _, _, mode = self.router(e16, e8, want_gate=False)
                join = torch.cuda.Event()
                join.record(self._side)
            zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None, prepared=self.prepared)
            cur.wait_event(join)
        elif self.fork_vq:
            cur = torch.cuda.current_stream(x.device)
            if self._side is None or self._side.device != x.device:
                self._side = torch.cuda.Stream(x.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            self._side.wait_event(fork)
            with torch.cuda.stream(self._side):
                zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None, prepared=self.prepared)
                join = torch.cuda.Event()
                join.record(self._side)
            e8, e16 = entropy_maps(x)
            mask, _, _, mode = self.router(e16, e8, want_gate=False)
            cur.wait_event(join)
        elif not self.fuse_router:
            e8, e16 = entropy_maps(x)
            mask, _, _, mode = self.router(e16, e8, want_gate=False)
            zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None, prepared=self.prepared)
        else:
            e8, e16 = entropy_maps(x)
            # VQ and the per-image router share one launch (the router rides in the VQ kernel's shadow)
            zq, loss, ind, mask, _, mode = vq_forward_route(
                z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, e16, e8,
                self.router.coarse_grain_ratio, self.router.medium_grain_ratio, per_image=True, prepared=self.prepared)
        comp = self.codec.compress(ind, mask, mode, hist=hist)      # usage histogram rides on the coder launch
        dec = self.codec.decompress(comp, decoder=decoder) if decode else None
        return {"e8": e8, "e16": e16, "mask": mask, "mode": mode, "z_q": zq, "loss": loss, "ind": ind,
                "comp": comp, "dec": dec}

    def run(self, x, z, hist=None, decode=True):
        """x [B,3,H,W], z [B,4,H/4,W/4] on the device -> list of per-chunk result dicts (views of the
        batch in order).  `hist` (int64 [n_e], optional) accumulates the usage histogram."""
        B = x.shape[0]
        n = min(self.chunks, B)
        if n == 1:
            return [self._chain(x, z, hist, decode)]
        cur = torch.cuda.current_stream(x.device)
        streams = self._get_streams(x.device)
        bounds = [(B * i) // n for i in range(n + 1)]
        fork = torch.cuda.Event()
        fork.record(cur)
        out = []
        for i in range(n):
            s = streams[i]
            s.wait_event(fork)
            with torch.cuda.stream(s):
                out.append(self._chain(x[bounds[i]:bounds[i + 1]], z[bounds[i]:bounds[i + 1]], hist, decode))
            done = torch.cuda.Event()
            done.record(s)
            cur.wait_event(done)
        return out


class BatchSlot:
    """static buffers + two captured hipGraphs (encode side / decode side) for one batch of the stream"""

    def __init__(self, x, z):
        self.x, self.z = x, z
        self.enc = self.dec = None          # dict of encode outputs / tuple of decode outputs (graph-owned memory)
        self.g_enc = self.g_dec = None
        self.ev_enc = torch.cuda.Event()
        self.ev_dec = torch.cuda.Event()
        self.used = False


class BatchStream:
    """Successive batches through the hot path, software-pipelined over TWO HIP streams.

    The three latency-bound kernels of a step (stream coder, prefix decoder, scatter/merge) keep at most a quarter
    of the CUs busy and the two throughput kernels (entropy maps, VQ) cannot use that idle time inside ONE batch --
    every kernel of a batch depends on the previous one.  Across batches nothing depends on anything: the encode side
    of batch i+1 (entropy -> VQ + router -> stream coder) runs on one stream while the decode side of batch i
    (prefix decode -> merge + gather) runs on the other.  Each side of each slot is one captured hipGraph; slots
    rotate, an event per slot and side carries the only two dependencies (decode i after encode i; encode i + R after
    decode i, because they share the slot's buffers).  Results are bit-identical to the one-stream order: same
    kernels, same inputs, no shared scratch between slots (per-launch tickets are library-owned).

    slots: list of (x [B,3,H,W], z [B,4,H/4,W/4]) device tensors -- the caller refills a slot's tensors in place
    (on `enc_stream`, after `slot.ev_dec`) to feed new data.
    """

    def __init__(self, quantizer, coarse_ratio, medium_ratio, slots, frequency=None, hist=None, decode=True):
        if len(slots) < 2:
            raise ValueError("BatchStream needs at least 2 slots (batch i+1 encodes while batch i decodes)")
        self.pipe = HotPathPipeline(quantizer, coarse_ratio, medium_ratio, frequency=frequency)
        self.hist = hist
        self.decode = bool(decode)
        self.slots = [BatchSlot(x, z) for x, z in slots]
        dev = self.slots[0].x.device
        self.device = dev
        self.enc_stream = torch.cuda.Stream(dev)
        self.dec_stream = torch.cuda.Stream(dev)
        self._next = 0
        self._captured = False

    # the two halves of HotPathPipeline._chain
    def _encode(self, s):
        p = self.pipe
        e8, e16 = entropy_maps(s.x)
        zq, loss, ind, mask, _, mode = vq_forward_route(
            s.z, p.vq.embedding.weight, p.vq.beta, p.vq.legacy, e16, e8,
            p.router.coarse_grain_ratio, p.router.medium_grain_ratio, per_image=True)
        comp = p.codec.compress(ind, mask, mode, hist=self.hist)
        s.enc = {"e8": e8, "e16": e16, "mask": mask, "mode": mode, "z_q": zq, "loss": loss, "ind": ind, "comp": comp}

    def _decode(self, s):
        s.dec = self.pipe.codec.decompress(s.enc["comp"])

    def capture(self, warmup=2):
        """run every slot eagerly (uploads tables, sets function attributes), then capture its two graphs"""
        cur = torch.cuda.current_stream(self.device)
        self.enc_stream.wait_stream(cur)
        with torch.cuda.stream(self.enc_stream):
            for s in self.slots:
                for _ in range(warmup):
                    self._encode(s)
                    if self.decode:
                        self._decode(s)
            for s in self.slots:
                s.g_enc, _ = capture_graph(lambda s=s: self._encode(s), self.enc_stream)
                if self.decode:
                    s.g_dec, _ = capture_graph(lambda s=s: self._decode(s), self.enc_stream)
        cur.wait_stream(self.enc_stream)
        self.dec_stream.wait_stream(cur)
        self._captured = True

    def submit(self, n=1):
        """enqueue the next n batches (slots in rotation); returns immediately"""
        if not self._captured:
            self.capture()
        for _ in range(n):
            s = self.slots[self._next]
            self._next = (self._next + 1) % len(self.slots)
            with torch.cuda.stream(self.enc_stream):
                if s.used and self.decode:
                    self.enc_stream.wait_event(s.ev_dec)        # the slot's previous decode still reads its streams
                s.g_enc.replay()
                s.ev_enc.record(self.enc_stream)
            if self.decode:
                with torch.cuda.stream(self.dec_stream):
                    self.dec_stream.wait_event(s.ev_enc)
                    s.g_dec.replay()
                    s.ev_dec.record(self.dec_stream)
            s.used = True

    def join(self, stream=None):
        """make `stream` (default: the current one) wait for everything submitted so far"""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        stream.wait_stream(self.enc_stream)
        stream.wait_stream(self.dec_stream)

    def fork(self, stream=None):
        """make both pipeline streams wait for `stream` (default: the current one), e.g. after refilling slots"""
        stream = torch.cuda.current_stream(self.device) if stream is None else stream
        self.enc_stream.wait_stream(stream)
        self.dec_stream.wait_stream(stream)


def capture_graph(fn, stream):
    """capture `fn()` (which only enqueues work on the current stream) into a hipGraph on `stream` -> (graph, fn's result).
    The ticket slots the captured launches take from the library's pool are returned when the graph object is
    garbage-collected (_lib.ticket_scope), so a long-lived process can capture per image shape for as long as it likes."""
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
    `launch_threads=True` hands each lane's replays to a thread of its own (graph launches release the GIL): the lanes
    start together instead of one host launch after the other.
    """

    def __init__(self, quantizer, coarse_ratio, medium_ratio, slots, lanes=4, frequency=None, hist=None, decode=True,
                 graph=True, ring=True, fuse_router=True, decoder=None, max_ring=8, launch_threads=False, quick_start=False,
                 prepare=True, native_launch=1):
        if not slots:
            raise ValueError("LaneStream needs at least one slot")
        # prepare: a stream of batches is inference against ONE codebook -- its image is made once, when capture() runs
        # (a captured launch holds a snapshot of the codebook either way: HotPathPipeline.prepare)
        self.pipe = HotPathPipeline(quantizer, coarse_ratio, medium_ratio, frequency=frequency, fuse_router=fuse_router, prepare=prepare)
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
        # quick_start: a lane's first run of a submit is split into (1 step) + (the rest): a graph launch costs the host
        # ~12 us + ~0.55 us per kernel node, and lane j only starts after the launches of lanes 0..j-1 (measured: no gain at
        # K=20 -- the window is bound by resources, not by the last lane's start -- hence off by default)
        self.quick_start = bool(quick_start) and nl > 1
        # native_launch: 1 (default) = all graph launches of a submit in ONE C call (cgic_launch_graphs: no interpreter between
        # them: host time of a K=20 submit 115 -> 93 us, the window 777 -> 756 us), 2 = ... each lane's launches on its own
        # persistent C thread (host 60 us, the window no shorter: the runtime serialises the launches), 0 = replay() from Python
        self.native_launch = int(native_launch) if self.graph and hasattr(torch.cuda.CUDAGraph, "raw_cuda_graph_exec") else 0
        self._pool = None
        if launch_threads and nl > 1:
            from concurrent.futures import ThreadPoolExecutor
            self._pool = ThreadPoolExecutor(max_workers=nl, thread_name_prefix="cgic-lane")
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

    def _plan(self, n):
        """[(lane, start, count), ...] per lane for the next n batches, without advancing"""
        L = len(self.lanes)
        todo = [0] * L
        for t in range(self._t, self._t + n):
            todo[t % L] += 1
        plans = []
        for j, lane in enumerate(self.lanes):
            m, pos, runs = len(lane["slots"]), lane["pos"], []
            c = todo[j]
            while c:
                k = min(c, self.max_ring)
                if self.quick_start and not runs and k > 2:
                    k = 1                                  # every lane starts after one short launch
                runs.append((pos, k))
                pos = (pos + k) % m
                c -= k
            plans.append(runs)
        return plans

    def prepare(self, n):
        """capture (outside any timed region) every graph the next submit(n) will replay"""
        if not self._captured:
            self.capture()
        if self.graph:
            plans = self._plan(n)
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
            execs, streams, lane_of = [], [], []
            depth = max(len(r) for r in plans)
            for i in range(depth):                                       # one launch per lane and turn, like the Python loop
                for j, (lane, runs) in enumerate(zip(self.lanes, plans)):
                    if i < len(runs):
                        execs.append(self._graph(lane, *runs[i])[0].raw_cuda_graph_exec())
                        streams.append(lane["stream"].cuda_stream)
                        lane_of.append(j)
            n = len(execs)
            cached = ((ctypes.c_void_p * n)(*execs), (ctypes.c_void_p * n)(*streams), (ctypes.c_int * n)(*lane_of), n)
            self.__dict__.setdefault("_native_args", {})[key] = cached
        return cached

    def _launch_native(self, plans):
        e, s, l, n = self._native_args_for(plans)
        with torch.cuda.device(self.device):
            _lib.call("cgic_launch_graphs", e, s, l, n, len(self.lanes) if self.native_launch >= 2 else 1)

    def _run_lane(self, lane, runs):
        with torch.cuda.device(self.device), torch.cuda.stream(lane["stream"]):
            for start, count in runs:
                self._graph(lane, start, count)[0].replay()

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
            elif self._pool is not None:
                for f in [self._pool.submit(self._run_lane, lane, runs) for lane, runs in zip(self.lanes, plans) if runs]:
                    f.result()
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
