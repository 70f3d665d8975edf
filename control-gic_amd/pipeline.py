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

from .codec import GrainCodec
from .entropy import entropy_maps
from .quantize import _vq_forward, vq_forward_route
from .router import TripleGrainFixedEntropyRouter


class HotPathPipeline:
    def __init__(self, quantizer, coarse_ratio, medium_ratio, chunks=1, frequency=None, fork_vq=False, fuse_router=True):
        self.vq = quantizer
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

    def _get_streams(self, device):
        if self._streams is None or self._streams[0].device != device:
            self._streams = [torch.cuda.Stream(device) for _ in range(self.chunks)]
        return self._streams

    def _chain(self, x, z, hist, decode):
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
            zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None)
            cur.wait_event(join)
        elif self.fork_vq:
            cur = torch.cuda.current_stream(x.device)
            if self._side is None or self._side.device != x.device:
                self._side = torch.cuda.Stream(x.device)
            fork = torch.cuda.Event()
            fork.record(cur)
            self._side.wait_event(fork)
            with torch.cuda.stream(self._side):
                zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None)
                join = torch.cuda.Event()
                join.record(self._side)
            e8, e16 = entropy_maps(x)
            mask, _, _, mode = self.router(e16, e8, want_gate=False)
            cur.wait_event(join)
        elif not self.fuse_router:
            e8, e16 = entropy_maps(x)
            mask, _, _, mode = self.router(e16, e8, want_gate=False)
            zq, loss, ind = _vq_forward(z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, None)
        else:
            e8, e16 = entropy_maps(x)
            # VQ and the per-image router share one launch (the router rides in the VQ kernel's shadow)
            zq, loss, ind, mask, _, mode = vq_forward_route(
                z, self.vq.embedding.weight, self.vq.beta, self.vq.legacy, e16, e8,
                self.router.coarse_grain_ratio, self.router.medium_grain_ratio, per_image=True)
        comp = self.codec.compress(ind, mask, mode, hist=hist)      # usage histogram rides on the coder launch
        dec = self.codec.decompress(comp) if decode else None
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
                s.g_enc = torch.cuda.CUDAGraph()
                with torch.cuda.graph(s.g_enc, stream=self.enc_stream):
                    self._encode(s)
                if self.decode:
                    s.g_dec = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(s.g_dec, stream=self.enc_stream):
                        self._decode(s)
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
