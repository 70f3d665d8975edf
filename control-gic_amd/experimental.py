"""Schedules that were built, measured and NOT adopted (NOTES.md has the numbers).  Kept importable for the probes under
tools/probes/ and as documentation of a negative result; nothing in the product path uses this module.

BatchStream (round 2): two-stream software pipelining across batches -- 92.1 vs 92.8 us per step against one stream, because
every event is a graph boundary and the overlapped kernels slow each other down by what they gain.  What did work is
pipeline.LaneStream: whole batches on independent hardware queues, no events at all.
"""
import torch

from .entropy import entropy_maps
from .pipeline import BatchSlot, HotPathPipeline, capture_graph
from .quantize import vq_forward_route


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
