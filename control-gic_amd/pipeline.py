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
    def __init__(self, quantizer, coarse_ratio, medium_ratio, chunks=1, frequency=None, fork_vq=False):
        self.vq = quantizer
        self.router = TripleGrainFixedEntropyRouter(coarse_ratio, medium_ratio, per_image=True)
        self.codec = GrainCodec(frequency if frequency is not None else quantizer.embedding_counter,
                                quantizer.embedding.weight)
        self.chunks = max(1, int(chunks))
        self.fork_vq = int(fork_vq)       # 1: VQ(+hist) on a side stream next to entropy -> router; 2: router on a side stream next to VQ
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
