#!/usr/bin/env python3
"""bench.py -- encode+decode MPixels/s of the Control-GIC hot path on MI355X.

One "step" = one pass of the hot path over one batch of synthetic input already resident in HBM:
  encode: entropy maps (p8+p16) -> [VQ forward (indices, z_q, loss) + per-image router, one launch]
          -> masked select + Huffman + mask packing  (the five .bin streams per image)
  decode: prefix decode -> mask/index scatter + x2/x4 merge -> embedding gather
Workload = BASELINE.json configs[1]: batch 64 of 256x256, codebook 1024x4, ratio (0.1, 0.8, 0.1).
The conv encoder/decoder of the codec are out of scope: `z` is a
synthetic N(0,1) latent standing in for the encoder output.

The K timed steps are K successive, DISTINCT batches: the inputs rotate through enough slots to exceed the
256 MiB Infinity Cache, so the image bytes of every step come from HBM.  The whole rotation of batches is captured
back to back in ONE hipGraph on one stream (successive graph launches are ~7 us apart on the device; `--no-ring`: one
hipGraph per batch; what is left of K after whole rotations always runs as per-batch graphs) (`--schedule pipelined`: control_gic_amd.experimental.BatchStream, encode side of batch i+1 next to the
decode side of batch i on two HIP streams -- measured no faster).  `value` = all pixels of the K
steps / wall time between two barrier + synchronize brackets, max over ranks.

  python bench.py [--gpus N] [--steps K] [--warmup W]
N > 1: one rank per GPU.  Under torch.distributed.run the ranks come from the environment; launched plainly
(`python bench.py --gpus N`) the script spawns the N ranks itself.  Images shard across ranks with no data-path
collective; ONE RCCL all-reduce of the int64[1024] usage histogram per run.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3     # MI355X_MICROARCH.md: fp32-input MFMA == fp32 vector peak
PEAK_HBM_GBS = 8000.0
IC_BYTES = 256 << 20             # Infinity Cache: the rotating inputs must exceed it
RATIO_SWEEP = [(0.0, 0.0), (0.0, 0.5), (0.0, 1.0), (0.25, 0.25), (0.1, 0.8), (0.5, 0.0), (0.5, 0.5), (0.7, 0.3),
               (0.3, 0.7), (1.0, 0.0)]          # all seven modes


def log(*a):
    print(*a, file=sys.stderr, flush=True)


class stdout_to_stderr:
    """RCCL prints a version banner on the process's stdout (fd 1) when a communicator is created; the
    contract is ONE JSON line on stdout, so fd 1 points at stderr while that happens."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)

    def __exit__(self, *exc):
        sys.stdout.flush()
        try:                                               # the banner sits in libc's stdout buffer (fully buffered when fd 1 is a
            import ctypes                                  # pipe or a file): push it out while fd 1 still points at stderr
            ctypes.CDLL(None).fflush(None)
        except Exception:                                  # noqa: BLE001
            pass
        os.dup2(self._saved, 1)
        os.close(self._saved)


def zipf_freq():
    return np.floor(2.0e6 / (1 + np.arange(1024)) ** 1.1).astype(np.int64)


def make_inputs(B, H, W, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((B, 3, H, W), dtype=np.float32)
    z = rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32)
    cb = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)   # same codebook on every rank
    return x, z, cb


def make_quantizer(dev, cb):
    import control_gic_amd as cg
    vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev).eval()
    vq.embedding.weight.data.copy_(torch.from_numpy(cb))
    vq.usage_counter.copy_(torch.from_numpy(zipf_freq().astype(np.float32)))
    return vq


class HotPath:
    """one batch: pre-built modules + a step() that only enqueues work (graph-capturable)"""

    def __init__(self, dev, x, z, cb, ratio, vq=None, codec=None, fuse_router=True):
        import control_gic_amd as cg
        self.cg = cg
        self.x = torch.from_numpy(x).to(dev) if isinstance(x, np.ndarray) else x
        self.z = torch.from_numpy(z).to(dev) if isinstance(z, np.ndarray) else z
        self.vq = vq if vq is not None else make_quantizer(dev, cb)
        self.codec = codec if codec is not None else cg.GrainCodec(self.vq.embedding_counter, self.vq.embedding.weight)
        self.router = cg.TripleGrainFixedEntropyRouter(ratio[0], ratio[1], per_image=True)
        self.hist = torch.zeros(1024, dtype=torch.int64, device=dev)
        self.out = None
        self.pipe = cg.pipeline.HotPathPipeline(self.vq, ratio[0], ratio[1], frequency=self.codec.huffman, fuse_router=fuse_router, prepare=True)

    def step(self):
        r = self.pipe.run(self.x, self.z, self.hist, decode=True)[0]
        self.out = (r["e8"], r["e16"], r["mask"], r["mode"], r["z_q"], r["ind"], r["comp"], *r["dec"])

    def capture(self):
        for _ in range(2):
            self.step()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            self.step()
        g, _ = self.cg.capture_graph(self.step, side)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = g
        return g


def time_events(fn, iters):
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    e.synchronize()
    return s.elapsed_time(e) * 1e3 / iters      # microseconds per call


def graph_kernel_time(fn, per_graph=20, reps=5):
    """average device time of one launch of fn's kernel(s): `per_graph` back-to-back launches captured in a hipGraph
    (no host-bound gaps), HIP events around the replay on the stream it runs on"""
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    import control_gic_amd as cg

    def body():
        for _ in range(per_graph):
            fn()
    g, _ = cg.capture_graph(body, side)
    with torch.cuda.stream(side):
        g.replay()
        side.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(side)
        for _ in range(reps):
            g.replay()
        e.record(side)
        e.synchronize()
    torch.cuda.current_stream().wait_stream(side)
    return s.elapsed_time(e) * 1e3 / (per_graph * reps)


def stage_breakdown(hp):
    """per-stage device time, microseconds per launch (each stage alone, graph-timed on its launch stream)"""
    from control_gic_amd.quantize import _vq_forward, vq_forward_route
    cg = hp.cg
    e8, e16 = cg.entropy_maps(hp.x)
    mask, _, _, mode = hp.router(e16, e8, want_gate=False)
    prep = hp.pipe.prepared                   # the codebook image the timed launches use (made once per codebook)
    _, _, ind = _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None)
    comp = hp.codec.compress(ind, mask, mode)
    st = {}
    st["entropy_maps"] = graph_kernel_time(lambda: cg.entropy_maps(hp.x))
    st["router_alone"] = graph_kernel_time(lambda: hp.router(e16, e8, want_gate=False))
    st["vq+router_fused_launch"] = graph_kernel_time(lambda: vq_forward_route(
        hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio, prepared=prep, pixels=hp.x))
    st["vq+router_fused_launch_no_refinement"] = graph_kernel_time(lambda: vq_forward_route(
        hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio, prepared=prep))
    st["vq+router_fused_launch_unprepared"] = graph_kernel_time(lambda: vq_forward_route(
        hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio, pixels=hp.x))
    st["vq_kernel_alone"] = graph_kernel_time(lambda: _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None, prepared=prep))
    st["vq_kernel_indices_only"] = graph_kernel_time(lambda: _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None, False, False, prepared=prep))
    st["compress_streams+hist"] = graph_kernel_time(lambda: hp.codec.compress(ind, mask, mode, hist=hp.hist))
    st["decompress_streams"] = graph_kernel_time(lambda: hp.codec.decompress(comp))
    # frames that arrive as uint8 [B,H,W,3] (the datasets' PIL images): ToTensor + both maps in one pass against torch's ToTensor
    # on the GPU followed by entropy_maps (extra data points; the timed step takes the fp32 tensor BASELINE's config names)
    try:
        frames = (hp.x.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous()
        st["u8_frames:totensor+entropy_maps_fused"] = graph_kernel_time(lambda: cg.entropy_maps_u8(frames))
        st["u8_frames:torch_totensor"] = graph_kernel_time(lambda: frames.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous())
    except Exception as e:          # noqa: BLE001 -- an extra data point never fails the bench line
        st["u8_frames:error"] = -1.0
    return {k: round(v, 2) for k, v in st.items()}


def saturated_launch_time(hp, lanes=4, per_graph=20, reps=6):
    """microseconds per launch of the fused VQ + router kernel when `lanes` independent streams replay graphs of it alone: what the
    launch costs the GPU with other batches in flight (its launch, ramp and tail hide behind the other streams' launches)"""
    from control_gic_amd.quantize import vq_forward_route
    from control_gic_amd.pipeline import GraphLanes
    e8, e16 = hp.cg.entropy_maps(hp.x)
    zs = [hp.z] + [hp.z.clone() for _ in range(lanes - 1)]

    def make(z):
        def fn():
            out = None
            for _ in range(per_graph):
                out = vq_forward_route(z, hp.vq.embedding.weight, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio,
                                       prepared=hp.pipe.prepared, pixels=hp.x)
            return out
        return fn
    gl = GraphLanes(hp.x.device, [make(z) for z in zs])
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl.replay(1)
        gl.join()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6 / (per_graph * lanes)


def saturated_stage_times(hp, lanes=4, per_graph=20, reps=4):
    """resource time of every launch of the step: microseconds per launch when `lanes` independent streams replay graphs of THAT
    launch alone (its ramp and tail hide behind the other streams' launches).  Their sum is what a step costs the GPU when the
    lanes overlap perfectly: the yardstick next to ms_per_step."""
    from control_gic_amd.quantize import vq_forward_route, _vq_forward
    from control_gic_amd.pipeline import GraphLanes
    cg = hp.cg
    e8, e16 = cg.entropy_maps(hp.x)
    mask, _, _, mode = hp.router(e16, e8, want_gate=False)
    _, _, ind = _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None)
    comp = hp.codec.compress(ind, mask, mode)
    w, prep = hp.vq.embedding.weight, hp.pipe.prepared
    stages = {
        "entropy_maps": lambda: cg.entropy_maps(hp.x),
        "vq+router": lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio,
                                              prepared=prep, pixels=hp.x),
        "compress_streams+hist": lambda: hp.codec.compress(ind, mask, mode, hist=hp.hist),
    }
    with cg.decoder_mode("throughput"):
        stages["decode+merge"] = lambda: hp.codec.decompress(comp)
        out = {}
        for name, call in stages.items():
            def make(call=call):
                def fn():
                    r = None
                    for _ in range(per_graph):
                        r = call()
                    return r
                return fn
            gl = GraphLanes(hp.x.device, [make() for _ in range(lanes)])
            best = 1e9
            for _ in range(reps):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                gl.replay(1)
                gl.join()
                torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
            out[name] = round(best * 1e6 / (per_graph * lanes), 2)
    return out


def cpu_baseline(x, z, cb, ratio, budget_s=12.0, threads=None):
    """the oracle (a scalar C port of the reference algorithm) on the HOST'S CORES: the same workload, images in parallel over
    `threads` worker threads (the C calls release the GIL; default: every core the process may run on, at most 64), bounded
    sample; the one-core figure of the same port is kept next to it.  SURVEY.md 8(d): the reference itself runs on torch's
    thread pool (inference.py:157-170)."""
    from concurrent.futures import ThreadPoolExecutor
    from oracle import cgic_oracle as orc
    htab = orc.HuffmanTable(zipf_freq())
    H, W = x.shape[2], x.shape[3]
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count() or 1
    threads = max(1, min(int(threads) if threads else avail, 64))

    def one(b):
        e8 = orc.entropy(x[b:b + 1], 8)
        e16 = orc.entropy(x[b:b + 1], 16)
        mc, mm, mf, _, mode = orc.router(e16, e8, ratio[0], ratio[1], per_image=True, want_gate=False)
        _, _, idx = orc.vq(z[b:b + 1], cb)
        ind = idx.reshape(H // 4, W // 4)
        streams = orc.compress_image(ind, mc[0, 0], mm[0, 0], mf[0, 0], mode, htab)
        dind, _, _, _ = orc.decompress_image(streams, mode, H // 4, W // 4, htab)
        orc.gather(dind, cb)
        return streams

    def timed(nthreads, budget):
        stop = [False]
        counts = [0] * nthreads

        def worker(k):
            i = k
            while not stop[0]:
                one(i % x.shape[0])
                counts[k] += 1
                i += nthreads

        t0 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=nthreads) as ex:
            futs = [ex.submit(worker, k) for k in range(nthreads)]
            time.sleep(budget)
            stop[0] = True
            for f in futs:
                f.result()
        dt = time.perf_counter() - t0
        return sum(counts), dt

    one(0)                                   # warm caches / page in the library
    n1, dt1 = timed(1, budget_s / 3)
    nN, dtN = timed(threads, budget_s) if threads > 1 else (n1, dt1)
    one_core = round(n1 * H * W / dt1 / 1e6, 4)
    return {"value": round(nN * H * W / dtN / 1e6, 4), "unit": "MPixels/s", "cores": threads, "kind": "port",
            "cpu_baseline_1core": one_core,
            "sample": f"{nN} images of the same workload ({H}x{W} each, cycling through the batch), encode+decode hot "
                      f"path, oracle/cgic_oracle.c, {threads} worker threads (one image each at a time) for {dtN:.1f} s; one thread: {n1} images "
                      f"in {dt1:.1f} s; host has {os.cpu_count()} logical cores, {avail} available to this process"}


def check_against_oracle(out, x, z, cb, ratio, images=None):
    """bitstream / index / mask parity of a finished step with the oracle, on EVERY image of the batch (outside the
    timed region).  Round 4: the masks -- hence every byte -- are checked FROM THE PIXELS: oracle entropy maps in the
    reference's own arithmetic (cgic_oracle_entropy_ref) -> oracle router -> oracle coder against what the timed step's
    default kernels left (entropy_maps -> router with its threshold-band refinement).  The maps themselves: <= 2e-6.
    Returns (all equal, list of per-image bpp)."""
    from oracle import cgic_oracle as orc
    e8, e16, mask, mode, zq, ind, comp, dind, dmask, dq, status = out
    torch.cuda.synchronize()
    H, W = x.shape[2], x.shape[3]
    h, w = H // 4, W // 4
    B = x.shape[0]
    images = range(B) if images is None else images
    htab = orc.HuffmanTable(zipf_freq())
    host = comp.to_host()
    ind_h = ind.view(-1, h, w).cpu().numpy()
    dind_h = dind.cpu().numpy()
    dq_h = dq.cpu().numpy()
    mk = [m.cpu().numpy() for m in mask]
    e8_h, e16_h = e8.cpu().numpy(), e16.cpu().numpy()
    ok = int(status.abs().max()) == 0
    bpp = []
    for b in images:
        _, _, oidx = orc.vq(z[b:b + 1], cb)
        oidx = oidx.reshape(h, w)
        ok = ok and bool(np.array_equal(ind_h[b], oidx))
        o8, o16 = orc.entropy_ref(x[b:b + 1], 8), orc.entropy_ref(x[b:b + 1], 16)
        ok = ok and float(np.abs(o8 - e8_h[b:b + 1]).max()) < 2e-6 and float(np.abs(o16 - e16_h[b:b + 1]).max()) < 2e-6
        omc, omm, omf, _, omode = orc.router(o16, o8, ratio[0], ratio[1])
        ok = ok and omode == mode and all(np.array_equal(mk[g][b, 0], o[0, 0]) for g, o in enumerate((omc, omm, omf)))
        ref = orc.compress_image(oidx, omc[0, 0], omm[0, 0], omf[0, 0], mode, htab)
        ok = ok and host[b] == ref
        oind, _, _, _ = orc.decompress_image(ref, mode, h, w, htab)
        ok = ok and bool(np.array_equal(dind_h[b], oind)) and bool(np.array_equal(dq_h[b], orc.gather(oind, cb)[0]))
        bpp.append(sum(len(v) for v in host[b].values()) * 8 / (H * W))
    return bool(ok), bpp


# ------------------------------------------------------------------------------------------------ extras (rank 0, N=1)
def mask_mismatch(hp, x, z, cb, ratio):
    """pixels -> masks -> bytes on the GPU against the CPU oracle from the SAME pixels, on the benchmark
    batch as the timed step left it.  Oracle side: entropy maps in the reference's own arithmetic (cgic_oracle_entropy_ref: torch's
    operation order, exp / log correctly rounded) -> oracle router -> oracle coder.  (Round 3 compared against the libm / sequential-
    sum oracle maps: both sides were ~1e-6 off the reference's arithmetic and agreed on noise only because noise has no ties.)"""
    from oracle import cgic_oracle as orc
    e8, e16, mask, mode, zq, ind, comp = hp.out[:7]
    torch.cuda.synchronize()
    B, H, W = x.shape[0], x.shape[2], x.shape[3]
    h, w = H // 4, W // 4
    htab = orc.HuffmanTable(zipf_freq())
    host = comp.to_host()
    mk = [m.cpu().numpy() for m in mask]
    ind_h = ind.view(-1, h, w).cpu().numpy()
    diff_elems = diff_images = diff_files = files = 0
    max_de = 0.0
    for b in range(B):
        o8, o16 = orc.entropy_ref(x[b:b + 1], 8), orc.entropy_ref(x[b:b + 1], 16)
        max_de = max(max_de, float(np.abs(o8 - e8[b:b + 1].cpu().numpy()).max()), float(np.abs(o16 - e16[b:b + 1].cpu().numpy()).max()))
        omc, omm, omf, _, omode = orc.router(o16, o8, ratio[0], ratio[1])
        d = sum(int((mk[g][b, 0] != o[0, 0]).sum()) for g, o in enumerate((omc, omm, omf)))
        diff_elems += d
        diff_images += d > 0
        ref = orc.compress_image(ind_h[b], omc[0, 0], omm[0, 0], omf[0, 0], omode, htab)
        files += len(ref)
        diff_files += sum(host[b].get(k) != v for k, v in ref.items())
    n_elems = B * (h * w + h * w // 4 + h * w // 16)
    return {"images": B, "mask_elements": n_elems, "differing_mask_elements": diff_elems, "images_with_a_difference": int(diff_images),
            "bin_files": files, "differing_bin_files": int(diff_files), "max_abs_entropy_diff": max_de,
            "note": "GPU entropy maps -> GPU router (threshold-band refinement) vs oracle reference-arithmetic maps -> oracle router on the same pixels; given equal masks every byte is identical (bpp_match)"}


def mask_flip_families(dev, z, cb, vq, codec, ratio):
    """pixels -> masks -> bytes on TIE-HEAVY content (round-2 verdict, item 3): four synthetic 8-bit families of 64 images of
    256x256 (quantised noise, smooth gradients + faint texture, flat regions with edges, 8x8-blocky) and 8 tiles of 768x768,
    GPU entropy -> GPU router -> GPU coder against the REFERENCE'S OWN ARITHMETIC for the entropy maps (oracle/entropy_torch.py:
    the same torch CPU operators on the same shapes as CGIC/models/model.py:433-483, pinned bit for bit against the real class
    in the build container) -> oracle router -> oracle coder, same latent indices on both sides."""
    from oracle import cgic_oracle as orc, entropy_torch as et
    from oracle.content_families import families
    from control_gic_amd.quantize import vq_forward_route
    htab = orc.HuffmanTable(zipf_freq())

    def run(x, zz, reference_order=False, refine=True):
        B, H, W = x.shape[0], x.shape[2], x.shape[3]
        h, w = H // 4, W // 4
        xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(zz).to(dev)
        e8, e16 = cg_entropy(xd, reference_order=reference_order)
        _, _, ind, mask, _, mode = vq_forward_route(zd, vq.embedding.weight, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True,
                                                    pixels=xd if refine else None)
        host = codec.compress(ind, mask, mode).to_host()
        torch.cuda.synchronize()
        mk = [m.cpu().numpy() for m in mask]
        ind_h = ind.view(-1, h, w).cpu().numpy()
        g8, g16 = e8.cpu().numpy(), e16.cpu().numpy()
        elems = imgs = files = dfiles = 0
        dmax = 0.0
        for b0 in range(0, B, 8):
            xt = torch.from_numpy(x[b0:b0 + 8])
            r8, r16 = et.entropy_map(xt, 8).numpy(), et.entropy_map(xt, 16).numpy()
            dmax = max(dmax, float(np.abs(r8 - g8[b0:b0 + 8]).max()), float(np.abs(r16 - g16[b0:b0 + 8]).max()))
            for i in range(r8.shape[0]):
                b = b0 + i
                omc, omm, omf, _, omode = orc.router(r16[i:i + 1], r8[i:i + 1], ratio[0], ratio[1])
                d = sum(int((mk[k][b, 0] != o[0, 0]).sum()) for k, o in enumerate((omc, omm, omf)))
                elems += d
                imgs += d > 0
                ref = orc.compress_image(ind_h[b], omc[0, 0], omm[0, 0], omf[0, 0], omode, htab)
                files += len(ref)
                dfiles += sum(host[b].get(k) != v for k, v in ref.items())
        return {"images": B, "size": f"{W}x{H}", "differing_mask_elements": elems, "mask_elements": B * (h * w + h * w // 4 + h * w // 16),
                "images_with_a_difference": int(imgs), "bin_files": files, "differing_bin_files": int(dfiles), "max_abs_entropy_diff": dmax}

    from control_gic_amd import entropy_maps as cg_entropy
    out = {}
    alone = {}
    short = lambda r: {k: r[k] for k in ("images", "differing_mask_elements", "images_with_a_difference", "differing_bin_files", "max_abs_entropy_diff")}

    def launch_us(x, zz):
        """the fused VQ + router launch on this content, with and without the refinement (what the band costs)"""
        xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(zz).to(dev)
        e8, e16 = cg_entropy(xd)
        f = lambda px, q=False: graph_kernel_time(lambda: vq_forward_route(zd, vq.embedding.weight, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True,
                                                                           pixels=px, refine_queues=q), per_graph=5, reps=3)
        import control_gic_amd as cg
        auto = cg.pipeline.HotPathPipeline(vq, ratio[0], ratio[1], frequency=codec.huffman).decide(xd)
        tq, tp = round(f(xd, True), 2), round(f(xd, False), 2)
        # vq+router_us: what a stream of such batches runs (the variant refine_queues="auto" picks from the content at capture)
        return {"vq+router_us": tq if auto else tp, "vq+router_plain_kernel_us": tp, "vq+router_refinement_queues_us": tq,
                "auto_picks_queues": bool(auto), "vq+router_no_refinement_us": round(f(None), 2)}

    fam = families(n=64)
    for name, x in fam.items():
        out[name] = run(x, z)
        out[name].update(launch_us(x, z))
        alone[name] = short(run(x, z, refine=False))
    t = families(n=2, H=768, W=768, seed=11)
    tiles = np.concatenate([t[k] for k in ("noise8", "smooth8", "flat_edges", "blocky8")])
    zt = np.random.default_rng(5).standard_normal((tiles.shape[0], 4, 192, 192), dtype=np.float32)
    out["tiles_768"] = run(tiles, zt)
    out["tiles_768"].update(launch_us(tiles, zt))
    alone["tiles_768"] = short(run(tiles, zt, refine=False))
    alone["note"] = "the same launches WITHOUT the pixels (round 3's default): the maps of the default kernel decide as given"
    out["maps_alone"] = alone
    xd = torch.from_numpy(fam["smooth8"]).to(dev)
    out["reference_order_kernel"] = {
        "smooth8": short(run(fam["smooth8"], z, reference_order=True, refine=False)),
        "us_per_launch_B64_256x256": round(graph_kernel_time(lambda: cg_entropy(xd, reference_order=True), per_graph=5, reps=3), 2),
        "default_kernel_us_per_launch": round(graph_kernel_time(lambda: cg_entropy(xd), per_graph=5, reps=3), 2),
        "note": "entropy_maps(reference_order=True) = cgic_entropy_maps_ref_f32: every patch in the reference's arithmetic (opt-in; the "
                "timed step evaluates only the patches inside a threshold band that way)"}
    out["note"] = ("DEFAULT path = what the timed step runs: entropy_maps -> [VQ + router with the pixels: patches within 4e-6 of a threshold "
                   "(k-th smallest value, strict '<') are re-evaluated in the reference's own arithmetic inside the router].  Reference side = the "
                   "reference's torch-CPU entropy arithmetic evaluated on this host (oracle/entropy_torch.py) -> oracle router -> oracle coder")
    return out


def lanes_rate(vq, codec, ratio, xz, lanes, steps, copies=1):
    """batches per second of a LaneStream over the given device (x, z) pairs (each used `copies` times as its own slot:
    distinct output buffers, same inputs)"""
    import control_gic_amd as cg
    slots = [p for p in xz for _ in range(copies)]
    ls = cg.pipeline.LaneStream(vq, ratio[0], ratio[1], slots, lanes=lanes, frequency=codec.huffman)
    ls.capture()
    ls.submit(len(slots)); ls.join(); torch.cuda.synchronize()
    ls.prepare(steps); torch.cuda.synchronize()
    t0 = time.perf_counter()
    ls.submit(steps); ls.join(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps, ls


def content_8bit(dev, z, cb, vq, codec, ratio, value, steps=64):
    """The timed step on the content the reference feeds the path: 8-bit images through ToTensor (inference.py:62-67), thresholds =
    k-th smallest entropies under a strict '<' (RouterTriple.py:21-34).  Three synthetic 8-bit families (oracle/content_families.py:
    quantised noise, smooth gradients + faint texture, flat regions with edges), each as a LaneStream of 8 slots (4 distinct batches
    of 64 images x 2, four in flight, same graphs as `value`), MPixel/s by wall time, and the parity of what the stream left behind
    checked from the pixels against the oracle (reference-arithmetic maps -> router -> coder) on a sample of images of two slots."""
    from oracle.content_families import families
    out = {}
    zd = torch.from_numpy(z).to(dev)
    worst = None
    for name in ("noise8", "smooth8", "flat_edges"):
        xs = [families(n=64, seed=7 + 13 * k)[name] for k in range(4)]
        xz = [(torch.from_numpy(x).to(dev), zd) for x in xs]
        dt, ls = lanes_rate(vq, codec, ratio, xz, 4, steps, copies=2)
        torch.cuda.synchronize()
        ok = True
        for k in (0, 7):
            okk, _ = check_against_oracle(slot_out(ls, k), xs[k // 2], z, cb, ratio, images=range(k, 64, 8))
            ok = ok and okk
        mp = 64 * 65536 / dt / 1e6
        out[name] = {"MPixels/s": round(mp, 1), "frac_of_value": round(mp / value, 3), "bpp_match": bool(ok)}
        worst = mp if worst is None else min(worst, mp)
        del ls
    out["min_MPixels/s"] = round(worst, 1)
    out["min_frac_of_value"] = round(worst / value, 3)
    out["note"] = ("fp32 images with 8-bit values (what ToTensor hands over), 4 distinct batches x 2 slots per family, four batches in flight, "
                   f"{steps} steps; parity from the pixels on 16 images per family")
    return out


def ratio_sweep(dev, x, z, cb, vq, codec, steps=30):
    out = []
    B, H, W = x.shape[0], x.shape[2], x.shape[3]
    xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
    for r in RATIO_SWEEP:
        hp = HotPath(dev, xd, zd, cb, r, vq=vq, codec=codec)
        g = hp.capture()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        ok, bpp = check_against_oracle(hp.out, x, z, cb, r, images=range(0, B, 16))
        dt4, _ = lanes_rate(vq, codec, r, [(xd, zd)], 4, 4 * steps, copies=4)
        out.append({"ratio": [r[0], r[1], round(1 - r[0] - r[1], 6)], "mode": int(hp.out[3]), "MPixels/s": round(steps * B * H * W / dt / 1e6, 1),
                    "ms_per_step": round(dt / steps * 1e3, 5), "MPixels/s_4_in_flight": round(B * H * W / dt4 / 1e6, 1),
                    "bpp_mean": round(float(np.mean(hp.out[6].bpp(H * W))), 6), "bpp_match": ok})
    return out


def input_regimes(dev, x, ratio, steps=30):
    """SURVEY.md 8(d) config 2 beyond N(0,1) latents against an N(0,1) codebook -- the regimes in which the candidate filter has
    near-ties to resolve and the streams are short: the reference's INIT codebook U(+-1/1024) (quantize.py:26) with latents of
    the same scale, a clustered codebook of near-duplicate rows, and latents whose index histogram follows the Huffman table
    (Zipf): same step, same kernels, parity of a sample of images checked against the oracle for each.  Telemetry per variant:
    vectors that needed the all-K exact scan, groups rerun on the exact loop, groups with a second candidate set (cgic_vq_stats),
    the self-synchronising decoder's sweeps per image (cgic_decode_stats), bits per coded symbol."""
    import control_gic_amd as cg
    B, H, W = x.shape[0], x.shape[2], x.shape[3]
    N = B * (H // 4) * (W // 4)
    rng = np.random.default_rng(77)
    cb_n = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)
    u = lambda shape: ((rng.random(shape, dtype=np.float32) * 2 - 1) / np.float32(1024)).astype(np.float32)
    centres = rng.standard_normal((64, 4), dtype=np.float32)
    cb_cl = (centres[rng.integers(0, 64, 1024)] + np.float32(1e-4) * rng.standard_normal((1024, 4), dtype=np.float32)).astype(np.float32)
    p = zipf_freq().astype(np.float64); p /= p.sum()
    pick = rng.choice(1024, size=(B, H // 4, W // 4), p=p)
    z_zipf = (cb_n[pick] + np.float32(0.02) * rng.standard_normal((B, H // 4, W // 4, 4), dtype=np.float32)).transpose(0, 3, 1, 2).copy()
    variants = [
        ("normal", rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32), cb_n, "z ~ N(0,1), codebook ~ N(0,1): the timed step's inputs"),
        ("init_uniform", u((B, 4, H // 4, W // 4)), u((1024, 4)), "z, codebook ~ U(+-1/1024): the reference's codebook at initialisation (quantize.py:26)"),
        ("clustered_codebook", rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32), cb_cl,
         "codebook = 64 centres ~ N(0,1), 16 near-duplicates each (1e-4 apart): many runner-up tiles inside the margin"),
        ("zipf_matched", z_zipf, cb_n, "z = codebook rows drawn from the Huffman table's own distribution + 0.02 N(0,1): short streams"),
    ]
    xd = torch.from_numpy(x).to(dev)
    lib = cg._lib.lib()
    out = {}
    base = None
    for name, z, cb, what in variants:
        vq = make_quantizer(dev, cb)
        codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
        zd = torch.from_numpy(z).to(dev)
        hp = HotPath(dev, xd, zd, cb, ratio, vq=vq, codec=codec)
        g = hp.capture()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            g.replay()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        ok, bpp = check_against_oracle(hp.out, x, z, cb, ratio, images=range(0, B, 16))
        dt4, _ = lanes_rate(vq, codec, ratio, [(xd, zd)], 4, 4 * steps, copies=4)
        # telemetry: one eager step with the counters on (throughput decoder = what the lanes run)
        cnt = torch.zeros(8, dtype=torch.int32, device=dev)
        lib.cgic_vq_stats(cnt.data_ptr()); lib.cgic_decode_stats(cnt.data_ptr() + 16)
        try:
            with cg.decoder_mode("throughput"):
                r = hp.pipe.run(hp.x, hp.z, None, decode=True)[0]
            torch.cuda.synchronize()
        finally:
            lib.cgic_vq_stats(None); lib.cgic_decode_stats(None)
        c = cnt.cpu().numpy().astype(np.int64)
        nsym = int(sum((m.sum() for m in r["mask"])))
        nbits = float(np.sum(r["comp"].bpp(H * W))) * H * W          # all five streams of all images, headers and mask streams included
        fused = graph_kernel_time(lambda: vq_route_call(hp), per_graph=10, reps=3)
        out[name] = {"what": what, "MPixels/s": round(B * H * W / dt / 1e6, 1), "ms_per_step": round(dt * 1e3, 5),
                     "MPixels/s_4_in_flight": round(B * H * W / dt4 / 1e6, 1), "vq+router_us": round(fused, 2),
                     "flagged_vector_fraction": round(float(c[0]) / N, 6), "fallback_groups": int(c[1]), "second_set_groups": int(c[2]),
                     "groups": N // 64, "decoder_sweeps_mean": round(float(c[4]) / max(int(c[5]), 1), 2), "decoder_sweeps_max": int(c[6]),
                     "bits_per_symbol_incl_masks": round(nbits / max(nsym, 1), 2), "bpp_mean": round(float(np.mean(bpp)), 6), "bpp_match": ok}
        if base is None:
            base = out[name]
        else:
            out[name]["slowdown_vs_normal_4_in_flight"] = round(base["MPixels/s_4_in_flight"] / out[name]["MPixels/s_4_in_flight"], 3)
    return out


def vq_route_call(hp):
    from control_gic_amd.quantize import vq_forward_route
    e8, e16 = hp.out[0], hp.out[1]
    return vq_forward_route(hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, hp.router.coarse_grain_ratio, hp.router.medium_grain_ratio,
                            prepared=hp.pipe.prepared, pixels=hp.x)


def div2k_image(dev, cb, vq, codec, iters=8):
    """one 2040x1356 image (DIV2K-typical) through the tiling driver of inference_high_resolution.py: zero-pad to x16,
    768-px grid (6 tiles in 4 shape groups), per-tile routing, same-shape tiles batched, pad / stack copies included;
    eager launches (shapes differ per group)"""
    from control_gic_amd import highres
    from control_gic_amd.quantize import vq_forward_route
    H, W = 1356, 2040
    rng = np.random.default_rng(4)
    x = torch.from_numpy(rng.random((1, 3, H, W), dtype=np.float32)).to(dev)
    zs = {}

    def encode(tiles):
        import control_gic_amd as cg
        T, _, th, tw = tiles.shape
        key = (T, th, tw)
        if key not in zs:
            zs[key] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
        e8, e16 = cg.entropy_maps(tiles)
        _, _, ind, mask, _, mode = vq_forward_route(zs[key], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
        return ind, mask, mode

    def once():
        tiled = highres.compress_tiled(x, encode, codec)
        per_tile, _ = highres.decompress_tiled(tiled, codec)
        return tiled, per_tile

    tiled, per_tile = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(iters):
        tiled, per_tile = once()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / iters
    # round trip: every tile's masks come back exactly, indices wherever the fine grain kept them
    ok = True
    for idxs, _, (ind0, masks0, _) in tiled.groups:
        T = len(idxs)
        fine = masks0[2].reshape(T, -1).bool()
        got = torch.cat([per_tile[t][0].reshape(1, -1) for t in idxs])
        ok = ok and bool(torch.equal(got[fine], ind0.reshape(T, -1)[fine]))
    res = {"workload": f"one {W}x{H} image via highres.compress_tiled + decompress_tiled ({len(tiled.tiles)} tiles, {len(tiled.groups)} shape groups), eager, incl. pad/stack and the host sync of decompress_tiled",
           "MPixels/s": round(H * W / dt / 1e6, 1), "ms_per_image": round(dt * 1e3, 4), "bpp": round(tiled.bpp(), 6), "round_trip_ok": ok}
    # the same image as ONE hipGraph: the four shape groups on parallel streams (highres concurrent=True), no host
    # synchronisation inside (the decoder status is read after the replay)
    try:
        def once_graph():
            t = highres.compress_tiled(x, encode, codec, concurrent=True)
            p, st = highres.decompress_tiled(t, codec, concurrent=True, check=False)
            return t, p, st
        once_graph(); torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        import control_gic_amd as cg
        g, (tg, pg, stg) = cg.capture_graph(once_graph, side)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * iters):
            g.replay()
        torch.cuda.synchronize()
        dtg = (time.perf_counter() - t0) / (4 * iters)
        same = int(stg.abs().max()) == 0 and tg.streams() == tiled.streams()
        res["graph_replay"] = {"ms_per_image": round(dtg * 1e3, 4), "MPixels/s": round(H * W / dtg / 1e6, 1), "streams_equal_eager": bool(same),
                               "note": "whole image captured once, shape groups on parallel streams"}
    except Exception as e:      # an extra data point: never fail the bench line
        res["graph_replay"] = {"error": str(e)[:200]}
    # the same image as ONE launch chain: the four shape groups through one launch per kernel (launch groups, cgic_group_begin /
    # _launch; highres chain=True): entropy | VQ + router | compress | decode | merge = 5 launches for the six tiles
    try:
        import control_gic_amd as cg

        def once_chain(check=True):
            t = highres.compress_tiled(x, encode, codec, chain=True)
            p, st = highres.decompress_tiled(t, codec, check=check, chain=True, decoder="latency")      # (one image at a time: the GPU is this call's)
            return t, p, st
        tc, pc, _ = once_chain(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            tc, pc, _ = once_chain()
        torch.cuda.synchronize()
        dtc = (time.perf_counter() - t0) / iters
        same_c = tc.streams() == tiled.streams() and all(torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) for a, b in zip(pc, per_tile))
        # the same chain as ONE foreign call per image (cgic_compress_tiled, highres.TiledCall): buffers allocated once
        tcall = highres.TiledCall(vq, 0.1, 0.8, 1, H, W, frequency=codec.huffman, decoder="latency")
        zl = []
        for (th, tw), idxs in tcall.groups:
            encode(torch.empty((len(idxs), 3, th, tw), device=dev).uniform_())      # (makes the group's latent like `encode` does)
            zl.append(zs[(len(idxs), th, tw)])
        for _ in range(3):
            t1 = tcall(x, zl)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * iters):
            t1 = tcall(x, zl)
        torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / (4 * iters)
        same_1 = t1.streams() == tiled.streams() and all(int(d[3].abs().max()) == 0 for d in tcall.decoded)
        res["chain"] = {"eager_ms_per_image": round(dt1 * 1e3, 4), "eager_MPixels/s": round(H * W / dt1 / 1e6, 1),
                        "eager_python_chain_ms_per_image": round(dtc * 1e3, 4), "streams_equal_eager": bool(same_c and same_1),
                        "note": "all shape groups through ONE launch per kernel (cgic_group_*): 4 launches -- the tiles are cut inside the entropy-map "
                                "launch (cgic_entropy_maps_tiles), decoder + merge are one launch.  eager_ms_per_image: one cgic_compress_tiled call per "
                                "image (highres.TiledCall, buffers allocated once); eager_python_chain: the same chain recorded call by call from Python "
                                "(round 4's eager figure)"}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        gc_, (tgc, pgc, stgc) = cg.capture_graph(lambda: once_chain(False), side)
        torch.cuda.current_stream().wait_stream(side)
        for _ in range(3):
            gc_.replay()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4 * iters):
            gc_.replay()
        torch.cuda.synchronize()
        dtgc = (time.perf_counter() - t0) / (4 * iters)
        res["chain"].update({"graph_replay_ms_per_image": round(dtgc * 1e3, 4), "graph_replay_MPixels/s": round(H * W / dtgc / 1e6, 1),
                             "graph_streams_equal_eager": bool(int(stgc.abs().max()) == 0 and tgc.streams() == tiled.streams())})
    except Exception as e:
        res["chain"] = {"error": str(e)[:300]}
    # four such images in flight: one hipGraph per image (its shape groups one after the other) on four independent HIP streams,
    # the small-footprint decoder -- the tiling driver under pipeline.LaneStream's schedule
    try:
        from control_gic_amd.pipeline import GraphLanes

        def once_lane():
            t = highres.compress_tiled(x, encode, codec)
            p, st = highres.decompress_tiled(t, codec, check=False)
            return t, p, st
        gl = GraphLanes(dev, [once_lane] * 4)
        gl.replay(2); gl.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl.replay(2 * iters); gl.join(); torch.cuda.synchronize()
        dt4 = (time.perf_counter() - t0) / (2 * iters * 4)
        same4 = all(int(o[2].abs().max()) == 0 and o[0].streams() == tiled.streams() for o in gl.results)
        res["four_in_flight"] = {"ms_per_image": round(dt4 * 1e3, 4), "MPixels/s": round(H * W / dt4 / 1e6, 1), "streams_equal_eager": bool(same4),
                                 "note": "four images on four independent HIP streams (pipeline.GraphLanes), one hipGraph per image, throughput decoder"}

        def once_lane_chain():
            t = highres.compress_tiled(x, encode, codec, chain=True)
            p, st = highres.decompress_tiled(t, codec, check=False, chain=True)
            return t, p, st
        glc = GraphLanes(dev, [once_lane_chain] * 4)
        glc.replay(2); glc.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        glc.replay(2 * iters); glc.join(); torch.cuda.synchronize()
        dt4c = (time.perf_counter() - t0) / (2 * iters * 4)
        same4c = all(int(o[2].abs().max()) == 0 and o[0].streams() == tiled.streams() for o in glc.results)
        res["four_in_flight"].update({"chain_ms_per_image": round(dt4c * 1e3, 4), "chain_MPixels/s": round(H * W / dt4c / 1e6, 1),
                                      "chain_streams_equal_eager": bool(same4c)})
    except Exception as e:
        res["four_in_flight"] = {"error": str(e)[:200]}
    # eight images of that size at once (highres.compress_tiled_batch: the equal-shape tiles of all the images are one batch per
    # shape group -- 16 + 8 + 16 + 8 tiles instead of 2 + 1 + 2 + 1), one at a time and four such batches in flight
    try:
        from control_gic_amd.pipeline import GraphLanes
        N = 8
        xs = torch.from_numpy(np.random.default_rng(5).random((N, 3, H, W), dtype=np.float32)).to(dev)

        def once_batch(concurrent, chain=False):
            def fn():
                ts = highres.compress_tiled_batch(xs, encode, codec, concurrent=concurrent, chain=chain)
                p, st = highres.decompress_tiled_batch(ts, codec, concurrent=concurrent, check=False, chain=chain)
                return ts, p, st
            return fn
        ts, p, st = once_batch(False)(); torch.cuda.synchronize()
        okb = int(st.abs().max()) == 0
        for t, pt in zip(ts, p):
            for idxs, _, (ind0, masks0, _) in t.groups:
                fine = masks0[2].reshape(len(idxs), -1).bool()
                got = torch.cat([pt[i][0].reshape(1, -1) for i in idxs])
                okb = okb and bool(torch.equal(got[fine], ind0.reshape(len(idxs), -1)[fine]))
        gl1 = GraphLanes(dev, [once_batch(True)], decoder="latency")
        gl1.replay(2); gl1.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl1.replay(2 * iters); gl1.join(); torch.cuda.synchronize()
        dt1 = (time.perf_counter() - t0) / (2 * iters * N)
        gl = GraphLanes(dev, [once_batch(False)] * 4)
        gl.replay(2); gl.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl.replay(2 * iters); gl.join(); torch.cuda.synchronize()
        dt4 = (time.perf_counter() - t0) / (2 * iters * 4 * N)
        # the same as ONE launch chain per batch (launch groups): all four shape groups through one launch per kernel
        chain_res = {}
        try:
            tsc, _, stc = once_batch(False, True)(); torch.cuda.synchronize()
            same_c = int(stc.abs().max()) == 0 and all(a.streams() == b.streams() for a, b in zip(tsc, ts))
            glc1 = GraphLanes(dev, [once_batch(False, True)], decoder="latency")
            glc1.replay(2); glc1.join(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            glc1.replay(2 * iters); glc1.join(); torch.cuda.synchronize()
            dtc1 = (time.perf_counter() - t0) / (2 * iters * N)
            glc = GraphLanes(dev, [once_batch(False, True)] * 4)
            glc.replay(2); glc.join(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            glc.replay(2 * iters); glc.join(); torch.cuda.synchronize()
            dtc4 = (time.perf_counter() - t0) / (2 * iters * 4 * N)
            chain_res = {"chain_ms_per_image": round(dtc1 * 1e3, 4), "chain_MPixels/s": round(H * W / dtc1 / 1e6, 1),
                         "chain_four_in_flight_ms_per_image": round(dtc4 * 1e3, 4), "chain_four_in_flight_MPixels/s": round(H * W / dtc4 / 1e6, 1),
                         "chain_streams_equal": bool(same_c)}
        except Exception as e:
            chain_res = {"chain_error": str(e)[:200]}
        res["batch_of_8"] = {"ms_per_image": round(dt1 * 1e3, 4), "MPixels/s": round(H * W / dt1 / 1e6, 1),
                             "four_in_flight_ms_per_image": round(dt4 * 1e3, 4), "four_in_flight_MPixels/s": round(H * W / dt4 / 1e6, 1),
                             **chain_res,
                             "round_trip_ok": bool(okb), "bpp_mean": round(float(np.mean([t.bpp() for t in ts])), 6),
                             "note": "highres.compress_tiled_batch + decompress_tiled_batch on 8 images of one size: one batch of tiles per shape group; "
                                     "one hipGraph per batch (groups on parallel streams), then four batches on four HIP streams"}
    except Exception as e:
        res["batch_of_8"] = {"error": str(e)[:200]}
    # the same eight images arriving as uint8 frames [N,H,W,3]: tiles cut as bytes, entropy_maps_u8 = ToTensor + maps in one pass.
    # NOT like for like with the fp32 figures above: the input is a quarter of the bytes and the fp32 tiles are produced on the way
    try:
        from control_gic_amd.pipeline import GraphLanes
        import control_gic_amd as cg
        frames = (xs.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous()

        def encode_u8(tiles):
            T, th, tw, _ = tiles.shape
            key = (T, th, tw)
            if key not in zs:
                zs[key] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
            _, e8, e16 = cg.entropy_maps_u8(tiles)
            _, _, ind, mask, _, mode = vq_forward_route(zs[key], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
            return ind, mask, mode

        def once_u8():
            ts = highres.compress_tiled_batch(frames, encode_u8, codec)
            p, st = highres.decompress_tiled_batch(ts, codec, check=False)
            return ts, p, st
        ts8, _, st8 = once_u8(); torch.cuda.synchronize()
        gl = GraphLanes(dev, [once_u8] * 4)
        gl.replay(2); gl.join(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        gl.replay(2 * iters); gl.join(); torch.cuda.synchronize()
        dt4 = (time.perf_counter() - t0) / (2 * iters * 4 * 8)
        # the same frames through the launch chain with ToTensor + pad + crop + maps as ONE pass (frames_fp32: `encode` gets fp32 tiles)
        extra_u8 = {}
        try:
            def once_u8c():
                ts = highres.compress_tiled_batch(frames, encode, codec, chain=True, frames_fp32=True)
                p, st = highres.decompress_tiled_batch(ts, codec, check=False, chain=True)
                return ts, p, st
            tsc8, _, stc8 = once_u8c(); torch.cuda.synchronize()
            same8 = int(stc8.abs().max()) == 0 and all(a.streams() == b.streams() for a, b in zip(tsc8, ts8))
            glc8 = GraphLanes(dev, [once_u8c] * 4)
            glc8.replay(2); glc8.join(); torch.cuda.synchronize()
            t0 = time.perf_counter()
            glc8.replay(2 * iters); glc8.join(); torch.cuda.synchronize()
            dtc8 = (time.perf_counter() - t0) / (2 * iters * 4 * 8)
            extra_u8 = {"chain_four_in_flight_ms_per_image": round(dtc8 * 1e3, 4), "chain_four_in_flight_MPixels/s": round(H * W / dtc8 / 1e6, 1),
                        "chain_streams_equal": bool(same8)}
        except Exception as e:
            extra_u8 = {"chain_error": str(e)[:200]}
        res["batch_of_8_uint8_frames"] = {**extra_u8, "four_in_flight_ms_per_image": round(dt4 * 1e3, 4), "four_in_flight_MPixels/s": round(H * W / dt4 / 1e6, 1),
                                          "status_ok": int(st8.abs().max()) == 0, "bpp_mean": round(float(np.mean([t.bpp() for t in ts8])), 6),
                                          "note": "input = uint8 [8,H,W,3] frames = round(255 x) of batch_of_8's images (so its bpp differs slightly: other pixels, "
                                                  "not a parity gap -- equality with frames / 255 as fp32 input is what tests/test_highres_container.py checks); "
                                                  "round 3 measured it NO faster than the fp32 input (byte-wise torch copies per tile); "
                                                  "with cgic_cut_tiles (all tiles in one launch, 12-byte units) the frames are ahead"}
    except Exception as e:
        res["batch_of_8_uint8_frames"] = {"error": str(e)[:200]}
    return res


def tiles_768(dev, cb, vq, codec, B, steps):
    x, z, _ = make_inputs(B, 768, 768, seed=77)
    hp = HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec)
    g = hp.capture()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        g.replay()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ok, _ = check_against_oracle(hp.out, x, z, cb, (0.1, 0.8), images=[0, B - 1])
    dt4, ls = lanes_rate(vq, codec, (0.1, 0.8), [(hp.x, hp.z)], 4, 4 * steps, copies=4)
    ok4, _ = check_against_oracle(slot_out(ls, 3), x, z, cb, (0.1, 0.8), images=[0, B - 1])
    return {"workload": f"{B} tiles of 768x768 (the tile size of the 2K path), ratio (0.1,0.8,0.1), encode+decode, hipGraph replay",
            "value": round(steps * B * 768 * 768 / dt / 1e6, 2), "unit": "MPixels/s", "ms_per_step": round(dt / steps * 1e3, 5),
            "note": "value = one batch of tiles after the other on one stream; 4_in_flight = four such batches on independent streams",
            "MPixels/s_4_in_flight": round(B * 768 * 768 / dt4 / 1e6, 2), "bpp_match": bool(ok and ok4)}


def b1_latency(dev, cb, vq, codec):
    """one 256x256 image (the only batch size the reference's compress() accepts): encode+decode latency"""
    x, z, _ = make_inputs(1, 256, 256, seed=5)
    hp = HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec)
    g = hp.capture()
    for _ in range(5):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 200
    for _ in range(n):
        g.replay()
    torch.cuda.synchronize()
    t_graph = (time.perf_counter() - t0) / n
    t0 = time.perf_counter()
    for _ in range(50):
        hp.step()
    torch.cuda.synchronize()
    t_eager = (time.perf_counter() - t0) / 50
    ok, _ = check_against_oracle(hp.out, x, z, cb, (0.1, 0.8))
    # the reference's calling pattern (one compress() per image, inference.py:157-166) through the C-level driver: ONE foreign
    # call per image (cgic_compress_image, pipeline.HotCall), buffers allocated once
    import control_gic_amd as cg
    hc = cg.pipeline.HotCall(vq, 0.1, 0.8, 1, 256, 256, frequency=codec.huffman)
    for _ in range(5):
        o = hc(hp.x, hp.z)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200):
        o = hc(hp.x, hp.z)
    torch.cuda.synchronize()
    t_one = (time.perf_counter() - t0) / 200
    ok1, _ = check_against_oracle((o["e8"], o["e16"], o["mask"], o["mode"], o["z_q"], o["ind"], o["comp"], *o["dec"]), x, z, cb, (0.1, 0.8))
    return {"graph_replay_us": round(t_graph * 1e6, 2), "eager_us": round(t_one * 1e6, 2), "eager_four_calls_us": round(t_eager * 1e6, 2),
            "bpp_match": bool(ok and ok1),
            "note": "B=1, 256x256, encode+decode, back-to-back (throughput of B=1 calls).  eager_us: one cgic_compress_image call per image "
                    "(pipeline.HotCall: one ctypes call, buffers allocated once); eager_four_calls_us: the four entry points from Python with "
                    "per-call allocation (round 4's eager_us)"}


def end_to_end_estimate(dev, hot_ms_per_batch, B, H, W):
    """SURVEY.md section 8(d) asks for the hot-path number next to the end-to-end one.  The stock conv encoder / decoder are out
    of scope and the reference cannot travel to the GPU box, so this is an ESTIMATE, labelled as such: the reference's
    measured FLOP counts per 256x256 image (encode 276.4 GFLOP, decode 824.2 GFLOP, fp32 -- the 1e-5 pixel
    tolerance excludes 16-bit) divided by the fp32 3x3-convolution throughput torch/MIOpen reaches on this GPU at the
    reference's channel widths and resolutions (vqvae_blocks.py:303-374, decoder.py:340-398: 128..512 channels, 256^2..16^2)."""
    import torch.nn.functional as F
    shapes = [(8, 128, 256, 256), (8, 256, 128, 128), (16, 512, 64, 64), (32, 512, 32, 32)]
    tf = []
    for n, c, hh, ww in shapes:
        xx = torch.randn(n, c, hh, ww, device=dev)
        wt = torch.randn(c, c, 3, 3, device=dev) * 0.01
        for _ in range(3):
            F.conv2d(xx, wt, padding=1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        it = 8
        for _ in range(it):
            F.conv2d(xx, wt, padding=1)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / it
        tf.append(2.0 * n * c * c * 9 * hh * ww / dt / 1e12)
        del xx, wt
    conv_tf = float(np.mean(tf))
    scale = (H * W) / 65536.0                                     # FLOPs scale with the pixel count
    enc_ms = 276.4e9 * scale / (conv_tf * 1e12) * 1e3 * B
    dec_ms = 824.2e9 * scale / (conv_tf * 1e12) * 1e3 * B
    mp = B * H * W / 1e6
    return {"kind": "estimate", "conv_fp32_TFLOPs_measured": [round(v, 1) for v in tf], "conv_fp32_TFLOPs_mean": round(conv_tf, 1),
            "encode_MPixels/s": round(mp / ((enc_ms + hot_ms_per_batch * 0.6) * 1e-3), 3),
            "encode+decode_MPixels/s": round(mp / ((enc_ms + dec_ms + hot_ms_per_batch) * 1e-3), 3),
            "hot_path_share_of_encode+decode": round(hot_ms_per_batch / (enc_ms + dec_ms + hot_ms_per_batch), 6),
            "cpu_reference_encode+decode_MPixels/s": 0.030, "cpu_reference_encode_MPixels/s": 0.085,
            "note": "stock conv encoder + decoder = 276.4 + 824.2 GFLOP per 256x256 image (SURVEY.md 3.1, torch FlopCounterMode on the "
                    "reference) at the mean measured fp32 3x3-conv rate of this GPU; the hot path (this repository) is the remainder. "
                    "CPU figures: the real reference on 8 Xeon cores in the survey container (SURVEY.md 8d), not re-measured here."}


# ------------------------------------------------------------------------------------------------ one rank
class StubStream:
    """CPU stand-in for BatchStream (tests of the launcher / reduction logic only; never used on a GPU box)"""

    def __init__(self, B, h, w):
        self.hist = torch.zeros(1024, dtype=torch.int64)
        self.n = B * h * w

    def submit(self, n=1):
        for _ in range(n):
            self.hist[7] += self.n

    def join(self):
        pass


MIXED_KODAK, MIXED_DIV2K = (512, 768), (1356, 2040)      # (H, W): BASELINE config 5


def mixed_sizes(n_div2k):
    """the stream of BASELINE config 5: 24 Kodak-sized images and n DIV2K-sized ones"""
    return [MIXED_KODAK] * 24 + [MIXED_DIV2K] * n_div2k


def mixed_share(sizes, rank, world):
    """(Kodak images, DIV2K images, pixels, latent vectors) of a rank: round-robin per size class -- the Kodak-sized images are
    dealt from rank 0 upwards, the DIV2K-sized ones (5.9 times the pixels each) from the last rank downwards, so that a remainder
    of one class lands where the other class left room"""
    nk_all, nd_all = sizes.count(MIXED_KODAK), sizes.count(MIXED_DIV2K)
    nk = len(range(rank, nk_all, world))
    nd = len(range(world - 1 - rank, nd_all, world))
    vec = lambda H, W: (-(-H // 16) * 4) * (-(-W // 16) * 4)                 # latent vectors of an image (DIV2K: padded to x16)
    return nk, nd, nk * MIXED_KODAK[0] * MIXED_KODAK[1] + nd * MIXED_DIV2K[0] * MIXED_DIV2K[1], nk * vec(*MIXED_KODAK) + nd * vec(*MIXED_DIV2K)


class StubMixed:
    """CPU stand-in for MixedStream (launcher / reduction tests only)"""

    def __init__(self, vectors):
        self.hist = torch.zeros(1024, dtype=torch.int64)
        self.vectors = vectors

    def submit(self, n=1, collective=None):
        for _ in range(n):
            self.hist[3] += self.vectors
        self.work = collective(self.hist) if collective is not None else None

    def join(self):
        if getattr(self, "work", None) is not None:
            self.work.wait()


class MixedStream:
    """One rank's share of BASELINE config 5 (Kodak 768x512 + DIV2K 2040x1356 images, round-robin over the ranks) as two
    independent chains on two hardware queues: the Kodak-sized images as ONE batch through the hot path, the DIV2K-sized ones
    through the tiling driver with the equal-shape tiles of all of them batched (highres.compress_tiled_batch).  Each chain is
    an encode hipGraph and a decode hipGraph; on the LAST step of a submit the histogram all-reduce -- the path's only
    collective -- is issued as soon as both encode graphs are enqueued and runs under the decode side."""

    def __init__(self, dev, rank, nk, nd, vq, codec, ratio, hist):
        import control_gic_amd as cg
        from control_gic_amd import highres
        from control_gic_amd.quantize import vq_forward_route
        self.dev, self.hist = dev, hist
        rng = np.random.default_rng(500 + rank)
        st = cg.pipeline.distinct_queue_streams(dev, 2)
        self.chains = []
        lib = cg._lib
        if nk:
            H, W = MIXED_KODAK
            xk = torch.from_numpy(rng.random((nk, 3, H, W), dtype=np.float32)).to(dev)
            zk = torch.from_numpy(rng.standard_normal((nk, 4, H // 4, W // 4), dtype=np.float32)).to(dev)
            pipe = cg.pipeline.HotPathPipeline(vq, ratio[0], ratio[1], frequency=codec.huffman, prepare=True)
            box = {}

            def k_enc():
                e8, e16 = cg.entropy_maps(xk)
                zq, loss, ind, mask, _, mode = vq_forward_route(zk, vq.embedding.weight, vq.beta, vq.legacy, e16, e8, ratio[0], ratio[1],
                                                              per_image=True, prepared=pipe.prepared, pixels=xk)
                box["comp"] = codec.compress(ind, mask, mode, hist=hist)
                return box["comp"]

            def k_dec():
                box["dec"] = codec.decompress(box["comp"], decoder="throughput")
                return box["dec"]
            self.chains.append((st[0], k_enc, k_dec, box))
        if nd:
            H, W = MIXED_DIV2K
            xd = torch.from_numpy(rng.random((nd, 3, H, W), dtype=np.float32)).to(dev)
            zs = {}
            box2 = {}

            def encode(tiles):
                T, _, th, tw = tiles.shape
                key = (T, th, tw)
                if key not in zs:
                    zs[key] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
                e8, e16 = cg.entropy_maps(tiles)
                _, _, ind, mask, _, mode = vq_forward_route(zs[key], vq.embedding.weight, 0.25, True, e16, e8, ratio[0], ratio[1], per_image=True,
                                                            pixels=tiles)
                lib.call("cgic_index_histogram", lib.ptr(ind), ind.numel(), 1024, lib.ptr(hist), lib.current_stream(dev))
                return ind, mask, mode

            # (chain: the four shape groups of the tiled images through one launch per kernel -- launch groups)
            def d_enc():
                box2["tiled"] = highres.compress_tiled_batch(xd, encode, codec, chain=True)
                return box2["tiled"]

            def d_dec():
                with cg.decoder_mode("throughput"):
                    box2["dec"] = highres.decompress_tiled_batch(box2["tiled"], codec, check=False, chain=True)
                return box2["dec"]
            self.chains.append((st[1], d_enc, d_dec, box2))
        self.graphs = []
        cur = torch.cuda.current_stream(dev)
        for stream, enc, dec, _ in self.chains:
            stream.wait_stream(cur)
            with torch.cuda.stream(stream):
                for _ in range(2):
                    enc(); dec()
            torch.cuda.synchronize(dev)
            ge, _ = cg.capture_graph(enc, stream)
            gd, _ = cg.capture_graph(dec, stream)
            self.graphs.append((stream, ge, gd))
        torch.cuda.synchronize(dev)
        self.work = None

    def submit(self, n=1, collective=None):
        cur = torch.cuda.current_stream(self.dev)
        for k in range(n):
            last = k == n - 1 and collective is not None
            for stream, ge, gd in self.graphs:
                with torch.cuda.stream(stream):
                    ge.replay()
                    if not last:
                        gd.replay()
            if last:
                for stream, _, _ in self.graphs:
                    cur.wait_stream(stream)
                self.work = collective(self.hist)          # async: RCCL's stream, ordered after the encode sides
                for stream, _, gd in self.graphs:
                    with torch.cuda.stream(stream):
                        gd.replay()

    def join(self):
        cur = torch.cuda.current_stream(self.dev)
        for stream, _, _ in self.graphs:
            cur.wait_stream(stream)
        if self.work is not None:
            self.work.wait()
            self.work = None

    def round_trip_ok(self):
        """decoded indices == encoded ones wherever the fine grain kept them, statuses zero (after a join + synchronize)"""
        ok = True
        for _, _, _, box in self.chains:
            if "comp" in box:
                ok = ok and int(box["dec"][3].abs().max()) == 0
            else:
                per, status = box["dec"]
                ok = ok and int(status.abs().max()) == 0
        return bool(ok)


def mixed_extra(dev, vq, codec, ratio, n_div2k=8, steps=10):
    """BASELINE config 5 on this one GPU (`python bench.py --workload mixed [--gpus N]` is the same as a bench line of its own)"""
    sizes = mixed_sizes(n_div2k)
    nk, nd, pix, vec = mixed_share(sizes, 0, 1)
    hist = torch.zeros(1024, dtype=torch.int64, device=dev)
    ms = MixedStream(dev, 0, nk, nd, vq, codec, ratio, hist)
    ms.submit(2); ms.join(); torch.cuda.synchronize()
    hist.zero_()
    t0 = time.perf_counter()
    ms.submit(steps); ms.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = ms.round_trip_ok() and int(hist.sum()) == steps * vec
    return {"workload": f"{nk} images of 768x512 (one batch) + {nd} of 2040x1356 (per-shape tile batches), two independent queues, encode + decode hipGraphs",
            "MPixels/s": round(pix / dt / 1e6, 1), "ms_per_pass": round(dt * 1e3, 4), "round_trip_and_histogram_ok": bool(ok),
            "note": "python bench.py --workload mixed --gpus N shards the same stream over N ranks (strong scaling) with the histogram all-reduce under the last decode"}


def teardown_watchdog(seconds=30.0):
    """The results are out; what is left is communicator teardown (a closing barrier + destroy_process_group), and an RCCL teardown
    that never returns -- seen once in a spawned one-rank child at the end of a run whose line and extras file were already written --
    would hang the caller (a test harness waiting for the pipes, the driver's next N).  A daemon timer ends the process with status
    0 if the teardown has not finished in time; a normal exit cancels nothing and loses nothing."""
    import threading
    t = threading.Timer(seconds, lambda: os._exit(0))
    t.daemon = True
    t.start()
    return t


def device_identity(dev):
    """(PCI bus id of this rank's GPU as HIP reports it, RCCL version) -- what a reader of an N-GPU line needs to see that the ranks
    sat on N different devices of one node and which collective library carried the histogram"""
    bus = None
    try:
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        buf = ctypes.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(dev.index or 0)) == 0:
            bus = buf.value.decode()
    except Exception:                                        # noqa: BLE001
        bus = None
    ver = None
    try:
        v = torch.cuda.nccl.version()
        ver = ".".join(str(q) for q in v) if isinstance(v, tuple) else str(v)
    except Exception:                                        # noqa: BLE001
        ver = None
    return bus, ver


def run_rank(a, rank, world, local):
    stub = bool(a.stub)
    dist = None
    if stub:
        dev = torch.device("cpu")
    else:
        if torch.cuda.device_count() <= local:
            raise SystemExit(f"bench.py: rank {rank} needs GPU {local} but only {torch.cuda.device_count()} are visible")
        torch.cuda.set_device(local)
        dev = torch.device("cuda", local)
    ranks_seen = None
    dist_note = None
    # N = 1 goes through the same code as N > 1: a one-rank communicator (RCCL on the GPU, gloo for the CPU stub), so that
    # `rccl_ranks` is read back from a communicator and the histogram all-reduce is the same call at every N
    if world > 1 or "RANK" in os.environ or not a.no_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            os.environ["MASTER_PORT"] = str(free_port()) if world == 1 else "29500"
        try:
            with stdout_to_stderr():
                if stub:
                    dist.init_process_group("gloo", rank=rank, world_size=world)
                else:
                    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)  # "nccl" is RCCL on ROCm
                one = torch.ones(1, dtype=torch.int64, device=dev)
                dist.all_reduce(one)                            # creates the communicator (and its banner) now
                ranks_seen = int(one.item())                    # read back from the communicator: how many ranks really joined
        except Exception as e:                                  # noqa: BLE001
            if world > 1:
                raise
            dist_note = f"one-rank communicator could not be created ({str(e)[:120]}): ran without torch.distributed"
            log("bench.py: " + dist_note)
            dist = None
        if dist is not None and ranks_seen != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but the communicator has {ranks_seen} ranks")

    def sync():
        if not stub:
            torch.cuda.synchronize()

    def barrier():
        sync()
        if dist is not None and world > 1:                  # (one rank: nothing to wait for -- a no-op collective would only add its launch)
            dist.barrier()
        sync()

    B, H, W = a.batch, a.size, a.size
    ratio = (0.1, 0.8)
    h, w = H // 4, W // 4
    mixed = a.workload == "mixed"
    if mixed:
        sizes = mixed_sizes(a.div2k)
        nk, nd, pix_rank, vec_rank = mixed_share(sizes, rank, world)
        pix_all = sum(hh * ww for hh, ww in sizes)
        vec_all = sum(mixed_share(sizes, r, world)[3] for r in range(world))
        n_slots, slots_np = 1, []
        if stub:
            stream = StubMixed(vec_rank)
            hist = stream.hist
        else:
            import control_gic_amd as cg
            cb = make_inputs(1, 16, 16, 0)[2]
            vq = make_quantizer(dev, cb)
            codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
            hist = torch.zeros(1024, dtype=torch.int64, device=dev)
            stream = MixedStream(dev, rank, nk, nd, vq, codec, ratio, hist)
            stream.submit(a.warmup)
            stream.join()
            sync()
            hist.zero_()
    elif stub:
        stream, slots_np, n_slots = StubStream(B, h, w), [], 2
        hist = stream.hist
    else:
        import control_gic_amd as cg
        per_slot = B * H * W * 13                            # image 12 B/pixel + latent 1 B/pixel
        n_slots = a.slots if a.slots > 0 else max(2, -(-int(IC_BYTES * 1.15) // per_slot))
        n_slots = max(2, min(n_slots, 64))
        if a.slots <= 0 and a.schedule == "sequential" and a.lanes > 1:
            n_slots = a.lanes * -(-n_slots // a.lanes)        # every lane the same number of batches
        slots_np = [make_inputs(B, H, W, seed=1000 + 97 * rank + s) for s in range(n_slots)]   # each rank owns its own images
        cb = slots_np[0][2]
        vq = make_quantizer(dev, cb)
        codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
        hist = torch.zeros(1024, dtype=torch.int64, device=dev)
        slots_dev = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
        if a.schedule == "pipelined" and not a.no_graph:
            stream = cg.experimental.BatchStream(vq, ratio[0], ratio[1], slots_dev, frequency=codec.huffman, hist=hist)
            stream.capture()
        else:
            # control_gic_amd.pipeline.LaneStream: batch t on HIP stream t % lanes, one ring graph per stream, no
            # dependency between the streams (lanes=1: one batch in flight, the round-1/2a configuration)
            stream = cg.pipeline.LaneStream(vq, ratio[0], ratio[1], slots_dev, lanes=a.lanes, frequency=codec.huffman, hist=hist,
                                            graph=not a.no_graph, ring=not a.no_ring, fuse_router=not a.split_router, max_ring=a.max_ring,
                                            )
            stream.capture()
            # every graph of the W warm-up steps AND of the K timed steps is captured before the warm-up runs: nothing but the
            # barrier lies between the last warm-up step and the first timed one (a capture in between left the GPU idle for
            # milliseconds and the timed window started from its idle clocks)
            stream.prepare(a.warmup)
            stream.prepare(a.steps, after=a.warmup)
            sync()
        stream.submit(a.warmup)
        stream.join()
        hist.zero_()
        sync()

    # ---- the timed region: barrier + synchronize in front; exactly K steps; at N > 1 the path's ONE collective -- the int64[1024]
    # histogram all-reduce, which no rank leaves before every rank has arrived: it IS the closing barrier (a dist.barrier() behind
    # it would put a second collective and its skew into a window of a few hundred microseconds) -- then synchronize.  Every rank
    # also times its own K steps with device events (no collective inside); the job's time is the MAX over ranks of the wall time.
    class _CountCollectives:
        """every torch.distributed collective issued while active (the timed region) is counted: what the line reports as
        `timed_collectives` is what was really called, not what the code above intends"""
        NAMES = ("all_reduce", "barrier", "all_gather", "broadcast", "reduce", "all_to_all", "reduce_scatter", "all_gather_into_tensor")

        def __init__(self):
            self.n = 0
            self.saved = {}

        def __enter__(self):
            if dist is not None:
                for name in self.NAMES:
                    fn = getattr(dist, name, None)
                    if fn is not None:
                        self.saved[name] = fn
                        setattr(dist, name, self._wrap(fn))
            return self

        def _wrap(self, fn):
            def counted(*args, **kw):
                self.n += 1
                return fn(*args, **kw)
            return counted

        def __exit__(self, *exc):
            for name, fn in self.saved.items():
                setattr(dist, name, fn)

    counter = _CountCollectives()
    ev = None if stub else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
    barrier()
    with counter:
        t0 = time.perf_counter()
        if ev:
            ev[0].record()
        if mixed:
            # the path's only exchange goes out as soon as the last pass's encode side is enqueued and runs under its decode side
            # (async; join() waits for it: no rank leaves the region before every rank's histogram has arrived)
            coll = (lambda hh: dist.all_reduce(hh, op=dist.ReduceOp.SUM, async_op=True)) if dist is not None and world > 1 else None
            stream.submit(a.steps, collective=coll)
            stream.join()
        else:
            stream.submit(a.steps)                              # exactly K steps
            stream.join()
        if ev:
            ev[1].record()
        if not mixed and dist is not None and world > 1:
            # the path's only exchange: global usage histogram (int64, exact) -- once per stream of batches
            dist.all_reduce(hist, op=dist.ReduceOp.SUM)
        sync()
        dt = time.perf_counter() - t0
    timed_collectives = counter.n
    steps_ms = ev[0].elapsed_time(ev[1]) if ev else dt * 1e3      # this rank's K steps by device events
    if dist is not None and world == 1:
        dist.all_reduce(hist, op=dist.ReduceOp.SUM)        # one rank: the identity, issued all the same (outside the timed region)
        sync()
    per_rank = [dt]
    per_rank_dev = [steps_ms * 1e-3]
    allreduce_us = None
    bus_id, rccl_version = (None, None) if stub else device_identity(dev)
    per_rank_bus = [bus_id]
    if dist is not None:
        try:
            every_bus = [None] * world
            dist.all_gather_object(every_bus, bus_id)
            per_rank_bus = every_bus
        except Exception:                                    # noqa: BLE001 -- identity is a report field, never a reason to fail
            per_rank_bus = [bus_id]
        mine = torch.tensor([dt, steps_ms * 1e-3], dtype=torch.float64, device=dev)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [float(v[0].item()) for v in every]
        per_rank_dev = [float(v[1].item()) for v in every]
        dt = max(per_rank)                                  # the job is as slow as its slowest rank
        # the path's only collective by itself (outside the timed region, communicator warm): int64[1024] SUM
        probe = torch.zeros(1024, dtype=torch.int64, device=dev)
        best = 1e9
        for _ in range(5):
            barrier()
            ta = time.perf_counter()
            dist.all_reduce(probe, op=dist.ReduceOp.SUM)
            sync()
            best = min(best, time.perf_counter() - ta)
        allreduce_us = round(best * 1e6, 1)
    hist_total = int(hist.sum().item())
    want_total = a.steps * vec_all if mixed else world * a.steps * B * h * w
    if hist_total != want_total:
        raise SystemExit(f"bench.py: usage histogram lost counts ({hist_total} != {want_total})")
    if mixed:
        if rank == 0:
            ok = True if stub else stream.round_trip_ok()
            pix = [mixed_share(sizes, r, world)[2] for r in range(world)]
            res = {
                "metric": "encode+decode MPixels/s at fixed granularity ratio; bpp match vs reference",
                "value": round(a.steps * pix_all / dt / 1e6, 2), "unit": "MPixels/s",
                "n_gpus": world, "rccl_ranks": ranks_seen, "steps": a.steps, "warmup": a.warmup,
                "rank_pci_bus_ids": per_rank_bus, "rccl_version": rccl_version, "spawned_ranks": bool(a.dry_nccl or (world > 1 and "TORCHELASTIC_RUN_ID" not in os.environ)),
                "per_rank_MPixels/s": [round(a.steps * p / t / 1e6, 2) for p, t in zip(pix, per_rank)],
                "per_rank_MPixels/s_device_events": [round(a.steps * p / t / 1e6, 2) for p, t in zip(pix, per_rank_dev)],
                "timed_collectives": timed_collectives,
                "histogram_allreduce_us": allreduce_us, "ms_per_step": round(dt / a.steps * 1e3, 5),
                "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
                "round_trip_ok": ok,
                "config": {"workload": f"mixed stream (BASELINE config 5): 24 images of 768x512 + {a.div2k} of 2040x1356 (tiled 768, 6 tiles each), "
                                       "sharded round-robin over the ranks; one step = one pass over the whole stream; hot path only, latent synthetic",
                           "launch": "stub" if stub else "per rank: the Kodak-sized images as one batch and the DIV2K-sized ones as per-shape tile batches, two "
                                     "independent hardware queues, an encode and a decode hipGraph each; the int64[1024] histogram all-reduce is issued "
                                     "async once the last pass's encode graphs are enqueued",
                           "sharding": f"images per rank: {[mixed_share(sizes, r, world)[:2] for r in range(world)]} (Kodak, DIV2K)"}}
            print(json.dumps(res), flush=True)
        if dist is not None:
            teardown_watchdog()
            barrier()
            dist.destroy_process_group()
        return

    if rank == 0:
        res = {
            "metric": "encode+decode MPixels/s at fixed granularity ratio; bpp match vs reference",
            "value": round(world * a.steps * B * H * W / dt / 1e6, 2),
            "unit": "MPixels/s",
            "n_gpus": world, "rccl_ranks": ranks_seen, "steps": a.steps, "warmup": a.warmup,
            "rank_pci_bus_ids": per_rank_bus, "rccl_version": rccl_version, "spawned_ranks": bool(a.dry_nccl or (world > 1 and "TORCHELASTIC_RUN_ID" not in os.environ)),
            "per_rank_MPixels/s": [round(a.steps * B * H * W / t / 1e6, 2) for t in per_rank],
            "per_rank_MPixels/s_device_events": [round(a.steps * B * H * W / t / 1e6, 2) for t in per_rank_dev],
            "timed_collectives": timed_collectives,
            "histogram_allreduce_us": allreduce_us,
            "ms_per_step": round(dt / a.steps * 1e3, 5),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"batch {B} of {H}x{W} per GPU per step, codebook 1024x4, ratio (0.1,0.8,0.1), "
                                   "hot path only (entropy maps + router + VQ + Huffman/mask coder, encode+decode); "
                                   "conv encoder/decoder out of scope, latent synthetic",
                       "launch": "stub" if stub else (
                           f"{a.schedule}: " + ("encode-side and decode-side hipGraphs of successive batches on two HIP streams"
                                                if a.schedule == "pipelined" and not a.no_graph else
                                                ("eager, one stream" if a.no_graph else "one hipGraph per batch, one stream" if a.no_ring else
                                                 f"{a.lanes} independent HIP stream(s), batch t on stream t % {a.lanes}, one hipGraph launch per run of up to "
                                                 f"{a.max_ring} consecutive batches of a stream (captured before the timed region)"))),
                       "inputs": f"{n_slots} distinct resident batches in rotation ({n_slots * B * H * W * 13 / 2**20:.0f} MiB > 256 MiB Infinity Cache)",
                       "sharding": "images round-robin over ranks; one RCCL all-reduce of the int64[1024] histogram per run"},
        }
        if dist_note:
            res["config"]["distributed"] = dist_note
        if not stub and not a.no_report:
            line, extras = report(a, dev, world, stream, slots_np, vq, codec, ratio, res["value"], res["ms_per_step"])
            res.update(line)
            try:
                path = a.extras_file or os.path.join(ROOT, "gpurun_out", "bench_extras.json")
                os.makedirs(os.path.dirname(path), exist_ok=True)
                with open(path, "w") as f:
                    json.dump(extras, f, indent=1)
                res["extras_file"] = os.path.relpath(path, ROOT)
            except OSError as e:
                res["extras_file"] = f"not written: {e}"
        print(json.dumps(res), flush=True)
    if dist is not None:
        sys.stdout.flush()
        teardown_watchdog()
        dist.barrier()
        dist.destroy_process_group()


def slot_out(stream, k):
    s = stream.slots[k]
    e = s.enc
    return (e["e8"], e["e16"], e["mask"], e["mode"], e["z_q"], e["ind"], e["comp"], *s.dec)


def report(a, dev, world, stream, slots_np, vq, codec, ratio, value, ms_per_step=None):
    """everything next to the headline value (outside the timed region) -> (what goes on the JSON line, the bulky rest).  The
    driver's record keeps the first ~2 KB of the line: roofline, cpu_baseline, the parity summary and the 8-bit-content figure
    sit there; every other extra goes to the side file (bench.py --extras-file, default gpurun_out/bench_extras.json)."""
    B, H, W = a.batch, a.size, a.size
    x, z, cb = slots_np[0]
    res = {}
    line = {}
    # parity of the timed work itself: ALL images of slot 0 and of the last slot, as left behind by the timed steps
    torch.cuda.synchronize()
    ok0, bpp = check_against_oracle(slot_out(stream, 0), x, z, cb, ratio)
    kl = len(slots_np) - 1
    okl, _ = check_against_oracle(slot_out(stream, kl), slots_np[kl][0], slots_np[kl][1], cb, ratio, images=range(0, B, 8))
    res["bpp"] = round(float(np.mean(bpp)), 6)
    res["bpp_match"] = bool(ok0 and okl)
    if not res["bpp_match"]:
        log("ERROR: bitstream / masks / indices differ from the oracle")

    hp = HotPath(dev, x, z, cb, ratio, vq=vq, codec=codec)
    g = hp.capture()
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g.replay()
    torch.cuda.synchronize()
    res["single_batch"] = {"ms_per_step": round((time.perf_counter() - t0) / 50 * 1e3, 5),
                           "note": "the SAME batch replayed back to back on one stream (round-1 headline configuration: inputs "
                                   "cache-resident, no overlap between batches) = latency of one batch through all five launches"}
    res["single_batch"]["MPixels/s"] = round(B * H * W / res["single_batch"]["ms_per_step"] / 1e3, 1)
    if a.schedule == "sequential" and a.lanes > 1 and not a.no_graph:
        dt1, _ = lanes_rate(vq, codec, ratio, [(sl.x, sl.z) for sl in stream.slots], 1, 20 * len(stream.slots))
        res["one_batch_in_flight"] = {"ms_per_step": round(dt1 * 1e3, 5), "MPixels/s": round(B * H * W / dt1 / 1e6, 1),
                                      "note": "the same rotation of distinct batches on ONE stream (lanes=1): every launch waits for the "
                                              "previous one, the round-1 / early round-2 configuration of `value`"}
    stages = stage_breakdown(hp)
    res["stages_us"] = stages
    # roofline of the dominant kernel: the launch the timed step really makes (VQ + router workgroups in one grid)
    N = B * (H // 4) * (W // 4)
    flops = 2.0 * N * 1024 * 4                        # 2*N*K*D per launch (0.512 kFLOP/pixel)
    t_live = stages["vq+router_fused_launch"] * 1e-6  # HIP events around 20 launches in a hipGraph, on the stream they run on
    prof = None
    pj = next((q for q in (os.path.join(ROOT, "profiles", f"r0{r}_roofline.json") for r in (6, 5, 4)) if os.path.exists(q)), "")
    if os.path.exists(pj) and (B, H) == (64, 256):
        try:
            prof = json.load(open(pj))
        except Exception:                             # noqa: BLE001
            prof = None
    # `frac` is THIS RUN'S own figure: HIP events around 100 launches of the dominant kernel (20 per hipGraph x 5 replays), on the
    # stream they run on.  `frac_rocprof` is the same command under rocprofv3 --kernel-trace --stats (tools/run_roofline_cmd.py,
    # profiles/rNN_roofline.json of the latest round, made by tools/gpu_profile.sh): the number a reader can recompute from
    # profiles/.  The two agree within the boxes' spread (~2 %).
    t_prof = prof["rocprof_avg_us_alone_graph"] * 1e-6 if prof else None
    achieved = flops / t_live / 1e12
    res["roofline"] = {
        "kernel": "vq_filter_router_kernel (VQ forward + the per-image router workgroups, the launch of the timed step)",
        "bound": "mfma", "achieved": round(achieved, 3), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
        "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
        "traffic": prof.get("hbm_bytes_per_launch") if prof else None,
        "duration_us": round(t_live * 1e6, 3),
        "duration_source": "live: HIP events on the launch stream, 20 launches per hipGraph x 5 replays (bench.graph_kernel_time)",
        "frac_rocprof": round(flops / t_prof / 1e12 / PEAK_F32_MFMA_TFLOPS, 4) if t_prof else None,
        "duration_rocprof_us": round(t_prof * 1e6, 3) if t_prof else None,
        "rocprof_source": (os.path.relpath(pj, ROOT) + ": rocprofv3 --kernel-trace average over the 100 timed launches of tools/run_roofline_cmd.py "
                           "(the same command as the live measurement); traffic and mfma_busy_frac are that profile's counter passes") if prof else None,
        "frac_one_lane_loop": prof.get("frac_lanes1_loop") if prof else None,
        "one_lane_loop_us": prof.get("rocprof_avg_us_lanes1_loop") if prof else None,
        "in_step_us": prof.get("rocprof_avg_us_lanes4_loop") if prof else None,
        "vq_alone_frac": round(flops / (stages["vq_kernel_alone"] * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
        "mfma_busy_frac": (prof.get("mfma") or {}).get("mfma_busy_frac") if prof else None,
        "mfma_counters": {k: v for k, v in (prof.get("mfma") or {}).items() if k.startswith("SQ_")} if prof else None,
        "note": "algorithmic flops = 2*N*K*D of the fp32 distance contraction per launch / the kernel's average duration, priced against the dense "
                "fp32 MFMA peak (results are bit-identical to the fp32 sequence); the kernel issues fp16 MFMAs with 16 K-slots per 4-dim contraction "
                f"({2.0 * N * 1024 * 16 / 1e9:.1f} GFLOP per launch = {2.0 * N * 1024 * 16 / t_live / 2.5e15:.2f} of the 2.5 PFLOP/s fp16 peak) "
                "and is bound by VALU + MFMA issue (one v_min3 per two scores), not by the matrix cores alone: mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x "
                "the launch's cycles), from the counter passes of the same command (profiles/rNN_pmc_sq_vq.md).  frac: the launch by itself, "
                "back to back, measured in this run; frac_one_lane_loop: the same kernel inside the one-batch-in-flight step (behind the entropy kernel's 50 MB: cold "
                "L2); in_step_us: its duration while the kernels of three other batches share the GPU (not a kernel property; under the "
                "profiler, which serialises part of the overlap); vq_alone_frac: the VQ kernel without the router workgroups, live"}
    # the STEP against its own floors: counted HBM bytes of its five launches (the round's FETCH_SIZE / WRITE_SIZE passes) at 8 TB/s and
    # 2NKD at the fp32 yardstick; next to them what each launch costs the GPU with four lanes in flight (measured here)
    if ms_per_step is not None and (B, H) == (64, 256):
        step_bytes = (prof or {}).get("step_hbm_bytes")
        floor_hbm = step_bytes / (PEAK_HBM_GBS * 1e9) * 1e6 if step_bytes else None
        floor_fp32 = flops / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6
        st = {"hbm_bytes": step_bytes, "hbm_bytes_by_kernel": (prof or {}).get("step_hbm_bytes_by_kernel"), "flops": flops,
              "floor_hbm_us": round(floor_hbm, 2) if floor_hbm else None, "floor_fp32_us": round(floor_fp32, 2),
              "ms_per_step": ms_per_step}
        st["step_frac"] = round(max(floor_hbm or 0.0, floor_fp32) / (ms_per_step * 1e3), 4)
        if not a.no_extra:
            try:
                st["resource_us"] = saturated_stage_times(hp)
                st["resource_us_sum"] = round(sum(st["resource_us"].values()), 2)
            except Exception as e:                               # an extra data point: never fail the bench line
                st["resource_us"] = {"error": str(e)[:200]}
        st["note"] = ("step = entropy maps -> VQ + router -> compress (+ histogram) -> decode -> merge of one batch; hbm_bytes = counted FETCH_SIZE x 2 + "
                      "WRITE_SIZE of the four-lane step's kernels (profiles/rNN_pmc_hbm.md); step_frac = max(floor_hbm, floor_fp32) / ms_per_step; "
                      "resource_us = each launch replayed alone by four streams at once (wall / launches): what it costs the GPU in flight")
        res["roofline"]["step"] = st
    if not a.no_extra:
        try:
            t_sat = saturated_launch_time(hp)
            res["roofline"]["four_streams"] = {"us_per_launch": round(t_sat, 2), "achieved": round(flops / (t_sat * 1e-6) / 1e12, 2),
                                                "frac": round(flops / (t_sat * 1e-6) / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                                                "note": "the same launch replayed by four independent streams at once (wall time / launches)"}
        except Exception as e:                                   # an extra data point: never fail the bench line
            res["roofline"]["four_streams"] = {"error": str(e)[:200]}
    if world == 1 and (B, H) == (64, 256) and not a.no_extra:
        try:
            res["8bit_content"] = content_8bit(dev, z, cb, vq, codec, ratio, value)
        except Exception as e:                                   # an extra data point: never fail the bench line
            res["8bit_content"] = {"error": str(e)[:300]}
        hp.step()
        res["mask_mismatch"] = mask_mismatch(hp, x, z, cb, ratio)
        try:
            res["mask_mismatch"]["tie_heavy_content"] = mask_flip_families(dev, z, cb, vq, codec, ratio)
        except Exception as e:                                   # an extra data point: never fail the bench line
            res["mask_mismatch"]["tie_heavy_content"] = {"error": str(e)[:200]}
        res["ratio_sweep"] = ratio_sweep(dev, x, z, cb, vq, codec)
        try:
            res["input_regimes"] = input_regimes(dev, x, ratio)
        except Exception as e:                                   # an extra data point: never fail the bench line
            res["input_regimes"] = {"error": str(e)[:300]}
        res["b1_latency"] = b1_latency(dev, cb, vq, codec)
        res["div2k_image"] = div2k_image(dev, cb, vq, codec)
        res["div2k_tiles"] = [tiles_768(dev, cb, vq, codec, 8, 60), tiles_768(dev, cb, vq, codec, 32, 30)]
        try:
            res["mixed_stream"] = mixed_extra(dev, vq, codec, ratio)
        except Exception as e:                                   # an extra data point: never fail the bench line
            res["mixed_stream"] = {"error": str(e)[:300]}
        try:
            res["end_to_end_estimate"] = end_to_end_estimate(dev, res["single_batch"]["ms_per_step"], B, H, W)
        except Exception as e:                                   # (MIOpen missing / out of memory: the estimate is optional)
            res["end_to_end_estimate"] = {"error": str(e)[:200]}
    if not a.no_cpu_baseline and world == 1:             # the CPU port is timed at N=1 only (rank 0's host cores)
        res["cpu_baseline"] = cpu_baseline(x, z, cb, ratio)
    # ---- the line: compact
    line["bpp"], line["bpp_match"] = res.pop("bpp"), res.pop("bpp_match")
    rf = res["roofline"]
    line["roofline"] = {k: rf[k] for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "duration_us", "frac_rocprof", "duration_rocprof_us",
                                           "in_step_us", "mfma_busy_frac") if k in rf}
    line["roofline"]["kernel"] = "vq_filter_router_kernel"
    if "step" in rf:
        line["roofline"]["step"] = {k: rf["step"][k] for k in ("hbm_bytes", "flops", "floor_hbm_us", "floor_fp32_us", "step_frac", "resource_us",
                                                               "resource_us_sum") if k in rf["step"]}
    if "cpu_baseline" in res:
        cbl = res.pop("cpu_baseline")
        line["cpu_baseline"] = {k: cbl[k] for k in ("value", "unit", "cores", "kind", "sample") if k in cbl}
        res["cpu_baseline_detail"] = cbl
    par = {"timed_batches_bpp_match": line["bpp_match"]}
    mm = res.get("mask_mismatch")
    if mm:
        par["timed_batch_from_pixels"] = {k: mm[k] for k in ("images", "differing_mask_elements", "differing_bin_files") if k in mm}
        th = mm.get("tie_heavy_content") or {}
        par["tie_heavy_content"] = {k: [v.get("differing_mask_elements"), v.get("differing_bin_files")] for k, v in th.items()
                                    if isinstance(v, dict) and "differing_mask_elements" in v}
        par["tie_heavy_launch_us"] = {k: [v.get("vq+router_us"), v.get("vq+router_no_refinement_us")] for k, v in th.items()
                                      if isinstance(v, dict) and "vq+router_us" in v}
    line["parity"] = par
    if "8bit_content" in res:
        c8 = res["8bit_content"]
        line["8bit_content"] = {k: (v["MPixels/s"] if isinstance(v, dict) else v) for k, v in c8.items() if k != "note"}
        line["8bit_content"]["bpp_match"] = all(v["bpp_match"] for v in c8.values() if isinstance(v, dict))
    if "one_batch_in_flight" in res:
        line["one_batch_in_flight_us"] = round(res["one_batch_in_flight"]["ms_per_step"] * 1e3, 2)
    if isinstance(res.get("b1_latency"), dict):
        line["b1_latency_us"] = {k: v for k, v in res["b1_latency"].items() if isinstance(v, (int, float))}
    return line, res


# ------------------------------------------------------------------------------------------------ launcher
def _spawned(local_rank, a, port):
    os.environ["RANK"] = os.environ["LOCAL_RANK"] = str(local_rank)
    os.environ["WORLD_SIZE"] = str(a.gpus)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    run_rank(a, local_rank, a.gpus, local_rank)


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", choices=["batch", "mixed"], default="batch",
                    help="batch: BASELINE config 2 (the headline); mixed: config 5, a Kodak + DIV2K stream sharded over the ranks (total work fixed: strong scaling)")
    ap.add_argument("--div2k", type=int, default=8, help="--workload mixed: DIV2K-sized images in the stream (next to 24 Kodak-sized ones)")
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--schedule", choices=["pipelined", "sequential"], default="sequential",
                    help="sequential: one hipGraph per batch on one stream; pipelined: BatchStream, encode side of batch i+1 next to the decode side of batch i")
    ap.add_argument("--slots", type=int, default=0, help="distinct resident input batches in rotation (0: enough to exceed the Infinity Cache)")
    ap.add_argument("--lanes", type=int, default=4, help="independent HIP streams the rotation of batches is dealt over (sequential schedule); 1 = one batch in flight")
    ap.add_argument("--max-ring", type=int, default=8, help="batches per hipGraph launch of a lane (sequential schedule)")
    ap.add_argument("--split-router", action="store_true", help="router as its own launch instead of riding in the VQ launch")
    ap.add_argument("--no-ring", action="store_true", help="one hipGraph per batch instead of one per rotation of batches (sequential schedule)")
    ap.add_argument("--no-graph", action="store_true", help="launch eagerly instead of replaying hipGraphs (sequential schedule)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-dist", action="store_true", help="N=1 only: do not create the one-rank communicator")
    ap.add_argument("--no-report", action="store_true", help="only the timed loop and the headline fields (for kernel traces: the last K chains of the trace are the timed steps)")
    ap.add_argument("--no-extra", action="store_true", help="skip the extra data points (mask mismatch, ratio sweep, DIV2K, B=1)")
    ap.add_argument("--extras-file", default="", help="where the extras that do not go on the JSON line are written (default gpurun_out/bench_extras.json)")
    ap.add_argument("--dry-nccl", action="store_true",
                    help="N=1 only: run the single rank in a SPAWNED child process (torch.multiprocessing.spawn + init_process_group('nccl', device_id=...)), "
                         "exactly as the ranks of --gpus N>1 are started, instead of in this process")
    ap.add_argument("--stub", action="store_true", help=argparse.SUPPRESS)      # CPU test of the launcher only (gloo, no kernels)
    return ap.parse_args(argv)


def main(argv=None):
    a = parse_args(argv)
    if a.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if "RANK" in os.environ:                                # launched by torch.distributed.run: one process per GPU already
        rank = int(os.environ["RANK"])
        world = int(os.environ.get("WORLD_SIZE", "1"))
        local = int(os.environ.get("LOCAL_RANK", str(rank)))
        if world != a.gpus:
            raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; refusing to report a {a.gpus}-GPU number")
        run_rank(a, rank, world, local)
        return
    if a.gpus == 1 and not a.dry_nccl:
        run_rank(a, 0, 1, 0)
        return
    # plain `python bench.py --gpus N`: start the N ranks here, one per GPU
    if not a.stub and torch.cuda.device_count() < a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but only {torch.cuda.device_count()} GPUs are visible")
    import torch.multiprocessing as mp
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    mp.spawn(_spawned, args=(a, free_port()), nprocs=a.gpus, join=True)      # raises (non-zero exit) if any rank fails


if __name__ == "__main__":
    main()
