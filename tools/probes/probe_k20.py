"""GPU probe: where the driver's short window (--steps 20 --warmup 5) loses against the steady state.
For each LaneStream configuration: wall time of submit(20)+join after submit(5)+join (bench.py's bracket), the host time of the
submit call itself, and per-lane GPU start / end offsets (events on the lane streams against an event on the null stream)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench

dev = torch.device("cuda", 0)
K = int(sys.argv[1]) if len(sys.argv) > 1 else 20
slots_np = [bench.make_inputs(64, 256, 256, seed=1000 + s) for s in range(8)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
slots = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
for lanes, max_ring, threads, one, native in ((4, 8, False, False, 0), (4, 8, False, False, 1), (4, 8, False, False, 2), (4, 8, False, False, 0), (4, 8, False, False, 1), (4, 8, False, False, 2)):
    hist = torch.zeros(1024, dtype=torch.int64, device=dev)
    ls = cg.pipeline.LaneStream(vq, 0.1, 0.8, slots, lanes=lanes, frequency=codec.huffman, hist=hist, max_ring=max_ring, launch_threads=threads, quick_start=one, native_launch=native)
    ls.capture()
    res, host = [], []
    for rep in range(9):
        ls.prepare(5); ls.submit(5); ls.join(); torch.cuda.synchronize()
        ls.prepare(K)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        ls.submit(K)
        t1 = time.perf_counter()
        ls.join(); torch.cuda.synchronize()
        t2 = time.perf_counter()
        res.append((t2 - t0) * 1e6); host.append((t1 - t0) * 1e6)
    # lane start / end offsets of one more run
    ls.prepare(5); ls.submit(5); ls.join(); torch.cuda.synchronize()
    ls.prepare(K)
    base = torch.cuda.Event(enable_timing=True)
    starts = [torch.cuda.Event(enable_timing=True) for _ in ls.lanes]
    ends = [torch.cuda.Event(enable_timing=True) for _ in ls.lanes]
    torch.cuda.synchronize()
    base.record()
    for lane, ev in zip(ls.lanes, starts):
        lane["stream"].wait_stream(torch.cuda.current_stream())
        ev.record(lane["stream"])
    ls.submit(K)
    for lane, ev in zip(ls.lanes, ends):
        ev.record(lane["stream"])
    torch.cuda.synchronize()
    offs = [(round(base.elapsed_time(s) * 1e3), round(base.elapsed_time(e) * 1e3)) for s, e in zip(starts, ends)]
    res.sort(); host.sort()
    print(f"lanes={lanes} max_ring={max_ring} threads={threads} quick_start={one} native_launch={native}: K={K} wall min {res[0]:.0f} med {res[len(res)//2]:.0f} max {res[-1]:.0f} us "
          f"({res[len(res)//2]/K:.1f} us/step), host submit med {host[len(host)//2]:.0f} us, lane (start,end) us {offs}", flush=True)
    del ls
