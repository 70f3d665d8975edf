"""GPU probe: when does each batch of the K=20 window complete, per lane (one graph per batch so that an event fits between them)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
dev = torch.device("cuda", 0)
slots_np = [bench.make_inputs(64, 256, 256, seed=1000 + s) for s in range(8)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
slots = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
hist = torch.zeros(1024, dtype=torch.int64, device=dev)
ls = cg.pipeline.LaneStream(vq, 0.1, 0.8, slots, lanes=4, frequency=codec.huffman, hist=hist, max_ring=1, quick_start=False)
ls.capture(); ls.prepare(20)
for rep in range(4):
    ls.submit(5); ls.join(); torch.cuda.synchronize()
    base = [torch.cuda.Event(enable_timing=True) for _ in ls.lanes]
    for lane, ev in zip(ls.lanes, base): ev.record(lane["stream"])
    torch.cuda.synchronize()
    evs = [[torch.cuda.Event(enable_timing=True) for _ in range(5)] for _ in ls.lanes]
    host = []
    t0 = time.perf_counter()
    for i in range(5):
        for j, lane in enumerate(ls.lanes):
            with torch.cuda.stream(lane["stream"]):
                ls._graph(lane, (lane["pos"] + i) % len(lane["slots"]), 1)[0].replay()
                evs[j][i].record(lane["stream"])
            host.append((time.perf_counter() - t0) * 1e6)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e6
    # event times relative to each lane's base event; the base events were recorded at about the same host time (idle GPU)
    off = [b.elapsed_time(base[0]) for b in base]
    print(f"rep {rep}: wall {wall:.0f} us; host time of the 20 launches: first {host[0]:.0f} last {host[-1]:.0f} us")
    ref = None
    for j in range(4):
        t = [base[j].elapsed_time(e) * 1e3 for e in evs[j]]
        if ref is None: ref = t[0] - 0
        print(f"  lane {j}: batch ends (us after its base event): " + " ".join(f"{v:.0f}" for v in t) + "   durations: " + " ".join(f"{b - a:.0f}" for a, b in zip([t[0] - (t[1] - t[0])] + t[:-1], t)))
