"""GPU probe: aggregate time per launch of each kernel of the step when 4 independent streams replay graphs of that
kernel alone (its 'resource time' under self-contention) vs alone on one stream (its latency)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
from control_gic_amd.quantize import _vq_forward, vq_forward_route
from control_gic_amd.pipeline import distinct_queue_streams
dev = torch.device("cuda", 0)
NL = int(sys.argv[1]) if len(sys.argv) > 1 else 4
slots_np = [bench.make_inputs(64, 256, 256, seed=s) for s in range(NL)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
hps = [bench.HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec) for x, z, _ in slots_np]
for h in hps: h.step()
torch.cuda.synchronize()
streams = distinct_queue_streams(dev, NL)
N = 20
def fns(hp):
    e8, e16, mask, mode, zq, ind, comp = hp.out[:7]
    return {
        "entropy": lambda: cg.entropy_maps(hp.x),
        "vq+router": lambda: vq_forward_route(hp.z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, prepared=hp.pipe.prepared),
        "vq": lambda: _vq_forward(hp.z, vq.embedding.weight, 0.25, True, None),
        "router": lambda: hp.router(e16, e8, want_gate=False),
        "compress": lambda: codec.compress(ind, mask, mode, hist=hp.hist),
        "decode+merge (throughput)": lambda: codec.decompress(comp, decoder="throughput"),
        "decode+merge (latency)": lambda: codec.decompress(comp, decoder="latency"),
    }
F = [fns(h) for h in hps]
def make_graph(fn, stream):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(N): fn()
    torch.cuda.synchronize()
    return g
def run(gs, reps=8):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g, s in gs:
            with torch.cuda.stream(s): g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6
tot1 = tot4 = 0
for k in F[0]:
    gs = [(make_graph(F[j][k], streams[j]), streams[j]) for j in range(NL)]
    t1 = run(gs[:1]) / N
    t4 = run(gs) / (N * NL)
    print(f"{k:14s} alone {t1:6.2f} us   {NL} lanes {t4:6.2f} us per launch")
    if k not in ("vq", "router", "decode+merge (latency)"): tot1 += t1; tot4 += t4
print(f"sum over the step's four launches: alone {tot1:.1f}, self-contended {tot4:.1f}")
