"""GPU probe: do two kernels of DIFFERENT batches overlap when replayed as two hipGraphs on two streams?
(decode+merge | decode | merge | router | compress) next to (entropy | VQ).  Reports alone / alone / together per pair."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
import bench
from control_gic_amd.quantize import _vq_forward
dev = torch.device("cuda", 0)
slots_np = [bench.make_inputs(64, 256, 256, seed=s) for s in range(2)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
hp = [bench.HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec) for x, z, _ in slots_np]
for h in hp: h.step()
torch.cuda.synchronize()
e8, e16, mask, mode, zq, ind, comp = hp[0].out[:7]
router = hp[0].router
N = 20
def make_graph(fn, stream):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    with torch.cuda.stream(stream):
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=stream):
            for _ in range(N): fn()
    torch.cuda.synchronize()
    return g
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
fns = {
    "decode+merge": lambda: codec.decompress(comp),
    "router": lambda: router(e16, e8, want_gate=False),
    "compress": lambda: codec.compress(ind, mask, mode, hist=hp[0].hist),
    "entropy": lambda: cg.entropy_maps(hp[1].x),
    "vq": lambda: _vq_forward(hp[1].z, vq.embedding.weight, 0.25, True, None),
}
def run(gs, reps=10):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for g, s in gs:
            with torch.cuda.stream(s): g.replay()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e6 / N
graphs = {}
for k in fns:
    graphs[k] = (make_graph(fns[k], sA), make_graph(fns[k], sB))
for a in ("decode+merge", "router", "compress"):
    for b in ("entropy", "vq"):
        ta = run([(graphs[a][0], sA)]); tb = run([(graphs[b][1], sB)]); tab = run([(graphs[a][0], sA), (graphs[b][1], sB)])
        print(f"{a:14s} {ta:6.2f} | {b:8s} {tb:6.2f} | together {tab:6.2f} us per pair (sum {ta+tb:6.2f}, max {max(ta,tb):6.2f})")
