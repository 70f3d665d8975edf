"""GPU probe for build variants (CGIC_LIB=tmp_libs/lib_X.so): VQ alone / fused with the router (graph-timed, prepared codebook),
and the four-lane stream rate at K=200."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
from control_gic_amd.quantize import _vq_forward, vq_forward_route
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
prep = hp.pipe.prepared
e8, e16 = cg.entropy_maps(hp.x)
w = hp.vq.embedding.weight
t_alone = min(bench.graph_kernel_time(lambda: _vq_forward(hp.z, w, 0.25, True, None, prepared=prep)) for _ in range(3))
t_fused = min(bench.graph_kernel_time(lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep)) for _ in range(3))
out = f"vq alone {t_alone:.2f} us, fused {t_fused:.2f} us"
if "nolanes" not in sys.argv:
    slots_np = [bench.make_inputs(64, 256, 256, seed=1000 + s) for s in range(8)]
    slots = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b, _ in slots_np]
    codec = cg.GrainCodec(hp.vq.embedding_counter, w)
    for with_hist in (False, True):
        hist = torch.zeros(1024, dtype=torch.int64, device=dev) if with_hist else None
        ls = cg.pipeline.LaneStream(hp.vq, 0.1, 0.8, slots, lanes=4, frequency=codec.huffman, hist=hist, quick_start=False)
        ls.capture(); ls.prepare(200); ls.submit(40); ls.join(); torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            t0 = time.perf_counter(); ls.submit(200); ls.join(); torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) / 200)
        ts.sort()
        out += f", 4 lanes K=200 hist={with_hist}: best {ts[0] * 1e6:.2f} median {ts[3] * 1e6:.2f} worst {ts[-1] * 1e6:.2f} us/step"
        del ls
print(out, flush=True)
