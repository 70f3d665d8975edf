"""GPU probe: per-kernel time of the hot path's launches for T tiles of SxS (graph-timed), router phase clocks with the dbg lib"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import _vq_forward, vq_forward_route
from tools_probe import graph_time
import bench
T = int(sys.argv[1]) if len(sys.argv) > 1 else 8
S = int(sys.argv[2]) if len(sys.argv) > 2 else 768
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(T, S, S, seed=77)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
e8, e16, mask, mode, zq, ind, comp = hp.out[:7]
router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
res = {
 "entropy": graph_time(lambda: cg.entropy_maps(hp.x)),
 "router alone": graph_time(lambda: router(e16, e8, want_gate=False)),
 "vq alone": graph_time(lambda: _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None)),
 "vq+router fused": graph_time(lambda: vq_forward_route(hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8)),
 "compress": graph_time(lambda: hp.codec.compress(ind, mask, mode, hist=hp.hist)),
 "decompress (decode + merge)": graph_time(lambda: hp.codec.decompress(comp)),
}
for k, (b, m) in res.items(): print(f"{T}x{S}x{S} {k:28s} best {b:7.2f} mean {m:7.2f} us")
l = _lib.lib()
if hasattr(l, "cgic_debug_phase_clocks"):
    l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    for _ in range(3): router(e16, e8, want_gate=False)
    torch.cuda.synchronize()
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
    names = ["stage", "coarse select", "coarse mask", "medium select", "medium mask", "fine mask"]
    print("router phases (us @2.1GHz):", " | ".join(f"{n} {(c[i+1]-c[i])/2.1e3:.2f}" for i, n in enumerate(names)))
