"""GPU probe (dbg build, CGIC_LIB=.../libcgic_hip_dbg.so): phase clocks of every launch of the B-image step, each launch alone.
Prints, per launch, the stamps that moved (index: microseconds since the launch's earliest stamp, GPU clock taken as 2.1 GHz)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from control_gic_amd import _lib
from control_gic_amd.quantize import _vq_forward, vq_forward_route

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(B, S, S, 1)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
cg = hp.cg
l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]


def clocks():
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c)
    return np.array(list(c), dtype=np.int64)


e8, e16 = cg.entropy_maps(hp.x)
mask, _, _, mode = hp.router(e16, e8, want_gate=False)
prep = hp.pipe.prepared
_, _, ind = _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None)
comp = hp.codec.compress(ind, mask, mode)
stages = {
    "entropy_maps": lambda: cg.entropy_maps(hp.x),
    "router_alone": lambda: hp.router(e16, e8, want_gate=False),
    "vq+router fused": lambda: vq_forward_route(hp.z, hp.vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=hp.x),
    "vq alone": lambda: _vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None, prepared=prep),
    "compress": lambda: hp.codec.compress(ind, mask, mode, hist=hp.hist),
    "decompress": lambda: hp.codec.decompress(comp),
    "decompress (throughput decoder)": lambda: hp.codec.decompress(comp, decoder="throughput"),
}
for name, f in stages.items():
    for _ in range(3): f()
    torch.cuda.synchronize()
    before = clocks()
    f(); torch.cuda.synchronize()
    after = clocks()
    moved = [i for i in range(32) if after[i] != before[i]]
    if not moved:
        print(name, ": no stamps"); continue
    t0 = min(after[i] for i in moved)
    print(f"{name} (graph-timed {bench.graph_kernel_time(f):.2f} us): " + "  ".join(f"[{i}] {(after[i] - t0) / 2.1e3:.2f}" for i in sorted(moved, key=lambda i: after[i])))
