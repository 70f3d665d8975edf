"""GPU probe: phase clocks of merge_kernel / decode kernels for T tiles of SxS (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
import bench
T = int(sys.argv[1]); S = int(sys.argv[2])
dev = torch.device("cuda")
x, z, cb = bench.make_inputs(T, S, S, 77)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
comp = hp.out[6]
l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
for _ in range(3): hp.codec.decompress(comp)
torch.cuda.synchronize()
c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
print("merge_kernel block 0 phases (us @2.1GHz): loads %.2f | bitsets+prefix %.2f | own-band prefix %.2f | scatter/gather %.2f | total %.2f" % (tuple((c[i+1]-c[i])/2.1e3 for i in range(10, 14)) + ((c[14]-c[10])/2.1e3,)))
print("raw:", [c[i] - c[10] for i in range(10, 15)])
