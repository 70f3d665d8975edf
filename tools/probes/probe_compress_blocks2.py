"""GPU probe: per-job duration of compress_streams_kernel for T tiles of SxS (CGIC_LIB=.../libcgic_hip_dbg.so)"""
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
e8, e16, mask, mode, zq, ind, comp = hp.out[:7]
for _ in range(3): comp = hp.codec.compress(ind, mask, mode, hist=hp.hist)
torch.cuda.synchronize()
l = _lib.lib(); n = T * 16
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
ok = t[:, 0] > 0
t0 = t[ok, 0].min(); st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
# jobs in launch order: the parts of the fine stream, the medium one's, the coarse one's, mask_medium, mask_coarse, histogram
for j in range(16):
    sel = (np.arange(n) % 16 == j) & ok
    if sel.any():
        print(f"job {j:2d}: duration median {np.median((en - st)[sel]):6.2f} us, start median {np.median(st[sel]):5.2f}, end max {en[sel].max():6.2f}")
l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
print("first fine-stream workgroup of image 0, phases (us @2.1GHz): setup %.2f | phase A %.2f | exchange %.2f | pack %.2f" % ((c[1]-c[0])/2.1e3, (c[2]-c[1])/2.1e3, (c[4]-c[2])/2.1e3, (c[3]-c[4])/2.1e3))
