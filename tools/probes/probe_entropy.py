"""GPU probe: phase clocks of entropy_maps_kernel (needs CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import time_events
g = np.random.default_rng(0)
xx = torch.from_numpy(g.random((64, 3, 256, 256), dtype=np.float32)).cuda()
for _ in range(3): cg.entropy_maps(xx)
torch.cuda.synchronize()
l = _lib.lib()
clk = (ctypes.c_longlong * 32)()
l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
l.cgic_debug_phase_clocks(clk)
c = list(clk)
names = {17: "loads done+gray", 18: "lds store", 19: "barrier", 20: "sub-patch 0", 21: "sub-patch 1", 22: "sub-patch 2", 23: "sub-patch 3", 24: "p16 + end"}
for i in range(17, 25):
    print(f"  {names[i]:16s} +{(c[i]-c[i-1])/2.29e3:6.2f} us  (t={(c[i]-c[16])/2.29e3:6.2f})")
print("entropy_maps B=64: %.1f us" % time_events(lambda: cg.entropy_maps(xx), 50))
cg.entropy_maps(xx); torch.cuda.synchronize()
n = 1024
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
print("workgroups: start min/med/p90/max %.2f %.2f %.2f %.2f | end min/med/p90/max %.2f %.2f %.2f %.2f | life med %.2f max %.2f us" % (
    st.min(), np.median(st), np.percentile(st, 90), st.max(), en.min(), np.median(en), np.percentile(en, 90), en.max(), np.median(en - st), (en - st).max()))
