"""GPU probe: workgroup start / end times of compress_streams_kernel (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import HotPath, make_inputs
dev = torch.device("cuda")
x, z, cb = make_inputs(64, 256, 256, 1000)
hp = HotPath(dev, x, z, cb, (0.1, 0.8))
e8, e16, mask, mode, zq, ind, comp = hp.encode()
for _ in range(3): comp = hp.codec.compress(ind, mask, mode, hist=hp.hist)
torch.cuda.synchronize()
l = _lib.lib(); n = 384
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
t0 = t[:, 0].min(); st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
print("starts percentiles 0/25/50/75/100:", [round(float(np.percentile(st, q)), 2) for q in (0, 25, 50, 75, 100)])
print("ends   percentiles 0/25/50/75/100:", [round(float(np.percentile(en, q)), 2) for q in (0, 25, 50, 75, 100)])
for s_ in range(6):
    sel = np.arange(n) % 6 == s_
    print(f"stream job {s_}: duration median {np.median((en - st)[sel]):.2f} us, start median {np.median(st[sel]):.2f}, end max {en[sel].max():.2f}")
