"""GPU probe: the one-wave router inside the VQ launch (CGIC_LIB=.../libcgic_hip_dbg.so): phase clocks of workgroup 0's last wave"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from tools_probe import graph_time
g = torch.Generator().manual_seed(0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
z = torch.randn(B, 4, 64, 64, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
e16 = (torch.rand(B, 16, 16, generator=g) * 2.6).cuda(); e8 = (torch.rand(B, 32, 32, generator=g) * 2.6).cuda()
l = _lib.lib()
f = lambda: vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True)
print("fused launch, graph-timed: best %.2f mean %.2f us" % graph_time(f))
if hasattr(l, "cgic_debug_phase_clocks"):
    l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    for _ in range(3): f()
    torch.cuda.synchronize()
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
    us = lambda a, b: (c[b] - c[a]) / 2.1e3
    print("wg0: start->stage done %.2f | ->loop end(w0) %.2f | ->end %.2f" % (us(0, 1), us(0, 5), us(0, 6)))
    print("router wave: starts at %.2f | loads+coarse select %.2f | coarse masks+parents %.2f | medium select %.2f | masks %.2f | total %.2f" % (
        us(0, 20), us(20, 21), us(21, 22), us(22, 23), us(23, 24), us(20, 24)))
    f(); f(); torch.cuda.synchronize()
    n = 256
    buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
    t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
    own = np.arange(n) % (n // B) == 0 if B < n else np.ones(n, bool)
    for name, sl in (("routing workgroups", own), ("other workgroups", ~own)):
        if sl.any():
            print(name, "start min/med/max %.1f %.1f %.1f | end min/med/max %.1f %.1f %.1f | duration med %.1f max %.1f" % (
                st[sl].min(), np.median(st[sl]), st[sl].max(), en[sl].min(), np.median(en[sl]), en[sl].max(), np.median((en - st)[sl]), (en - st)[sl].max()))
