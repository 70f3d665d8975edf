"""GPU probe: workgroup start/end times of the fused VQ + router launch (CGIC_LIB=.../libcgic_hip_dbg.so), by g_early"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from tools_probe import graph_time
g = torch.Generator().manual_seed(0)
B = 64
z = torch.randn(B, 4, 64, 64, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
e16 = (torch.rand(B, 16, 16, generator=g) * 2.6).cuda(); e8 = (torch.rand(B, 32, 32, generator=g) * 2.6).cuda()
l = _lib.lib()
f = lambda: vq_forward_route(z, w, 0.25, True, e16, e8, 0.1, 0.8, per_image=True)
print("fused launch, graph-timed: best %.2f mean %.2f us" % graph_time(f))
for _ in range(3): f()
torch.cuda.synchronize()
n = 320
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
t0 = t[:, 0].min()
st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
for name, sl in (("router workgroups [0,64)", slice(0, 64)), ("early VQ workgroups", slice(64, 256)), ("late VQ workgroups", slice(256, 320))):
    print(name, "start min/med/max %.1f %.1f %.1f | end min/med/max %.1f %.1f %.1f | duration med %.1f" % (
        st[sl].min(), np.median(st[sl]), st[sl].max(), en[sl].min(), np.median(en[sl]), en[sl].max(), np.median((en - st)[sl])))
