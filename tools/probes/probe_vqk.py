"""GPU probe: phase clocks of vq_mfma_kernel, workgroup 0 (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import _vq_forward
from bench import time_events
g = torch.Generator().manual_seed(0)
z = torch.randn(64, 4, 64, 64, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
for name, f in (("full (z_q + loss)", lambda: _vq_forward(z, w, 0.25, True, None)), ("indices only", lambda: _vq_forward(z, w, 0.25, True, None, False, False))):
    for _ in range(3): f()
    torch.cuda.synchronize()
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
    print(name, " | ".join(f"{n} {(c[i+1]-c[i])/2.29e3:.2f}" for i, n in enumerate(["stage codebook", "split z (last group)", "scan", "decide", "outputs", "loss"])), "| total %.2f us; events %.1f us" % ((c[6]-c[0])/2.29e3, time_events(f, 100)), "| wave 3: loop end at %.2f, kernel end %.2f us" % ((c[13]-c[0])/2.29e3, (c[14]-c[0])/2.29e3))
for name, f in (("full", lambda: _vq_forward(z, w, 0.25, True, None)), ("indices only", lambda: _vq_forward(z, w, 0.25, True, None, False, False))):
    f(); f(); torch.cuda.synchronize()
    n = 256
    buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
    t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
    t0 = t[:, 0].min()
    st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
    print(name, "workgroups: start max %.1f us; end percentiles 0/25/50/75/90/99/100 = %s us" % (st.max(), " / ".join("%.1f" % np.percentile(en, q) for q in (0, 25, 50, 75, 90, 99, 100))))
