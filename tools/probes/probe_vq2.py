"""GPU probe (round 2): the VQ kernel alone, timed from a hipGraph of 20 launches (no host-bound gaps), with the
dbg build's phase clocks and workgroup end times when CGIC_LIB points at libcgic_hip_dbg.so; checks the filter path
against the VALU restatement on the same inputs.
usage: python tools/probes/probe_vq2.py [B] [size]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import _vq_forward
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 256
g = torch.Generator().manual_seed(0)
z = torch.randn(B, 4, S // 4, S // 4, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
l = _lib.lib()
dbg = hasattr(l, "cgic_debug_phase_clocks")


def graph_time(fn, per_graph=20, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr, stream=side):
            for _ in range(per_graph): fn()
    torch.cuda.current_stream().wait_stream(side)
    gr.replay(); torch.cuda.synchronize()
    best = 1e9; tot = 0.0
    for _ in range(reps):
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(); gr.replay(); e.record(); e.synchronize()
        t = s.elapsed_time(e) * 1e3 / per_graph
        best = min(best, t); tot += t
    return best, tot / reps


full = lambda: _vq_forward(z, w, 0.25, True, None)
idx_only = lambda: _vq_forward(z, w, 0.25, True, None, False, False)
zq, loss, idx = full()
zq2, loss2, idx2 = _vq_forward(z, w, 0.25, True, None, kernel="valu")
torch.cuda.synchronize()
print("filter == valu:", bool(torch.equal(idx, idx2)), bool(torch.equal(zq, zq2)), "loss", float(loss), float(loss2))
for name, f in (("full", full), ("indices only", idx_only)):
    b, a = graph_time(f)
    print(f"{name:13s} B={B} {S}x{S}: graph-timed per launch: best {b:.2f} us, mean {a:.2f} us")
if dbg:
    l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    for name, f in (("full", full), ("indices only", idx_only)):
        for _ in range(3): f()
        torch.cuda.synchronize()
        c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
        names = ["stage", "first group(s)", "scan(last)", "decide(last)", "outputs(last)", "tail"]
        print(name, " | ".join(f"{n} {(c[i+1]-c[i])/2.1e3:.2f}" for i, n in enumerate(names)), "| total %.2f us @2.1GHz" % ((c[6]-c[0])/2.1e3),
              "| wave 3: loop end %.2f, end %.2f" % ((c[13]-c[0])/2.1e3, (c[14]-c[0])/2.1e3))
        f(); f(); torch.cuda.synchronize()
        n = 256
        buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
        t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
        t0 = t[:, 0].min()
        st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
        print(name, "workgroups: start max %.1f us; end percentiles 0/25/50/75/90/99/100 = %s us" % (st.max(), " / ".join("%.1f" % np.percentile(en, q) for q in (0, 25, 50, 75, 90, 99, 100))))
