"""GPU probe (dbg build: CGIC_LIB=control-gic_amd/libcgic_hip_dbg.so): timeline of the filter VQ kernel alone at B=64 x 64x64
latents -- workgroup start / end distribution and the phase stamps of workgroup 0 (waves 0 and 3)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import _vq_forward, prepare_codebook
g = torch.Generator().manual_seed(0)
B = 64
z = torch.randn(B, 4, 64, 64, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
prep = prepare_codebook(w) if "noprep" not in sys.argv else None
l = _lib.lib()
full = "idx" not in sys.argv
f = lambda: _vq_forward(z, w, 0.25, True, None, full, full, prepared=prep)
for _ in range(4): f()
torch.cuda.synchronize()
big = (ctypes.c_longlong * (2 * 4096))(); l.cgic_debug_block_times(big, 4096)
c0 = np.array(list(big), dtype=np.int64).reshape(4096, 2)[2048:2048 + 256].copy()
f()
torch.cuda.synchronize()
l.cgic_debug_block_times(big, 4096)
c1 = np.array(list(big), dtype=np.int64).reshape(4096, 2)[2048:2048 + 256]
cnt = c1 - c0                     # per workgroup: (flagged vectors, other-half blocks) of the last launch
n = 256
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
t0 = t[:, 0].min()
st = (t[:, 0] - t0) / 100.0; en = (t[:, 1] - t0) / 100.0
print("workgroups: start min/med/max %.2f %.2f %.2f | end min/med/max %.2f %.2f %.2f | duration min/med/max %.2f %.2f %.2f us" % (
    st.min(), np.median(st), st.max(), en.min(), np.median(en), en.max(), (en - st).min(), np.median(en - st), (en - st).max()))
dur = en - st
order = np.argsort(dur)
print("flagged vectors per workgroup: total %d max %d | other-half blocks: total %d (of %d groups)" % (cnt[:, 0].sum(), cnt[:, 0].max(), cnt[:, 1].sum(), 4096))
print("corr(duration, flagged) = %.2f; slowest 8 workgroups: %s" % (np.corrcoef(dur, cnt[:, 0])[0, 1], [(int(i), round(float(dur[i]), 1), int(cnt[i, 0]), int(cnt[i, 1])) for i in order[-8:]]))
print("fastest 8: %s" % [(int(i), round(float(dur[i]), 1), int(cnt[i, 0]), int(cnt[i, 1])) for i in order[:8]])
for k in range(0, 6):
    sel = cnt[:, 0] == k
    if sel.any(): print(f"  workgroups with {k} flagged: {sel.sum()}, median duration {np.median(dur[sel]):.2f}")
print("duration by XCD (blk % 8): " + " ".join(f"{np.median(dur[np.arange(256) % 8 == x]):.2f}" for x in range(8)))
ph = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(ph)
p = np.array(list(ph), dtype=np.int64)
ghz = 2.4
names = ["start", "staged", "prep g0", "scan g0", "decide g0", "outputs.. end loop", "handoff"]
w0 = [(p[i] - p[0]) / ghz / 1e3 for i in range(7)]
print("wg0 wave0 stamps (us from start; stamps 2-4 are of the LAST group of the wave):", " | ".join(f"{n} {v:.2f}" for n, v in zip(names, w0)))
w3 = [(p[8 + i] - p[0]) / ghz / 1e3 for i in range(2, 8)]
print("wg0 wave3 stamps:", " ".join(f"{v:.2f}" for v in w3))
if "starts" in sys.argv:
    print("start by workgroup (us):", " ".join(f"{v:.1f}" for v in st))
    print("end by workgroup (us):", " ".join(f"{v:.1f}" for v in en))
if "events" in sys.argv:
    # is the start stagger by XCD real, or are the XCDs' clocks offset?  kernel time by HIP events (isolated launches) vs the device span
    ts = []
    for _ in range(20):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    print("isolated launch by HIP events: min %.2f med %.2f us; device span (max end - min start) %.2f us" % (min(ts), sorted(ts)[10], en.max() - st.min()))
    import bench
    print("back to back in a graph: %.2f us per launch" % bench.graph_kernel_time(f))
if "waves" in sys.argv:
    f(); torch.cuda.synchronize()
    l.cgic_debug_block_times(big, 4096)
    a = np.array(list(big), dtype=np.int64).reshape(4096, 2)
    wv = a[512:512 + 512].reshape(64, 8, 2)
    wg = a[:64]
    for g in (0, 1, 2, 3, 8, 9, 16, 40):
        t0g = wg[g, 0]
        print(f"wg {g}: loop start per wave " + " ".join(f"{(wv[g, w, 0] - t0g) / 100:.1f}" for w in range(8)) + " | loop end " + " ".join(f"{(wv[g, w, 1] - t0g) / 100:.1f}" for w in range(8)) + f" | wg end {(wg[g, 1] - t0g) / 100:.1f}")
    le = (wv[:, :, 1] - wg[:, :1]) / 100.0
    print("loop end by wave index, median over 64 workgroups:", " ".join(f"{np.median(le[:, w]):.2f}" for w in range(8)), "| max over waves, median:", f"{np.median(le.max(1)):.2f}", "| min:", f"{np.median(le.min(1)):.2f}")
if "acc" in sys.argv:
    # per-phase clock sums over the groups of wave 0 / wave 4 of workgroup 0 (CGIC_PHASE_ACC): prep | scan | decide | outputs
    l.cgic_debug_phase_clocks(ph); a0 = np.array(list(ph), dtype=np.int64).copy()
    f(); torch.cuda.synchronize()
    l.cgic_debug_phase_clocks(ph); a1 = np.array(list(ph), dtype=np.int64)
    d = (a1 - a0) / ghz / 1e3
    print("wave 0 of wg 0, us per launch: prep %.2f | scan %.2f | decide %.2f | outputs %.2f | sum %.2f" % (d[16], d[17], d[18], d[19], d[16:20].sum()))
    print("wave 4 of wg 0, us per launch: prep %.2f | scan %.2f | decide %.2f | outputs %.2f | sum %.2f" % (d[22], d[23], d[24], d[25], d[22:26].sum()))
