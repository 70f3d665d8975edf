"""GPU probe: phase clocks of router / compress / merge kernels (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import HotPath, make_inputs, time_events
dev = torch.device("cuda")
B_, S_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
x, z, cb = make_inputs(B_, S_, S_, 1000)
hp = HotPath(dev, x, z, cb, (0.1, 0.8))
e8, e16, mask, mode, zq, ind, comp = hp.encode()
torch.cuda.synchronize()
l = _lib.lib()
l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
def clocks():
    torch.cuda.synchronize()
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); return list(c)
def show(title, c, idx, names):
    print(title)
    for a, b, n in zip(idx[:-1], idx[1:], names):
        print(f"   {n:28s} +{(c[b]-c[a])/2.29e3:6.2f} us")
    print(f"   total {(c[idx[-1]]-c[idx[0]])/2.29e3:6.2f} us")
for _ in range(3): hp.router(e16, e8, want_gate=False)
c = clocks(); show("router (image 0)", c, [0,1,2,3,4,5,6], ["stage e16/e8 to LDS", "select coarse thr", "coarse gate + mask_c", "select medium thr", "mask_m", "mask_f"])
print("   events: %.1f us" % time_events(lambda: hp.router(e16, e8, want_gate=False), 50))
for _ in range(3): hp.codec.compress(ind, mask, mode)
c = clocks(); show("compress (image 0, fine stream)", c, [0,1,2,3], ["table->LDS, setup", "phase A (select+scan)", "phase B (gather bits)"])
print("   events: %.1f us" % time_events(lambda: hp.codec.compress(ind, mask, mode), 50))
for _ in range(3): hp.decode(comp)
c = clocks(); show("merge (image 0, band 0)", c, [10,11,12,13,14], ["staging loads", "bitsets + prefixes", "fine base count", "scatter/merge/gather"])
print("   events (decode+merge): %.1f us" % time_events(lambda: hp.decode(comp), 50))
print("compress without hist: %.1f us; with hist: %.1f us" % (time_events(lambda: hp.codec.compress(ind, mask, mode), 50), time_events(lambda: hp.codec.compress(ind, mask, mode, hist=hp.hist), 50)))
l.cgic_debug_reset_span(); hp.codec.compress(ind, mask, mode); c = clocks()
print("compress launch span over all workgroups: %.2f us; last-ending workgroup ran %.2f us, id (stream + 1000*image) = %d" % ((c[29]-c[28])/100.0, c[30]/100.0, c[31]))
