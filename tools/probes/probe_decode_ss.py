"""GPU probe: phase clocks of decode_image_kernel, image 0 (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import HotPath, make_inputs
from tools_probe import graph_time
dev = torch.device("cuda")
B_, S_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
x, z, cb = make_inputs(B_, S_, S_, 1000)
hp = HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
comp = hp.out[6]
for _ in range(3): hp.codec.decompress(comp)
torch.cuda.synchronize()
l = _lib.lib()
if hasattr(l, "cgic_debug_phase_clocks"):
    l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
    names = ["header + LUT issue", "stage + barrier", "first walk", "sweeps", "scan", "final walk + stores"]
    for k, n in enumerate(names):
        print(f"   {n:22s} +{(c[k+1]-c[k])/2.29e3:6.2f} us")
    print(f"   chunks {c[23]} R {c[24]} max symbols in a chunk {c[25]}")
    print(f"   total {(c[6]-c[0])/2.29e3:6.2f} us; sweeps {c[9]}; nbytes image 0: {comp.nbytes[0].tolist()}")
print("decompress (decode + merge) graph-timed: best %.2f mean %.2f us" % graph_time(lambda: hp.codec.decompress(comp)))
