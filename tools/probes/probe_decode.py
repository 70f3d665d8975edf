"""GPU probe: phase clocks of decode_streams_kernel, medium-stream workgroup (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import HotPath, make_inputs
dev = torch.device("cuda")
B_, S_ = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (64, 256)
x, z, cb = make_inputs(B_, S_, S_, 1000)
hp = HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
comp = hp.out[6]
for _ in range(3): hp.codec.decompress(comp)
torch.cuda.synchronize()
l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
names = ["LUT + header loads", "setup", "pass A (functions)", "barrier", "pass B", "pass C (decode)", "barrier", "tail"]
for k, n in enumerate(names):
    print(f"   {n:22s} +{(c[k+1]-c[k])/2.29e3:6.2f} us")
print(f"   total {(c[8]-c[0])/2.29e3:6.2f} us; nbytes image 0: {comp.nbytes[0].tolist()}")
print("   pass A detail (wave 0): fill %.2f | lookups %.2f | doubling %.2f | compose %.2f | whole 2nd group %.2f us; dbl_rounds?" % tuple((c[b]-c[a])/2.29e3 for a, b in ((2,16),(16,17),(17,18),(18,19),(19,20))))
