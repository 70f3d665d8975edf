import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
dev = torch.device("cuda", 0)
name, which, q = sys.argv[1], sys.argv[2], int(sys.argv[3])
n = int(sys.argv[4]) if len(sys.argv) > 4 else 64
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
router = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)
i0 = int(os.environ.get("I0", "0"))
x = families(n=64)[name][i0:i0 + n]
z = rng.standard_normal((n, 4, 64, 64)).astype(np.float32)
xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
e8, e16 = cg.entropy_maps(xd, want_flat=os.environ.get("NOFLAT") is None)
torch.cuda.synchronize(); print("maps ok", flush=True)
_lib.REFINE_QUEUES = bool(q)
for rep in range(3):
    if which == "fused":
        r = vq_forward_route(zd, cb, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd)
    else:
        r = router(e16, e8, want_gate=False, pixels=xd)
    torch.cuda.synchronize(); print("rep", rep, "ok", flush=True)
if os.environ.get("CGIC_LIB", "").endswith("dbg.so"):
    import ctypes
    l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c)
    names = ["owner rounds", "helper rounds", "items in owner rounds", "items in helper rounds", "failed claims", "VQ wgs helping", "publications"]
    print({n: int(c[16 + i]) for i, n in enumerate(names)}, "(3 reps)")
    import time
    for _ in range(3):
        t0 = time.perf_counter()
        if which == "fused":
            vq_forward_route(zd, cb, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd)
        else:
            router(e16, e8, want_gate=False, pixels=xd)
        torch.cuda.synchronize(); print("wall us", round((time.perf_counter() - t0) * 1e6))
