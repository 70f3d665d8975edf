"""dbg build: timeline of the fused VQ + router launch on a content family (router workgroups: start | coarse done | medium done | end;
VQ workgroups: end incl. helping), + refinement-queue counters.  usage: family [q]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route, prepare_codebook
from oracle.content_families import families
dev = torch.device("cuda", 0)
name = sys.argv[1]; _lib.REFINE_QUEUES = bool(int(sys.argv[2])) if len(sys.argv) > 2 else True
rng = np.random.default_rng(0)
x = families(n=64)[name] if name != "rand" else rng.random((64, 3, 256, 256)).astype(np.float32)
z = rng.standard_normal((64, 4, 64, 64)).astype(np.float32)
xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
w = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
prep = prepare_codebook(w)
e8, e16 = cg.entropy_maps(xd)
f = lambda: vq_forward_route(zd, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=xd)
l = _lib.lib()
l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
for _ in range(3): f()
torch.cuda.synchronize()
c0 = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c0)
f()
torch.cuda.synchronize()
c1 = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c1)
names = ["sum of helper 16x16 round times [10 ns]", "helper rounds", "max helper round time (cumulative max) [10ns]", "items in helper rounds", "failed claims", "VQ wgs helping", "publications"]
print({n: int(c1[20 + i] - c0[20 + i]) for i, n in enumerate(names)})
big = (ctypes.c_longlong * (2 * 4096))(); l.cgic_debug_block_times(big, 4096)
a = np.array(list(big), dtype=np.int64).reshape(4096, 2)
vq = a[:256]; rt = a[1024:1024 + 128].reshape(64, 4)
rq = a.reshape(-1)[4096:4096 + 8 * 128].reshape(128, 8)
t0g = {b: vq[np.arange(256) % 8 == b % 8][:, 0].min() for b in range(8)}
print("queues (image bank: publish, exhausted, done [us]; items, own):")
for q in range(128):
    t0 = t0g[(q >> 1) % 8]
    if 0 < rq[q, 3] <= 9216 and 0 < (rq[q, 0] - t0) < 100000 and rq[q, 2] > rq[q, 0]:
        print(f"  img {q >> 1} bank {q & 1}: {(rq[q,0]-t0)/100:.1f} {(rq[q,1]-t0)/100:.1f} {(rq[q,2]-t0)/100:.1f}  items {rq[q,3]} own {rq[q,4]}")
rows = []
for xcd in range(8):
    v = vq[np.arange(256) % 8 == xcd]
    t0 = v[:, 0].min()
    r = rt[np.arange(64) % 8 == xcd]
    vend = (v[:, 1] - t0) / 100.0
    rs = (r - t0) / 100.0
    print(f"XCD {xcd}: VQ wgs end med {np.median(vend):.1f} max {vend.max():.1f} | routers (start, coarse, medium, end):", " ".join(f"[{q[0]:.0f} {q[1]:.0f} {q[2]:.0f} {q[3]:.0f}]" for q in rs))
