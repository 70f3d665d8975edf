"""dbg build: where one 768x768 tile's refinement spends its time in the fused VQ + router launch (row band 0 of the tile; MANY path):
router start | coarse done | medium done | end, and per select: entry | flat table | flat values | own members | exchange | others' values | final select.
usage: family index [split 0/1]"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
from oracle.content_families import families
dev = torch.device("cuda", 0)
fam, idx = sys.argv[1], int(sys.argv[2])
if len(sys.argv) > 3 and sys.argv[3] == "0": _lib.REFINE_QUEUES = False      # no scratch: the plain fused kernel, every row band evaluates everything
t = families(n=2, H=768, W=768, seed=11)
xd = torch.from_numpy(t[fam][idx:idx + 1]).to(dev)
rng = np.random.default_rng(0)
cb = torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)).to(dev)
zt = torch.from_numpy(np.random.default_rng(5).standard_normal((1, 4, 192, 192)).astype(np.float32)).to(dev)
e8, e16 = cg.entropy_maps(xd)
f = lambda: vq_forward_route(zt, cb, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=xd)
l = _lib.lib()
for _ in range(3): f()
torch.cuda.synchronize()
f()
torch.cuda.synchronize()
big = (ctypes.c_longlong * (2 * 4096))(); l.cgic_debug_block_times(big, 4096)
a = np.array(list(big), dtype=np.int64).reshape(-1)
rt = a[2 * 1024:2 * 1024 + 4 * 8].reshape(8, 4)
t0 = rt[:, 0].min()
print("router workgroups (start, coarse, medium, end) [us]:", " ".join("[" + " ".join(f"{(v - t0) / 100:.1f}" for v in r) + "]" for r in rt))
for q in range(2):
    s = a[4096 + 8 * q:4096 + 8 * q + 8]
    print("select", q, "band 0: entry", f"{(s[0] - t0) / 100:.1f}", "| +pass A, +flat values, +pass B and scan, +own members, +exchange, +others, +final select:",
          " ".join(f"{(s[k] - s[k - 1]) / 100:.1f}" if s[k] > s[k - 1] > 0 else "-" for k in range(1, 8)))
for q in range(2):
    s = a[4096 + 1024 + 8 * q:4096 + 1024 + 8 * q + 4]
    print("select", q, "front: entry | past the shortcut | counted | past the early returns:", " ".join(f"{(v - t0) / 100:.1f}" if 0 <= v - t0 < 10**6 else "-" for v in s))
