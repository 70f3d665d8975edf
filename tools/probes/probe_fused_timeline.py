"""GPU probe (dbg build): the fused VQ + router launch at B=64 x 256x256 -- when the router workgroups start, pass their selects and end,
next to the VQ workgroups of the same XCD (the XCDs' clocks are offset against each other: only same-XCD differences mean anything)."""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench, control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route, prepare_codebook
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1)
xd, zd = torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)
w = torch.from_numpy(cb).to(dev)
prep = prepare_codebook(w)
e8, e16 = cg.entropy_maps(xd)
px = None if "plain" in sys.argv else xd
f = lambda: vq_forward_route(zd, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=px)
for _ in range(4): f()
torch.cuda.synchronize()
l = _lib.lib()
big = (ctypes.c_longlong * (2 * 4096))(); l.cgic_debug_block_times(big, 4096)
a = np.array(list(big), dtype=np.int64).reshape(4096, 2)
vq = a[:256]; rt = a[1024:1024 + 128].reshape(64, 4)
for xcd in range(8):
    v = vq[np.arange(256) % 8 == xcd]
    t0 = v[:, 0].min()
    r = rt[np.arange(64) % 8 == xcd]            # router block i sits at grid index 256 + i: XCD (256 + i) % 8 = i % 8
    vend = (v[:, 1] - t0) / 100.0
    rs = (r - t0) / 100.0
    print(f"XCD {xcd}: VQ wgs end med {np.median(vend):.1f} max {vend.max():.1f} | routers: start {np.median(rs[:,0]):.1f}  coarse done {np.median(rs[:,1]):.1f}  medium done {np.median(rs[:,2]):.1f}  end med {np.median(rs[:,3]):.1f} max {rs[:,3].max():.1f}")
