"""same-box A/B of the fused launch and the VQ-only launch (prepared codebook, fp32 noise): CGIC_LIB selects the library; BASE=1: a
pre-ABI-7 library (no refinement scratch)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from control_gic_amd import _lib
if os.environ.get("BASE"):
    for _n in ("cgic_router_refine_scratch_bytes", "cgic_compress_image", "cgic_compress_tiled"):
        _lib.PROTOTYPES.pop(_n, None)
    _lib.REFINE_QUEUES = False
if os.environ.get("NOQ"):
    _lib.REFINE_QUEUES = False
import torch, control_gic_amd as cg, bench
from control_gic_amd.quantize import vq_forward_route, _vq_forward
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
e8, e16 = cg.entropy_maps(hp.x)
w, prep = hp.vq.embedding.weight, hp.pipe.prepared
r = {}
r["fused"] = bench.graph_kernel_time(lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=hp.x))
r["fused_nopixels"] = bench.graph_kernel_time(lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=None))
r["vq"] = bench.graph_kernel_time(lambda: _vq_forward(hp.z, w, 0.25, True, None, prepared=prep))
print({k: round(v, 2) for k, v in r.items()}, flush=True)
