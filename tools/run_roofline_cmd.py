"""The roofline command of bench.py's dominant kernel, on its own (for `rocprofv3 --kernel-trace --stats -- python tools/run_roofline_cmd.py`):
the fused VQ + router launch of the timed step (B=64 x 256x256, prepared codebook image), 20 launches captured back to back in one
hipGraph, replayed 6 times on the stream it was captured on, HIP events around the last 5 replays -- exactly bench.graph_kernel_time.
Prints the HIP-event average per launch; the profiler's average duration of vq_filter_router_kernel over the same launches is the
figure profiles/r03_roofline.json stores next to it."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
from control_gic_amd.quantize import vq_forward_route, _vq_forward
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
e8, e16 = cg.entropy_maps(hp.x)
w, prep = hp.vq.embedding.weight, hp.pipe.prepared
which = sys.argv[1] if len(sys.argv) > 1 else "fused"
if which == "fused":
    fn = lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=hp.x)
elif which == "fusedplain":        # ... without the pixels: no refinement (what the band evaluation of the batch's one or two images costs the launch)
    fn = lambda: vq_forward_route(hp.z, w, 0.25, True, e16, e8, 0.1, 0.8, prepared=prep, pixels=None)
elif which == "router":            # the stand-alone router launch on the same batch (refinement from the pixels)
    fn = lambda: hp.router(e16, e8, want_gate=False, pixels=hp.x)
elif which == "router768":         # ... on eight 768x768 tiles (eight row bands each)
    import numpy as np
    xt = torch.from_numpy(np.random.default_rng(3).random((8, 3, 768, 768), dtype=np.float32)).to(dev)
    t8, t16 = cg.entropy_maps(xt)
    fn = lambda: hp.router(t16, t8, want_gate=False, pixels=xt)
else:
    fn = lambda: _vq_forward(hp.z, w, 0.25, True, None, prepared=prep)
t = bench.graph_kernel_time(fn)
print(f"{which}: HIP events, 20 launches per graph x 5 replays: {t:.2f} us per launch", flush=True)
