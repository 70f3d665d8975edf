"""Round 6 A/B: the throughput decoder per image (shipping) against one workgroup per (image, stream) (CGIC_SS_PER_STREAM=1, read by the
library at its first decompress call): decode + merge alone, back to back (graph, HIP events), its resource time with four streams
replaying it, and the four-lane step.  Run twice: CGIC_SS_PER_STREAM=0 / 1 python tools/probes/probe_decode_ab.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import control_gic_amd as cg
import bench
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
e8, e16 = cg.entropy_maps(hp.x)
mask, _, _, mode = hp.router(e16, e8, want_gate=False)
_, _, ind = cg.quantize._vq_forward(hp.z, hp.vq.embedding.weight, 0.25, True, None)
comp = hp.codec.compress(ind, mask, mode)
ref = hp.codec.decompress(comp, decoder="latency")
got = hp.codec.decompress(comp, decoder="throughput")
torch.cuda.synchronize()
assert torch.equal(ref[0], got[0]) and torch.equal(ref[2], got[2]) and int(got[3].abs().max()) == 0
alone = bench.graph_kernel_time(lambda: hp.codec.decompress(comp, decoder="throughput"))
res = bench.saturated_stage_times(hp)["decode+merge"]
slots = [bench.make_inputs(64, 256, 256, 1000 + s) for s in range(8)]
xz = [(torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)) for a, b, _ in slots]
dt, _ = bench.lanes_rate(hp.vq, hp.codec, (0.1, 0.8), xz, 4, 400)
print(f"per_stream={os.environ.get('CGIC_SS_PER_STREAM', '0')}: decode+merge alone {alone:.2f} us | resource (4 streams) {res:.2f} us | four-lane step {dt * 1e6:.2f} us = {64 * 65536 / dt / 1e9:.1f} GPixel/s")
