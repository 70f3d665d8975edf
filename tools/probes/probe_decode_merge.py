"""dev: decompress (decoder + merge) as one launch vs two, per call, for small launches.  CGIC_NO_DECODE_MERGE=1 (dev knob of the `make dbg` build: CGIC_LIB=.../libcgic_hip_dbg.so) = two launches"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import control_gic_amd as cg
from control_gic_amd.quantize import vq_forward_route
dev = torch.device("cuda", 0)
cb = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
for B, H, W in ((1, 768, 768), (2, 768, 768), (4, 768, 768), (1, 512, 768)):
    rng = np.random.default_rng(B + H)
    x = torch.from_numpy(rng.random((B, 3, H, W), dtype=np.float32)).to(dev)
    z = torch.from_numpy(rng.standard_normal((B, 4, H // 4, W // 4), dtype=np.float32)).to(dev)
    e8, e16 = cg.entropy_maps(x)
    _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=x)
    comp = codec.compress(ind, mask, mode)
    out = codec.decompress(comp)
    torch.cuda.synchronize()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())

    def ten():
        for _ in range(10):
            r = codec.decompress(comp)
        return r
    g, r = cg.capture_graph(ten, side)
    torch.cuda.current_stream().wait_stream(side)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20):
        g.replay()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 200
    ok = torch.equal(r[0], out[0]) and torch.equal(r[2], out[2]) and int(r[3].abs().max()) == 0
    print(f"B={B} {H}x{W}: decompress {dt * 1e6:.2f} us per call (10 per graph), same={ok}, no_fuse={os.environ.get('CGIC_NO_DECODE_MERGE')}")
