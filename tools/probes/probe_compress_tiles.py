"""GPU probe: workgroup start / end times of compress_streams_kernel on 768x768 tiles (CGIC_LIB=.../libcgic_hip_dbg.so)"""
import sys, os, ctypes
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import control_gic_amd as cg
from control_gic_amd import _lib
from control_gic_amd.quantize import vq_forward_route
import bench
dev = torch.device("cuda")
cb = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
rng = np.random.default_rng(3)
x = torch.from_numpy(rng.random((B, 3, 768, 768), dtype=np.float32)).to(dev)
z = torch.from_numpy(rng.standard_normal((B, 4, 192, 192), dtype=np.float32)).to(dev)
e8, e16 = cg.entropy_maps(x)
_, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=x)
for _ in range(3):
    comp = codec.compress(ind, mask, mode)
torch.cuda.synchronize()
l = _lib.lib(); n = 16 * B
buf = (ctypes.c_longlong * (2 * n))(); l.cgic_debug_block_times(buf, n)
t = np.array(list(buf), dtype=np.int64).reshape(n, 2)
live = t[:, 1] > 0
t0 = t[live, 0].min()
for i in range(n):
    if live[i]:
        print(f"image {i // 16} job {i % 16:2d}: start {(t[i, 0] - t0) / 100.0:6.2f}  end {(t[i, 1] - t0) / 100.0:6.2f}  dur {(t[i, 1] - t[i, 0]) / 100.0:6.2f} us")
print("nbytes", comp.nbytes[0].tolist())
