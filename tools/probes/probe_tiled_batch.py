"""kernel times of highres.compress_tiled_batch + decompress_tiled_batch on 8 images of 2040x1356 (run under rocprofv3 --kernel-trace --stats)"""
import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import control_gic_amd as cg
from control_gic_amd import highres
from control_gic_amd.quantize import vq_forward_route
dev = torch.device("cuda", 0)
rng = np.random.default_rng(0)
vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev)
with torch.no_grad():
    vq.embedding.weight.copy_(torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)))
vq.usage_counter.copy_(torch.from_numpy(rng.integers(1, 1000, 1024).astype(np.float32)))
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight.detach())
N, H, W = int(sys.argv[1]) if len(sys.argv) > 1 else 8, 1356, 2040
xs = torch.from_numpy(rng.random((N, 3, H, W), dtype=np.float32)).to(dev)
zs = {}
def encode(tiles):
    T, _, th, tw = tiles.shape
    if (T, th, tw) not in zs:
        zs[(T, th, tw)] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
    e8, e16 = cg.entropy_maps(tiles)
    _, _, ind, mask, _, mode = vq_forward_route(zs[(T, th, tw)], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True)
    return ind, mask, mode
def once():
    ts = highres.compress_tiled_batch(xs, encode, codec)
    return highres.decompress_tiled_batch(ts, codec, check=False)
for _ in range(3): once()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10): once()
torch.cuda.synchronize()
print("eager ms per batch", (time.perf_counter() - t0) / 10 * 1e3)
