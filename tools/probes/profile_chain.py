"""dev: the 2040x1356 tiling driver as one launch chain -- cProfile of the eager call (host side) or a loop for rocprofv3 --kernel-trace.
usage: profile_chain.py host | loop [n] [nochain]"""
import cProfile, os, pstats, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
import bench
import control_gic_amd as cg
from control_gic_amd import highres
from control_gic_amd.quantize import vq_forward_route
dev = torch.device("cuda", 0)
cb = np.random.default_rng(12345).standard_normal((1024, 4), dtype=np.float32)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
H, W = 1356, 2040
x = torch.from_numpy(np.random.default_rng(4).random((1, 3, H, W), dtype=np.float32)).to(dev)
zs = {}


def encode(tiles):
    T, _, th, tw = tiles.shape
    key = (T, th, tw)
    if key not in zs:
        zs[key] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
    e8, e16 = cg.entropy_maps(tiles)
    _, _, ind, mask, _, mode = vq_forward_route(zs[key], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True, pixels=tiles)
    return ind, mask, mode


CHAIN = "nochain" not in sys.argv


def once(check=False):
    t = highres.compress_tiled(x, encode, codec, chain=CHAIN)
    return t, highres.decompress_tiled(t, codec, check=check, chain=CHAIN)


once(); torch.cuda.synchronize()
if sys.argv[1] == "host":
    pr = cProfile.Profile()
    pr.enable()
    for _ in range(50):
        once()
    pr.disable()
    torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
    for _ in range(n):
        once()
    torch.cuda.synchronize()
