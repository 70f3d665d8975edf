"""GPU probe: one 2040x1356 image through the tiling driver -- eager / hipGraph, shape groups sequential / on parallel streams"""
import sys, os, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import control_gic_amd as cg
from control_gic_amd import highres
from control_gic_amd.quantize import vq_forward_route
dev = torch.device("cuda", 0)
x1, z1, cb = bench.make_inputs(1, 256, 256, seed=1)
hp = bench.HotPath(dev, x1, z1, cb, (0.1, 0.8))
vq, codec = hp.vq, hp.codec
H, W = 1356, 2040
x = torch.from_numpy(np.random.default_rng(4).random((1, 3, H, W), dtype=np.float32)).to(dev)
zs = {}


def encode(tiles):
    T, _, th, tw = tiles.shape
    key = (T, th, tw)
    if key not in zs:
        zs[key] = torch.from_numpy(np.random.default_rng(th * 7 + tw).standard_normal((T, 4, th // 4, tw // 4), dtype=np.float32)).to(dev)
    e8, e16 = cg.entropy_maps(tiles)
    _, _, ind, mask, _, mode = vq_forward_route(zs[key], vq.embedding.weight, 0.25, True, e16, e8, 0.1, 0.8, per_image=True)
    return ind, mask, mode


def once(conc):
    t = highres.compress_tiled(x, encode, codec, concurrent=conc)
    p, st = highres.decompress_tiled(t, codec, concurrent=conc, check=False)
    return t, p, st


def timeit(f, n=40):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for conc in (False, True):
    once(conc); torch.cuda.synchronize()
    print(f"eager, concurrent={conc}: {timeit(lambda: once(conc)):.3f} ms per image")
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(side):
        with torch.cuda.graph(g, stream=side):
            once(conc)
    torch.cuda.current_stream().wait_stream(side)
    print(f"graph, concurrent={conc}: {timeit(g.replay):.3f} ms per image")
