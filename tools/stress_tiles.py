"""GPU stress: the tiling driver as ONE launch chain (launch groups, cgic_entropy_maps_tiles / cgic_cut_tiles, decode_merge_kernel) ==
the shape groups launched one by one, over random image sizes, batch sizes, content families (tie-heavy ones included: the router's
refinement runs inside the grouped launch), ratios (all routing modes), fp32 images and uint8 frames, both decoders.
Not a pytest (minutes); run by hand:  python tools/stress_tiles.py [seed] [seconds]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import control_gic_amd as cg
from control_gic_amd import highres
from control_gic_amd.quantize import vq_forward_route, prepare_codebook
from oracle.content_families import families
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
budget = float(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
vq = cg.VectorQuantizer(1024, 4, beta=0.25).to(dev)
with torch.no_grad():
    vq.embedding.weight.copy_(torch.from_numpy(rng.standard_normal((1024, 4)).astype(np.float32)))
vq.usage_counter.copy_(torch.from_numpy(rng.integers(1, 1000, 1024).astype(np.float32)))
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight.detach())
prepared = prepare_codebook(vq.embedding.weight.detach())
RATIOS = [(0.1, 0.8), (0.1, 0.4), (0.7, 0.3), (0.3, 0.7), (0.0, 0.4), (0.4, 0.0), (1.0, 0.0), (0.0, 1.0), (0.0, 0.0), (0.5, 0.5)]
FAM = ["noise", "noise8", "smooth8", "flat_edges", "blocky8"]
t0 = time.time(); n = 0; pixels = 0
while time.time() - t0 < budget:
    H = int(rng.choice([rng.integers(16, 400), rng.integers(400, 1200), rng.integers(1200, 2200)]))
    W = int(rng.choice([rng.integers(16, 400), rng.integers(400, 1200), rng.integers(1200, 2200)]))
    N = int(rng.integers(1, 4)) if H * W < 1500 * 1500 else 1
    fam = FAM[int(rng.integers(0, len(FAM)))]
    frames = bool(rng.integers(0, 2)) and fam != "noise"
    cr, mr = RATIOS[int(rng.integers(0, len(RATIOS)))]
    Hp, Wp = -(-H // 16) * 16 + 16, -(-W // 16) * 16 + 16
    if fam == "noise":
        xn = rng.random((N, 3, H, W), dtype=np.float32)
    else:
        xn = families(N, -(-Hp // 8) * 8, -(-Wp // 8) * 8, seed=int(rng.integers(0, 1 << 30)))[fam][:, :, :H, :W]
    x = torch.from_numpy(np.ascontiguousarray(xn)).to(dev)
    inp = (x.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous() if frames else x

    def latent(tiles_f32):
        z = torch.nn.functional.avg_pool2d(tiles_f32, 4)
        return (torch.cat([z, z[:, :1] * 2 - 1], dim=1) * 3 - 1.5).contiguous()

    def encode(tiles):
        if tiles.dtype == torch.uint8:
            z = latent(tiles.permute(0, 3, 1, 2).float() / 255)
            _, e8, e16 = cg.entropy_maps_u8(tiles, want_x=False)
        else:
            z = latent(tiles)
            e8, e16 = cg.entropy_maps(tiles)
        _, _, ind, mask, _, mode = vq_forward_route(z, vq.embedding.weight, 0.25, True, e16, e8, cr, mr, per_image=True, pixels=tiles, prepared=prepared)
        return ind, mask, mode

    ref = highres.compress_tiled_batch(inp, encode, codec)
    got = highres.compress_tiled_batch(inp, encode, codec, chain=True)
    for a, b in zip(ref, got):
        assert a.streams() == b.streams(), ("streams", H, W, N, fam, frames, cr, mr)
    for mode in ("latency", "throughput"):
        with cg.decoder_mode(mode):
            dref = highres.decompress_tiled_batch(ref, codec)
            dgot = highres.decompress_tiled_batch(got, codec, chain=True)
        for pa, pb in zip(dref, dgot):
            for (i0, m0, z0), (i1, m1, z1) in zip(pa, pb):
                assert torch.equal(i0, i1) and torch.equal(z0, z1) and all(torch.equal(p, q) for p, q in zip(m0, m1)), ("decode", mode, H, W, N, fam, frames, cr, mr)
    n += 1; pixels += N * H * W
print(f"stress_tiles: {n} cases ({pixels / 1e6:.0f} MPixel) in {time.time() - t0:.0f} s: chain == groups one by one everywhere")
