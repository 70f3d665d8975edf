"""GPU probe: the throughput decoder (decode_image_kernel) on streams that never re-synchronise (one 13-bit codeword repeated over the
whole fine grid, Zipf table) against the benchmark's streams: decode + merge time per batch of 64, and the sweep counters"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch, bench, control_gic_amd as cg
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1)
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
freq = bench.zipf_freq()
long_sym = int(np.argmin(freq))
e16 = torch.rand(64, 16, 16, device=dev) * 2.6; e8 = torch.rand(64, 32, 32, device=dev) * 2.6
lib = cg._lib.lib()
for name, ind, ratio in (("one 13-bit codeword repeated, all-fine grid", np.full((64, 64, 64), long_sym), (0.0, 0.0)),
                         ("uniform random symbols, all-fine grid", np.random.default_rng(1).integers(0, 1024, (64, 64, 64)), (0.0, 0.0)),
                         ("uniform random symbols, ratio (0.1,0.8,0.1)", np.random.default_rng(1).integers(0, 1024, (64, 64, 64)), (0.1, 0.8))):
    mask, _, _, mode = cg.TripleGrainFixedEntropyRouter(*ratio, per_image=True)(e16, e8)
    indt = torch.from_numpy(ind.astype(np.int64)).to(dev)
    comp = codec.compress(indt, mask, mode)
    cnt = torch.zeros(4, dtype=torch.int32, device=dev)
    lib.cgic_decode_stats(cnt.data_ptr())
    d = codec.decompress(comp, decoder="throughput"); torch.cuda.synchronize()
    lib.cgic_decode_stats(None)
    ok = int(d[3].abs().max()) == 0 and bool(torch.equal(torch.where(mask[2][:, 0] == 1, d[0], indt), indt) if ratio == (0.0, 0.0) else True)
    t = bench.graph_kernel_time(lambda: codec.decompress(comp, decoder="throughput"), per_graph=5, reps=3)
    t2 = bench.graph_kernel_time(lambda: codec.decompress(comp, decoder="latency"), per_graph=5, reps=3)
    c = cnt.cpu().numpy()
    print(f"{name}: throughput decoder {t:.1f} us (passes per image: mean {c[0] / max(c[1], 1):.1f} max {c[2]}), latency decoder {t2:.1f} us, ok {ok}")
