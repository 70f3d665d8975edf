"""GPU probe: decompress (decode + merge) per decoder mode, alone and with four streams (resource time)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
import bench
from tools_probe import graph_time
from control_gic_amd.pipeline import distinct_queue_streams
dev = torch.device("cuda", 0)
hps = []
for s in range(4):
    x, z, cb = bench.make_inputs(64, 256, 256, seed=s)
    if s == 0:
        vq = bench.make_quantizer(dev, cb); codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
    hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8), vq=vq, codec=codec); hp.step(); hps.append(hp)
torch.cuda.synchronize()
streams = distinct_queue_streams(dev, 4)
N = 20
for mode in ("latency", "throughput"):
    with cg.decoder_mode(mode):
        comp = hps[0].out[6]
        print(mode, "alone: best %.2f mean %.2f us" % graph_time(lambda: codec.decompress(comp)))
        gs = []
        for j in range(4):
            c = hps[j].out[6]
            for _ in range(2): codec.decompress(c)
            torch.cuda.synchronize()
            with torch.cuda.stream(streams[j]):
                g = torch.cuda.CUDAGraph()
                with torch.cuda.graph(g, stream=streams[j]):
                    for _ in range(N): codec.decompress(c)
            gs.append(g)
        torch.cuda.synchronize()
        best = 1e9
        for _ in range(8):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for g, st in zip(gs, streams):
                with torch.cuda.stream(st): g.replay()
            torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
        print(mode, "4 lanes: %.2f us per launch" % (best * 1e6 / (4 * N)))
