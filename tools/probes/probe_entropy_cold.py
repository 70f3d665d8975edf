"""GPU probe: the entropy kernel over a rotation of batches that exceeds the Infinity Cache (every image byte from HBM), with and
without the flat8 by-product"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, bench, control_gic_amd as cg
xs = [torch.from_numpy(bench.make_inputs(64, 256, 256, 10 + s)[0]).cuda() for s in range(8)]
for flat in (True, False, True, False):
    def f():
        for x in xs:
            cg.entropy_maps(x, want_flat=flat)
    t = bench.graph_kernel_time(f, per_graph=4, reps=5) / len(xs)
    print(f"want_flat={flat}: {t:.2f} us per launch (8 distinct batches in rotation)")
