"""GPU probe: does the short timed window (K=20) depend on what the GPU did just before (clock ramp)?"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
dev = torch.device("cuda", 0)
slots_np = [bench.make_inputs(64, 256, 256, seed=1000 + s) for s in range(8)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
slots = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
hist = torch.zeros(1024, dtype=torch.int64, device=dev)
ls = cg.pipeline.LaneStream(vq, 0.1, 0.8, slots, lanes=4, frequency=codec.huffman, hist=hist, quick_start=False)
ls.capture(); ls.prepare(5); ls.prepare(20); ls.prepare(200)
def timed(K):
    torch.cuda.synchronize(); t0 = time.perf_counter(); ls.submit(K); ls.join(); torch.cuda.synchronize(); return (time.perf_counter() - t0) * 1e6 / K
for idle_ms, warm in ((300, 5), (50, 5), (0, 5), (300, 200), (0, 200), (300, 1000), (0, 2000)):
    res = []
    for rep in range(5):
        ls.prepare(warm)
        time.sleep(idle_ms / 1e3) if idle_ms else None
        ls.submit(warm); ls.join(); torch.cuda.synchronize()
        res.append(timed(20))
    res.sort()
    print(f"idle {idle_ms} ms, then {warm} warm-up steps, then K=20: min {res[0]:.2f} median {res[2]:.2f} max {res[-1]:.2f} us/step", flush=True)
print("back-to-back K=20 windows:", " ".join(f"{timed(20):.1f}" for _ in range(12)))
