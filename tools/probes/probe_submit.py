"""GPU probe: host cost of BatchStream.submit() vs device time per step"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
import bench
dev = torch.device("cuda", 0)
slots_np = [bench.make_inputs(64, 256, 256, seed=s) for s in range(6)]
cb = slots_np[0][2]
vq = bench.make_quantizer(dev, cb)
codec = cg.GrainCodec(vq.embedding_counter, vq.embedding.weight)
hist = torch.zeros(1024, dtype=torch.int64, device=dev)
slots = [(torch.from_numpy(x).to(dev), torch.from_numpy(z).to(dev)) for x, z, _ in slots_np]
bs = cg.pipeline.BatchStream(vq, 0.1, 0.8, slots, frequency=codec.huffman, hist=hist)
bs.capture()
bs.submit(20); bs.join(); torch.cuda.synchronize()
for n in (50, 200):
    t0 = time.perf_counter(); bs.submit(n); t1 = time.perf_counter(); bs.join(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"pipelined n={n}: host submit {1e6*(t1-t0)/n:.1f} us/step, total {1e6*(t2-t0)/n:.1f} us/step")
# components
s = bs.slots[0]
t0 = time.perf_counter()
for _ in range(200): s.g_enc.replay()
t1 = time.perf_counter(); torch.cuda.synchronize()
print(f"g_enc.replay host {1e6*(t1-t0)/200:.1f} us; device-bound total {1e6*(time.perf_counter()-t0)/200:.1f}")
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(1000): ev.record()
t1 = time.perf_counter()
for _ in range(1000): torch.cuda.current_stream().wait_event(ev)
t2 = time.perf_counter()
print(f"event record {1e6*(t1-t0)/1000:.2f} us, wait_event {1e6*(t2-t1)/1000:.2f} us")
torch.cuda.synchronize()
# enc-only and dec-only device rates
t0 = time.perf_counter()
for i in range(200): bs.slots[i % 6].g_enc.replay()
torch.cuda.synchronize(); print(f"enc graphs back to back: {1e6*(time.perf_counter()-t0)/200:.1f} us/step")
t0 = time.perf_counter()
for i in range(200): bs.slots[i % 6].g_dec.replay()
torch.cuda.synchronize(); print(f"dec graphs back to back: {1e6*(time.perf_counter()-t0)/200:.1f} us/step")
