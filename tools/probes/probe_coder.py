"""GPU probe: where does decode time go?  (dev tool, not part of the product or the tests)"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import time_events, zipf_freq

dev = torch.device("cuda")
class V:
    def __init__(s, v): s.v = v
    def item(s): return s.v
freq = zipf_freq()
h = cg.HuffmanCoding({str(i): V(float(freq[i])) for i in range(1024)})
rng = np.random.default_rng(0)
p = (freq + 1.0) / (freq + 1.0).sum()
l = _lib.lib()
for n in (64, 256, 1024, 4096, 8192, 36864):
    sym = torch.from_numpy(rng.choice(1024, n, p=p)).to(dev)
    cap = l.cgic_stream_capacity(h.table.handle, n)
    out = torch.zeros(cap, dtype=torch.uint8, device=dev); nb = torch.zeros(1, dtype=torch.int32, device=dev)
    wsb = l.cgic_stream_workspace_bytes(n); ws = torch.empty(max(wsb, 16), dtype=torch.uint8, device=dev)
    s = torch.cuda.current_stream().cuda_stream
    enc = lambda: _lib.call("cgic_encode_stream", h.table.handle, sym.data_ptr(), 8, n, out.data_ptr(), cap, nb.data_ptr(), ws.data_ptr(), s)
    enc(); torch.cuda.synchronize(); nbytes = int(nb.item())
    dsym = torch.empty(n + 8, dtype=torch.int64, device=dev); cnt = torch.zeros(1, dtype=torch.int64, device=dev)
    dec = lambda: _lib.call("cgic_decode_stream", h.table.handle, out.data_ptr(), nbytes, dsym.data_ptr(), n + 8, cnt.data_ptr(), s)
    dec(); torch.cuda.synchronize()
    assert int(cnt.item()) == n and torch.equal(dsym[:n], sym)
    te, td = time_events(enc, 50), time_events(dec, 50)
    print(f"n={n:6d} bytes={nbytes:6d} encode {te:8.1f} us  decode {td:8.1f} us  ({td*1e3/n:7.1f} ns/sym)")

# ---- decompress phase clocks (needs CGIC_LIB=.../libcgic_hip_dbg.so)
if os.environ.get("CGIC_LIB"):
    import ctypes
    from bench import HotPath, make_inputs
    x, z, cb = make_inputs(64, 256, 256, 1000)
    hp = HotPath(dev, x, z, cb, (0.1, 0.8))
    e8, e16, mask, mode, zq, ind, comp = hp.encode()
    for _ in range(3):
        hp.decode(comp)
    torch.cuda.synchronize()
    clk = (ctypes.c_longlong * 16)()
    l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
    l.cgic_debug_phase_clocks(clk)
    c = list(clk)
    names = ["start", "lut loaded", "passA begin", "passA end", "sync", "passB end", "passC end", "sync", "masks+prefix", "merge end"]
    for i in range(1, 10):
        print(f"  {names[i]:14s} +{(c[i]-c[i-1])/2.29e3:7.2f} us   (t={(c[i]-c[0])/2.29e3:7.2f})")
    print("  wave 3 (first medium-stream wave):", " ".join(f"{n}+{(c[8+i]-c[8+i-1])/2.29e3:.2f}" for i, n in zip(range(3, 8), ["passA", "sync", "passB", "passC", "sync"])), "us")
    print("decompress total (events):", time_events(lambda: hp.decode(comp), 50), "us; nbytes[0] =", comp.nbytes[0].tolist())
