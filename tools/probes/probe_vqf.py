"""GPU probe: filter-path VQ kernel vs the exact fp32-MFMA loop (CGIC_VQ_EXACT=1), event-timed; equality of all outputs"""
import os, sys, subprocess
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import control_gic_amd as cg
    from control_gic_amd.quantize import _vq_forward
    from bench import time_events
    g = torch.Generator().manual_seed(0)
    out = {}
    for B in (64, 8, 1):
        z = torch.randn(B, 4, 64, 64, generator=g).cuda(); w = torch.randn(1024, 4, generator=g).cuda()
        full = lambda: _vq_forward(z, w, 0.25, True, None)
        idx_only = lambda: _vq_forward(z, w, 0.25, True, None, False, False)
        zq, loss, idx = full()
        print(f"B={B}: full {time_events(full, 100):.1f} us, indices only {time_events(idx_only, 100):.1f} us", flush=True)
        out[f"idx{B}"] = idx.cpu().numpy(); out[f"zq{B}"] = zq.cpu().numpy(); out[f"loss{B}"] = loss.cpu().numpy()
    np.savez(sys.argv[2], **out)
else:
    res = []
    for exact in ("0", "1"):
        env = dict(os.environ, CGIC_VQ_EXACT=exact)
        path = f"/tmp/vqf_{exact}.npz"
        print("CGIC_VQ_EXACT=" + exact, flush=True)
        subprocess.run([sys.executable, __file__, "child", path], env=env, check=True)
        res.append(np.load(path))
    for k in res[0].files:
        same = np.array_equal(res[0][k], res[1][k])
        print(k, "identical" if same else f"DIFFERENT ({(res[0][k] != res[1][k]).sum()} elements)")
