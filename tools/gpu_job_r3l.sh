#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export CGIC_LIB=$GRAFT_REPO_ROOT/control-gic_amd/libcgic_hip_dbg.so
echo "== merge (latency mode, 4 bands)"; timeout 300 python tools/probe_merge.py 64 256 2>&1 | grep -v "amdgpu.ids\|Warning"
echo "== decode ss"; timeout 300 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|Warning"
import sys, os, ctypes
sys.path.insert(0, "tools"); sys.path.insert(0, ".")
import numpy as np, torch
import control_gic_amd as cg
from control_gic_amd import _lib
from bench import HotPath, make_inputs
dev = torch.device("cuda")
x, z, cb = make_inputs(64, 256, 256, 1000)
hp = HotPath(dev, x, z, cb, (0.1, 0.8))
hp.step(); torch.cuda.synchronize()
comp = hp.out[6]
l = _lib.lib(); l.cgic_debug_phase_clocks.argtypes = [ctypes.c_void_p]
for mode in ("throughput", "latency"):
    for _ in range(3): hp.codec.decompress(comp, decoder=mode)
    torch.cuda.synchronize()
    c = (ctypes.c_longlong * 32)(); l.cgic_debug_phase_clocks(c); c = list(c)
    if mode == "throughput":
        names = ["header + LUT issue", "stage + barrier", "first walk", "sweeps", "scan", "final walk + stores"]
        print("decode_image_kernel image 0:", " | ".join(f"{n} {(c[k+1]-c[k])/2.4e3:.2f}" for k, n in enumerate(names)), f"| total {(c[6]-c[0])/2.4e3:.2f} us; sweeps {c[9]}; chunks {c[23]}, max cw {c[25]}")
    print(mode, "merge_kernel block 0 (us): loads %.2f | bitsets+prefix %.2f | own-band prefix %.2f | scatter/gather %.2f | total %.2f" % (tuple((c[i+1]-c[i])/2.4e3 for i in range(10, 14)) + ((c[14]-c[10])/2.4e3,)))
PY
