#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -m gpu -k "uint8 or entropy" 2>&1 | tail -4
timeout 300 python - <<'PY'
import torch, bench
import control_gic_amd as cg
dev = torch.device("cuda", 0)
x, z, cb = bench.make_inputs(64, 256, 256, 1000)
hp = bench.HotPath(dev, x, z, cb, (0.1, 0.8))
frames = (hp.x.permute(0, 2, 3, 1) * 255).round().to(torch.uint8).contiguous()
xf = frames.permute(0, 3, 1, 2).float().div(255).contiguous()
print("fused u8 (x + maps):", round(bench.graph_kernel_time(lambda: cg.entropy_maps_u8(frames)), 2))
print("fused u8 (maps only):", round(bench.graph_kernel_time(lambda: cg.entropy_maps_u8(frames, want_x=False)), 2))
print("torch ToTensor on the GPU:", round(bench.graph_kernel_time(lambda: frames.permute(0, 3, 1, 2).to(torch.float32).div(255).contiguous()), 2))
print("entropy_maps fp32:", round(bench.graph_kernel_time(lambda: cg.entropy_maps(xf)), 2))
PY
