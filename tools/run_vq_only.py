"""GPU driver for counter passes: only the VQ kernel (B=64 x 256x256 latents, z_q + loss), N launches.
usage: rocprofv3 --pmc ... --kernel-trace -d DIR -o pmc -- python tools/run_vq_only.py [iters] [B] [size]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from control_gic_amd.quantize import _vq_forward
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
S = int(sys.argv[3]) if len(sys.argv) > 3 else 256
g = torch.Generator().manual_seed(0)
z = torch.randn(B, 4, S // 4, S // 4, generator=g).cuda()
w = torch.randn(1024, 4, generator=g).cuda()
for _ in range(iters):
    _vq_forward(z, w, 0.25, True, None)
torch.cuda.synchronize()
print("done", iters)
