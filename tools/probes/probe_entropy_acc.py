"""GPU probe: entropy_maps accuracy against the oracle on several image families + graph-timed launch"""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
from oracle import cgic_oracle as orc
from tools_probe import graph_time

g = torch.Generator().manual_seed(5)
fam = {}
fam["uniform"] = torch.rand(2, 3, 256, 256, generator=g)
yy, xx = torch.meshgrid(torch.linspace(0, 1, 256), torch.linspace(0, 1, 256), indexing="ij")
sm = (0.5 + 0.5 * torch.sin(6 * xx + 2 * yy))[None, None].repeat(2, 3, 1, 1)
fam["smooth"] = (sm + 0.01 * torch.randn(2, 3, 256, 256, generator=g)).clamp(0, 1)
fam["u8"] = torch.randint(0, 256, (2, 3, 256, 256), generator=g).float() / 255
fam["const"] = torch.full((1, 3, 64, 64), 0.5)
fam["flat+noise"] = (0.37 + 0.002 * torch.randn(1, 3, 128, 128, generator=g))
fam["signed"] = torch.rand(1, 3, 64, 64, generator=g) * 2 - 1
fam["out_of_range"] = torch.rand(1, 3, 64, 64, generator=g) * 6 - 3
blocks = torch.rand(1, 3, 16, 16, generator=g).repeat_interleave(16, 2).repeat_interleave(16, 3)
fam["blocks16"] = blocks
for name, x in fam.items():
    e8, e16 = cg.entropy_maps(x.cuda())
    o8, o16 = orc.entropy(x.numpy(), 8), orc.entropy(x.numpy(), 16)
    d8 = np.abs(e8.cpu().numpy() - o8); d16 = np.abs(e16.cpu().numpy() - o16)
    print(f"{name:14s} e8 max {d8.max():.2e} mean {d8.mean():.2e} | e16 max {d16.max():.2e} mean {d16.mean():.2e} | range [{o8.min():.4f}, {o8.max():.4f}]")
xn = torch.rand(1, 3, 32, 32, generator=g); xn[0, 1, 3, 5] = float("nan"); xn[0, 0, 20, 30] = float("nan")
e8, e16 = cg.entropy_maps(xn.cuda())
print("nan e8 at", torch.isnan(e8).nonzero().tolist(), "e16 at", torch.isnan(e16).nonzero().tolist())
for B, H, W in [(64, 256, 256), (8, 768, 768), (32, 768, 768), (1, 256, 256)]:
    x = torch.rand(B, 3, H, W, device="cuda")
    b, m = graph_time(lambda: cg.entropy_maps(x))
    print(f"{B:4d} x {H}x{W}: best {b:6.2f} mean {m:6.2f} us = {B*3*H*W*4/b/1e6:5.2f} TB/s")
