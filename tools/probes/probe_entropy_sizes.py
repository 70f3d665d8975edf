"""GPU probe: entropy_maps launch time over batch / image shapes (graph-timed)"""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import control_gic_amd as cg
from tools_probe import graph_time
for B, H, W in [(64, 256, 256), (72, 256, 256), (80, 256, 256), (128, 256, 256), (8, 768, 768), (9, 768, 768), (8, 768, 1024), (32, 256, 768), (18, 512, 512), (16, 512, 512)]:
    x = (torch.rand(B, 3, H, W, device="cuda") * 2 - 1)
    b, m = graph_time(lambda: cg.entropy_maps(x))
    wgs = 0
    print(f"{B:4d} x {H}x{W}: {wgs:5d} workgroups, best {b:6.2f} us = {b * 1e3 / (B * H * W / 1e3):5.2f} ns/Kpixel")
