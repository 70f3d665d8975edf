"""GPU probe: achieved HBM bandwidth of the decoder-side blend / avg-pool kernels and the encoder latent merge at the
reference decoder's shape (512 channels at the latent grid of 256x256 images, config_inference.yaml: ch 128 x mult 4)"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import control_gic_amd as cg
from bench import time_events
B, C, h, w = 32, 512, 64, 64
g = torch.Generator().manual_seed(1)
hf = torch.randn(B, C, h, w, generator=g).cuda(); own = torch.randn(B, C, h, w, generator=g).cuda()
hm = hf[:, :, :h // 2, :w // 2].contiguous(); ownm = own[:, :, :h // 2, :w // 2].contiguous()
e16 = torch.rand(B, h // 4, w // 4, generator=g).cuda(); e8 = torch.rand(B, h // 2, w // 2, generator=g).cuda()
mask, _, _, _ = cg.TripleGrainFixedEntropyRouter(0.1, 0.8, per_image=True)(e16, e8)
out = torch.empty_like(hf); outm = torch.empty_like(hm)
up2 = torch.nn.Upsample(scale_factor=2, mode="nearest"); up4 = torch.nn.Upsample(scale_factor=4, mode="nearest")
rows = [
    ("decoder_blend_fine   [%d,%d,%d,%d]" % (B, C, h, w), lambda: cg.decoder_blend_fine(hf, own, mask, out=out), 3 * hf.numel() * 4,
     lambda: hf * up4(mask[0].float()) + hf * up2(mask[1].float()) + own * mask[2]),
    ("decoder_blend_medium [%d,%d,%d,%d]" % (B, C, h // 2, w // 2), lambda: cg.decoder_blend_medium(hm, ownm, mask, out=outm), 3 * hm.numel() * 4,
     lambda: hm * up2(mask[0].float()) + ownm * mask[1]),
    ("avg_pool 4           [%d,%d,%d,%d]" % (B, C, h, w), lambda: cg.avg_pool(hf, 4), hf.numel() * 4 * (1 + 1 / 16), lambda: torch.nn.functional.avg_pool2d(hf, 4)),
    ("avg_pool 2           [%d,%d,%d,%d]" % (B, C, h, w), lambda: cg.avg_pool(hf, 2), hf.numel() * 4 * (1 + 1 / 4), lambda: torch.nn.functional.avg_pool2d(hf, 2)),
]
for name, f, nbytes, ref in rows:
    t = time_events(f, 30); tr = time_events(ref, 10)
    print(f"{name}: {t:8.1f} us  {nbytes / t / 1e6:6.2f} TB/s algorithmic ({nbytes / 1e6:.0f} MB) | stock torch expression {tr:8.1f} us")
