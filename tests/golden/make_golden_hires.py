#!/usr/bin/env python3
"""Golden vectors for the high-resolution (tiled) path -- BASELINE config 4 in miniature.

Runs the REAL reference (inference_high_resolution.py helpers + CGIC.compress on CPU) on a synthetic
800x1040 image: zero-pad to x16, non-overlapping 768-px grid (tiles 768x768, 768x272, 32x768, 32x272),
per-tile compress, bpp accounting.  Build container only (needs /root/reference); the fixture holds
data: per-tile latents, entropy maps, masks, indices, the .bin bytes and the bpp numbers.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hires.py            # 800 x 1040
    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_hires.py 1356 2040  # the DIV2K-typical size of SURVEY.md 8a-H / 8d:
                                                                                  # pad to 1360 x 2048, tiles 768/592 x 768/768/512
"""
import io
import os
import sys
import tempfile
import types
from unittest.mock import MagicMock

import numpy as np
import torch
import yaml

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.dont_write_bytecode = True
sys.path.insert(0, REF)
sys.path.insert(0, ROOT)
pl = types.ModuleType("pytorch_lightning")
pl.LightningModule = torch.nn.Module
pl.LightningDataModule = object
sys.modules["pytorch_lightning"] = pl
for name in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils",
             "omegaconf", "PIL", "PIL.Image"):
    sys.modules.setdefault(name, MagicMock())
torch.nn.Module.cuda = lambda self, device=None: self

import inference_high_resolution as hr  # noqa: E402  (module-level code only defines functions/classes)
from CGIC.models.model import CGIC  # noqa: E402
from CGIC.tools.indices_coding import HuffmanCoding  # noqa: E402
from CGIC.tools.mask_coding import BinaryCoding  # noqa: E402
from oracle import cgic_oracle as orc  # noqa: E402
from make_golden import freq_tables  # noqa: E402

torch.set_num_threads(8)
H, W = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (800, 1040)
OUT = f"hires_{H}x{W}.npz"
RATIO = (0.1, 0.8)

params = yaml.safe_load(open(os.path.join(REF, "configs/config_inference.yaml")))["model"]["params"]
params["ckpt_path"] = None
params["lossconfig"] = None
params["ddconfig"]["router_config"]["params"] = {"coarse_grain_ratio": RATIO[0], "medium_grain_ratio": RATIO[1]}
torch.manual_seed(0)
_so = sys.stdout
sys.stdout = io.StringIO()
try:
    model = CGIC(**params).eval()
finally:
    sys.stdout = _so
freq = freq_tables()["zipf"]
for i, v in enumerate(freq):
    model.quantize.embedding_counter[str(i)].data.fill_(float(v))
hcoder, bcoder = HuffmanCoding(model.quantize.embedding_counter), BinaryCoding()
htab = orc.HuffmanTable(freq)

g = torch.Generator().manual_seed(4)
x = torch.rand(1, 3, H, W, generator=g)
pad, unpad = hr.compute_padding(H, W, min_div=2 ** 4)
x_padded = torch.nn.functional.pad(x, pad, mode="constant", value=0)
h_list, w_list, th_list, tw_list = hr.nonoverlapping_grid_indices(x_padded)
out = {"image_hw": np.array([H, W]), "pad": np.array(pad), "padded_hw": np.array(x_padded.shape[-2:]),
       "h_list": np.array(h_list), "w_list": np.array(w_list), "tile_h": np.array(th_list), "tile_w": np.array(tw_list),
       "ratio": np.array(RATIO), "codebook": model.quantize.embedding.weight.data.numpy().copy()}
bit_sum = 0.0
t = 0
for i in range(len(h_list)):
    for j in range(len(w_list)):
        hi, wi, th, tw = h_list[i], w_list[j], th_list[i], tw_list[j]
        tile = x_padded[:, :, hi:hi + th, wi:wi + tw]
        print(f"tile {t}: {th}x{tw} at ({hi},{wi}) ...", flush=True)
        with torch.no_grad(), tempfile.TemporaryDirectory() as d:
            e8 = model.entropy_calculation_p8(tile)
            e16 = model.entropy_calculation_p16(tile)
            enc = model.encoder(tile, e16, e8)
            z = model.quant_conv(enc["h"])
            cap = {}
            orig = model.decode
            model.decode = lambda q, mk: cap.update(q=q.clone()) or torch.zeros(1, 3, th, tw)
            try:
                _, bpp, _ = model.compress(tile, d, hcoder, bcoder, False)
            finally:
                model.decode = orig
            files = {n: open(os.path.join(d, n + ".bin"), "rb").read() for n in orc.STREAM_NAMES
                     if os.path.exists(os.path.join(d, n + ".bin"))}
        mode = enc["compression_mode"]
        masks = [m.numpy()[0, 0] for m in enc["mask"]]
        # oracle on the captured tensors
        _, _, oidx = orc.vq(z.numpy(), out["codebook"])
        ind = oidx.reshape(th // 4, tw // 4)
        omc, omm, omf, _, omode = orc.router(e16.numpy(), e8.numpy(), RATIO[0], RATIO[1])
        assert omode == mode and all(np.array_equal(a[0, 0], b) for a, b in zip((omc, omm, omf), masks)), "router"
        streams = orc.compress_image(ind, masks[0], masks[1], masks[2], mode, htab)
        assert streams == files, f"tile {t}: streams differ from the reference's files"
        obpp = sum(len(v) for v in streams.values()) * 8 / (th * tw)
        assert obpp == bpp
        oind, _, _, _ = orc.decompress_image(streams, mode, th // 4, tw // 4, htab)
        assert np.array_equal(orc.gather(oind, out["codebook"]), cap["q"].numpy())
        out[f"t{t}_z"] = z.numpy().astype(np.float32)
        out[f"t{t}_e8"], out[f"t{t}_e16"] = e8.numpy(), e16.numpy()
        out[f"t{t}_ind"] = ind.astype(np.int16)
        out[f"t{t}_mode"] = np.int32(mode)
        out[f"t{t}_bpp"] = np.float64(bpp)
        for k, m in zip("cmf", masks):
            out[f"t{t}_m{k}"] = np.packbits(m.astype(np.uint8))
        for n, v in files.items():
            out[f"t{t}_{n}"] = np.frombuffer(v, np.uint8)
        bit_sum += bpp * tw * th                                     # inference_high_resolution.py:250
        print(f"   mode {mode}, bpp {bpp:.5f}, files {[len(v) for v in files.values()]}", flush=True)
        t += 1
out["n_tiles"] = np.int32(t)
out["bpp_image"] = np.float64(bit_sum / W / H)                      # :256 -- over the UNPADDED size
np.savez_compressed(os.path.join(HERE, OUT), **out)
print("wrote", OUT, os.path.getsize(os.path.join(HERE, OUT)), "bytes; bpp_image", out["bpp_image"])
