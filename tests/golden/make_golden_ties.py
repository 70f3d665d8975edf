#!/usr/bin/env python3
"""Golden fixture for the tie-heavy content families (tests/golden/ties.npz), from the REAL reference.

Runs only in the build container (needs /root/reference).  For small 8-bit images of every family of
oracle/content_families.py it runs the reference's own `Entropy(8)`, `Entropy(16)` (CGIC/models/model.py:433-483) and
`TripleGrainFixedEntropyRouter` (CGIC/modules/vqvae/RouterTriple.py) and stores inputs (uint8) and outputs.  Before writing
it requires that oracle/entropy_torch.py -- the torch-CPU restatement the GPU tests and bench.py use as "the reference's
arithmetic" on the GPU box -- reproduces the real class BIT FOR BIT (1 and 8 threads), on these images and on 256x256
ones, and that the oracle's router reproduces the real router's masks from those maps.

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden_ties.py
"""
import os
import platform
import sys
import types
from unittest.mock import MagicMock

import numpy as np
import torch

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
tv = MagicMock()
sys.modules["torchvision"] = tv
sys.modules["torchvision.transforms"] = tv.transforms
torch.nn.Module.cuda = lambda self, device=None: self

from CGIC.models.model import Entropy  # noqa: E402
from CGIC.modules.vqvae.RouterTriple import TripleGrainFixedEntropyRouter  # noqa: E402

from oracle import cgic_oracle as orc  # noqa: E402
from oracle import entropy_torch as et  # noqa: E402
from oracle.content_families import families  # noqa: E402


def check(cond, msg):
    if not cond:
        raise SystemExit("MISMATCH: " + msg)


def main():
    out = {"torch_version": np.array(torch.__version__), "machine": np.array(platform.machine())}
    # 1. the restatement == the real class, bit for bit, on full-size images of every family, 1 and 8 threads
    for name, x in families(n=4, H=256, W=256, seed=21).items():
        xt = torch.from_numpy(x)
        for p in (8, 16):
            for nt in (8, 1):
                torch.set_num_threads(nt)
                check(torch.equal(Entropy(p)(xt), et.entropy_map(xt, p)), f"{name} p{p} threads {nt}: entropy_torch differs from the reference")
    torch.set_num_threads(8)
    # 2. the committed vectors: small images, reference outputs
    router = TripleGrainFixedEntropyRouter(0.1, 0.8)
    for name, x in families(n=2, H=96, W=128, seed=5).items():
        u8 = np.round(x * 255.0).astype(np.uint8)
        check(np.array_equal(u8.astype(np.float32) / 255.0, x), f"{name}: not exactly 8-bit")
        xt = torch.from_numpy(x)
        e8, e16 = Entropy(8)(xt), Entropy(16)(xt)
        check(torch.equal(e8, et.entropy_map(xt, 8)) and torch.equal(e16, et.entropy_map(xt, 16)), f"{name}: restatement")
        out[name + "_u8"] = u8
        out[name + "_e8"] = e8.numpy()
        out[name + "_e16"] = e16.numpy()
        for b in range(x.shape[0]):
            mask, _, _, mode = router(e16[b:b + 1], e8[b:b + 1])
            o = orc.router(e16[b:b + 1].numpy(), e8[b:b + 1].numpy(), 0.1, 0.8)
            check(o[4] == mode and all(np.array_equal(m.numpy(), q) for m, q in zip(mask, o[:3])), f"{name}[{b}]: oracle router")
            for k, m in zip("cmf", mask):
                out[f"{name}_{b}_m{k}"] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
        d8 = float(np.abs(orc.entropy(x, 8) - e8.numpy()).max())
        print(f"  {name}: distinct e8 values {len(np.unique(e8.numpy()))} of {e8.numel()}, C oracle max |diff| {d8:.2e}")
    # 3. (round 6) bands of realistic length for the GPU's pixels -> masks check: two 256x256 images per family and one smooth
    # 768x768 tile, per-image masks of the REAL router on the REAL Entropy maps, and the real router's batch-global masks
    # (the reference's encode() routes over the flattened batch: RouterTriple.py:21,40,52,63) of each pair.  Only inputs
    # (uint8) and masks (bits) are stored.  The masks must also follow from the correctly rounded restatement
    # (cgic_oracle_entropy_ref == the GPU's reference-order arithmetic, bit for bit): a fixture whose masks hang on MKL's
    # last-bit exp / log rounding could not be demanded of any other implementation.
    big = {name + "_256": x for name, x in families(n=2, H=256, W=256, seed=33).items()}
    big["smooth8_768"] = families(n=1, H=768, W=768, seed=11)["smooth8"]
    for name, x in big.items():
        u8 = np.round(x * 255.0).astype(np.uint8)
        check(np.array_equal(u8.astype(np.float32) / 255.0, x), f"{name}: not exactly 8-bit")
        xt = torch.from_numpy(x)
        e8, e16 = Entropy(8)(xt), Entropy(16)(xt)
        a8, a16 = orc.entropy_ref(x, 8), orc.entropy_ref(x, 16)
        out[name + "_u8"] = u8
        nb = x.shape[0]
        for b in list(range(nb)) + (["batch"] if nb > 1 else []):
            sl = slice(0, nb) if b == "batch" else slice(b, b + 1)
            mask, _, _, mode = router(e16[sl], e8[sl])
            o = orc.router(a16[sl], a8[sl], 0.1, 0.8)
            check(o[4] == mode and all(np.array_equal(m.numpy(), q) for m, q in zip(mask, o[:3])),
                  f"{name}[{b}]: the masks of the correctly rounded restatement differ from the real router's")
            for k, m in zip("cmf", mask):
                out[f"{name}_{b}_m{k}"] = np.packbits(m.numpy().astype(np.uint8).reshape(-1))
        t16 = np.sort(e16.numpy().reshape(-1))
        print(f"  {name}: distinct e16 {len(np.unique(t16))} of {t16.size}, "
              f"within 4e-6 of the coarse threshold: {int((np.abs(e16.numpy() - t16[max(round(t16.size / nb * 0.1) - 1, 0)]) < 4e-6).sum())}")
    path = os.path.join(HERE, "ties.npz")
    np.savez_compressed(path, **out)
    print(f"  wrote ties.npz ({os.path.getsize(path)} bytes)")


if __name__ == "__main__":
    main()
