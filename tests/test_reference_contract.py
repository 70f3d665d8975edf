"""CPU, build container only: install() against the REAL reference class.

The GPU tests exercise install() on a CGIC-shaped stub; this one builds the real 130 M-parameter `CGIC`
(CGIC/models/model.py) on the CPU -- in a subprocess, with the harness shims of SURVEY.md Appendix A -- and checks the
attribute contract install() relies on, that install() runs on it without launching anything, keeps every state_dict key and
shape, and points the router target at the drop-in.  Skipped where /root/reference does not exist (the GPU box)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

REF = "/root/reference"

SCRIPT = r'''
import sys, types
from unittest.mock import MagicMock
import torch, yaml
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference"); sys.path.insert(0, sys.argv[1])
pl = types.ModuleType("pytorch_lightning"); pl.LightningModule = torch.nn.Module; pl.LightningDataModule = object
sys.modules["pytorch_lightning"] = pl
tv = MagicMock()
for n in ("torchvision", "torchvision.transforms", "torchvision.transforms.functional", "torchvision.utils"):
    sys.modules[n] = tv
torch.nn.Module.cuda = lambda self, device=None: self
from CGIC.models.model import CGIC
import control_gic_amd as cg

params = yaml.safe_load(open("/root/reference/configs/config_inference.yaml"))["model"]["params"]
params["ckpt_path"] = None; params["lossconfig"] = None
torch.manual_seed(0)
model = CGIC(**params).eval()

# ---- the attribute contract install() / compress_batch() rely on (model.py:42-60, 99-117; vqvae_blocks.py:354-355)
q = model.quantize
assert type(q).__name__ == "VectorQuantize2" and (q.n_e, q.e_dim) == (1024, 4) and hasattr(q, "beta") and q.legacy is True
assert tuple(q.embedding.weight.shape) == (1024, 4) and len(q.embedding_counter) == 1024
assert list(q.embedding_counter.keys())[:4] == ["0", "1", "10", "100"]          # ParameterDict sorts plain-dict keys as strings
rc = model.encoder.router_config
assert rc["target"].endswith("TripleGrainFixedEntropyRouter") and {"coarse_grain_ratio", "medium_grain_ratio"} <= set(rc["params"])
assert type(model.entropy_calculation_p8).__name__ == "Entropy" and type(model.entropy_calculation_p16).__name__ == "Entropy"
assert isinstance(model.quant_conv, torch.nn.Conv2d) and model.quant_conv.kernel_size == (1, 1) and model.quant_conv.in_channels == 4
assert isinstance(model.post_quant_conv, torch.nn.Conv2d) and model.post_quant_conv.kernel_size == (1, 1)
assert callable(model.encode) and callable(model.decode) and callable(model.compress)

keys = {k: tuple(v.shape) for k, v in model.state_dict().items()}
emb = q.embedding.weight.detach().clone()
q.embedding_counter["7"].data.fill_(42.0)

cg.install(model)                                                                  # CPU: nothing is launched

assert isinstance(model.quantize, cg.VectorQuantize2) and isinstance(model.entropy_calculation_p8, cg.Entropy)
assert model.encoder.router_config["target"] == "control_gic_amd.router.TripleGrainFixedEntropyRouter"
assert model.encoder.router_config["params"]["per_image"] is False                # the reference's batch semantics stay the default
from CGIC.util import instantiate_from_config
r = instantiate_from_config(model.encoder.router_config)
assert isinstance(r, cg.TripleGrainFixedEntropyRouter) and r.per_image is False
after = {k: tuple(v.shape) for k, v in model.state_dict().items()}
assert after == keys, (set(after) ^ set(keys))                                     # same keys, same shapes: checkpoints keep loading
assert torch.equal(model.quantize.embedding.weight, emb) and model.quantize.embedding_counter["7"].item() == 42.0
assert list(model.quantize.embedding_counter)[:4] == ["0", "1", "10", "100"]
# the table HuffmanCoding(model.quantize.embedding_counter) is built from, in the reference's order (inference.py:137-139)
h = cg.HuffmanCoding(model.quantize.embedding_counter)
from CGIC.tools.indices_coding import HuffmanCoding as RefHuffman
ref = RefHuffman({k: model.quantize.embedding_counter[k] for k in model.quantize.embedding_counter})
assert h.codes == ref.codes
print("CONTRACT_OK", len(keys))
'''


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree exists in the build container only")
def test_install_on_the_real_cgic_class(tmp_path):
    script = tmp_path / "contract.py"
    script.write_text(SCRIPT)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    r = subprocess.run([sys.executable, str(script), ROOT], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0 and "CONTRACT_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
