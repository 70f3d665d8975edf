#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3r; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "reference_arithmetic or entropy or tie_heavy" 2>&1 | tail -15 | cut -c1-400 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python - <<'PY' 2>&1 | grep -v "amdgpu.ids\|Warning"
import sys; sys.path.insert(0, ".")
import torch, numpy as np, control_gic_amd as cg, bench
from oracle.content_families import families
x = torch.from_numpy(families(n=64)["smooth8"]).cuda()
print("reference-order entropy kernel B=64 256x256: %.1f us per launch; default %.1f" % (bench.graph_kernel_time(lambda: cg.entropy_maps(x, reference_order=True), 5, 3), bench.graph_kernel_time(lambda: cg.entropy_maps(x), 5, 3)))
PY
