#!/bin/bash
# A/B of the entropy kernel (with / without the flat8 by-product) and of the one-lane step across variant builds; usage: gpu_ab_ent.sh name [name ...]
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do for n in "$@"; do
  CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python - <<PY
import sys; sys.path.insert(0, ".")
import torch, bench, control_gic_amd as cg
x, z, cb = bench.make_inputs(64, 256, 256, 1)
xd = torch.from_numpy(x).cuda()
a = bench.graph_kernel_time(lambda: cg.entropy_maps(xd))
b = bench.graph_kernel_time(lambda: cg.entropy_maps(xd, want_flat=False))
print("$n entropy_maps: with flat8 %.2f us, without %.2f us" % (a, b))
PY
done; done 2>&1 | grep -v amdgpu
