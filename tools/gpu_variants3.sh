#!/bin/bash
# usage: gpu_variants3.sh name [name ...]  -- tools/probes/probe_vq_variant.py per variant library in tmp_libs/
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/variants3; mkdir -p $O
for n in "$@"; do
  echo "=== $n"
  CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 300 python tools/probes/probe_vq_variant.py 2>&1 | grep -v "amdgpu.ids\|Warning" | tee $O/$n.txt
done
