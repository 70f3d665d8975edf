#!/bin/bash
# usage: gpu_variants2.sh "ENV=.. ENV2=.." name [ "ENV" name ... ]  -- probe_vq2 with per-variant environment
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/variants; mkdir -p $O
while [ $# -ge 2 ]; do
  e=$1; n=$2; shift 2
  echo "=== $n ($e)"
  env $e CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probe_vq2.py 2>&1 | grep -v amdgpu.ids | tee $O/$n.txt
done
