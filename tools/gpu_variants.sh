#!/bin/bash
# run tools/probe_vq2.py against every tmp_libs/lib_<name>.so given on the command line
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/variants; mkdir -p $O
for n in "$@"; do
  echo "=== $n"
  CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probe_vq2.py 2>&1 | grep -v amdgpu.ids | tee $O/$n.txt
done
