#!/bin/bash
# run tools/probes/probe_entropy_sizes.py against every tmp_libs/lib_<name>.so given on the command line
cd "$GRAFT_REPO_ROOT" || exit 1
for n in "$@"; do
  echo "=== $n"
  CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probes/probe_entropy_sizes.py 2>&1 | grep -v amdgpu.ids
done
