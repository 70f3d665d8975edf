#!/bin/bash
# run tools/probes/probe_tiles.py (T S from the first two arguments) against every tmp_libs/lib_<name>.so given after them
cd "$GRAFT_REPO_ROOT" || exit 1
T=$1; S=$2; shift 2
for n in "$@"; do
  echo "=== $n"
  CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probes/probe_tiles.py $T $S 2>&1 | grep -v amdgpu.ids
done
