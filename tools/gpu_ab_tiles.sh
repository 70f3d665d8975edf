#!/bin/bash
# A/B of variant builds on the tile path and the 256x256 batch in ONE gpurun call: usage gpu_ab_tiles.sh test_lib name [name ...]
cd "$GRAFT_REPO_ROOT" || exit 1
t=$1; shift
CGIC_LIB=$PWD/tmp_libs/lib_$t.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "router or mask or refine or flat or tie or tile or highres" 2>&1 | tail -2
for rep in 1 2; do
  for n in "$@"; do
    for a in "2 768" "64 256"; do
      echo -n "$n [$a]: "; CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probes/probe_b1.py $a 2>&1 | grep -E "router_alone|fused_launch:|whole" | tr '\n' ' '; echo
    done
  done
done
