#!/bin/bash
# A/B of variant builds in ONE gpurun call (boxes of the pool differ by ~1 us): usage gpu_ab.sh name [name ...] -- tmp_libs/lib_<name>.so
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for n in "$@"; do
    for w in vq fused; do
      echo -n "$n $w: "; CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 120 python tools/run_roofline_cmd.py $w 2>&1 | grep -o "[0-9.]* us per launch"
    done
  done
done
