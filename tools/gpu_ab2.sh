#!/bin/bash
# A/B of variant builds, fused launch on the bench's ordinary batch with the refinement queues' scratch (CGIC_REFINE_FUSED_QUEUES=1) and without
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
  for n in "$@"; do
    for q in 1 0; do
      echo -n "$n queues=$q fused: "; CGIC_REFINE_FUSED_QUEUES=$q CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 120 python tools/run_roofline_cmd.py fused 2>&1 | grep -o "[0-9.]* us per launch"
    done
  done
done
