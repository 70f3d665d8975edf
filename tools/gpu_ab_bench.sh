#!/bin/bash
# A/B of variant builds on the WHOLE step (bench.py's timed loop, K=20 and K=200) in ONE gpurun call: usage gpu_ab_bench.sh name [name ...] -- tmp_libs/lib_<name>.so
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2 3; do
  for n in "$@"; do
    for k in 20 200; do
      echo -n "$n K=$k: "; CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python bench.py --steps $k --warmup 5 --no-report 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); print(d[\"value\"], d[\"ms_per_step\"])"
    done
  done
done
