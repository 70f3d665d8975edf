#!/bin/bash
# A/B of variant builds on the bench's headline (K=20, K=200 at 4 lanes; K=200 at 1 lane) in ONE gpurun call; usage gpu_ab_bench.sh name [name ...]
cd "$GRAFT_REPO_ROOT" || exit 1
one() { CGIC_LIB=$PWD/tmp_libs/lib_$1.so timeout 300 python bench.py --steps $2 --warmup 20 --lanes $3 --no-report --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1 K=$2 lanes=$3', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
  for n in "$@"; do one $n 20 4; one $n 200 4; one $n 200 1; done
done
