#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('stagger=$CGIC_LANE_STAGGER_CYCLES K=$1', d['value'], d['ms_per_step'])"; }
for st in 0 10000 20000 40000 0 20000; do
export CGIC_LANE_STAGGER_CYCLES=$st
run 20 5; run 20 5; run 200 20
done
