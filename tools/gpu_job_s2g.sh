#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes 4 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 W=$2 $3', d['value'], d['ms_per_step'])"; }
for rep in 1 2 3; do
run 20 5 ""
run 20 5 "--alt-router"
run 192 24 ""
run 192 24 "--alt-router"
done
