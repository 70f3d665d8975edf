#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes 4 $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 W=$2 $3', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
run 192 24 "--slots 8"
run 192 24 "--slots 12"
run 192 24 "--slots 16"
run 192 24 "--slots 8 --no-ring"
run 20 5 "--slots 8"
run 20 5 "--slots 12"
run 20 5 "--slots 16"
run 20 5 "--slots 8 --no-ring"
done
