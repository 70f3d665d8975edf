#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 lanes=$3', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do run 20 5 4; run 200 20 4; run 2000 40 4; run 200 20 1; done
