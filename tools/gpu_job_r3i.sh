#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3 4; do
for f in "" "--no-dist"; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-report $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=20 $f', d['value'], d['ms_per_step'])"
done; done
for f in "" "--no-dist"; do
timeout 600 python bench.py --steps 200 --warmup 20 --no-report $f 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=200 $f', d['value'], d['ms_per_step'])"
done
