#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for q in 4 6 8; do for l in 4 6 8; do
[ $l -gt $q ] && continue
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 192 --warmup 24 --no-report --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q lanes $l', d['value'], d['ms_per_step'])"
done; done
