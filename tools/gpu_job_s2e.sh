#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
for q in 4 8; do for l in 4 6 8; do
GPU_MAX_HW_QUEUES=$q timeout 300 python bench.py --steps 192 --warmup 24 --no-extra --no-cpu-baseline --lanes $l 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('queues $q lanes $l', d['value'], d['ms_per_step'], d.get('bpp_match'))"
done; done
