#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('K=$1 lanes=$3', d['value'], d['ms_per_step'])"; }
python tools/probe_stage.py 2>/dev/null | tail -1
run 192 24 4; run 192 24 1; run 20 5 4
python tools/probe_tiles.py 8 2>/dev/null | grep "vq\|router"
python tools/probe_tiles.py 32 2>/dev/null | grep "vq\|router"
python tools/probe_tiles.py 1 2>/dev/null | grep "vq\|router"
python tools/probe_tiles.py 16 512 2>/dev/null | grep "vq\|router"
