#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes $3 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$4 K=$1 lanes=$3', d['value'], d['ms_per_step'])"; }
for rep in 1 2; do
unset HIP_FORCE_DEV_KERNARG; run 20 5 4 base; run 200 20 4 base; run 200 20 1 base
export HIP_FORCE_DEV_KERNARG=1; run 20 5 4 devkernarg; run 200 20 4 devkernarg; run 200 20 1 devkernarg
export HIP_FORCE_DEV_KERNARG=0; run 20 5 4 devkernarg0; run 200 20 4 devkernarg0; run 200 20 1 devkernarg0
done
