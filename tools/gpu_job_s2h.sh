#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes 4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3 K=$1', d['value'], d['ms_per_step'])"; }
for lib in nofuse base f512 f1024; do
if [ $lib = base ]; then unset CGIC_LIB; else export CGIC_LIB=$PWD/tmp_libs/lib_$lib.so; fi
python tools/probe_dec_modes.py 2>&1 | grep throughput
run 20 5 $lib; run 200 20 $lib; run 2000 40 $lib
done
