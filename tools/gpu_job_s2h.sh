#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 300 python bench.py --steps $1 --warmup $2 --no-report --lanes $4 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$3 K=$1 lanes=$4', d['value'], d['ms_per_step'])"; }
for lib in base e512 base e512; do
if [ $lib = base ]; then unset CGIC_LIB; else export CGIC_LIB=$PWD/tmp_libs/lib_$lib.so; fi
python tools/probe_stage.py 2>/dev/null | tail -1 | cut -c1-200
run 200 20 $lib 4; run 2000 40 $lib 4; run 200 20 $lib 1
done
python -m pytest tests -m gpu -x -q -k "compress or coder or huffman or codec" 2>&1 | tail -2
