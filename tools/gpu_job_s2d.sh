#!/bin/bash
# usage: gpu_job_s2d.sh "<bench args>" lib1 lib2 ...   ("" = the product library)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2d; mkdir -p $O
ARGS="$1"; shift
for lib in "$@"; do
for kw in "--steps 200 --warmup 20" "--steps 20 --warmup 5"; do
L=""; [ "$lib" != "base" ] && L="$PWD/tmp_libs/lib_$lib.so"
CGIC_LIB=$L timeout 300 python bench.py $kw --no-extra --no-cpu-baseline $ARGS 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('lib=$lib', '$kw', '$ARGS', d['value'], d['ms_per_step'], d.get('bpp_match'))"
done; done
tail -2 $O/err
