#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2c; mkdir -p $O
for v in "--lanes 4" "--lanes 4 --split-router" "--lanes 1" "--lanes 1 --split-router"  "--lanes 8 --split-router"; do
for kw in "--steps 20 --warmup 5" "--steps 200 --warmup 20"; do
timeout 300 python bench.py $kw --no-extra --no-cpu-baseline $v 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$kw', '$v', d['value'], d['ms_per_step'], d.get('bpp_match'), d['config']['inputs'][:12])"
done; done
tail -3 $O/err
