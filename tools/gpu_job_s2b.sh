#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/s2b; mkdir -p $O
for v in "" "--no-ring"; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline $v 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('20 steps', '$v', d['value'], d['ms_per_step'], d.get('bpp_match'))"
timeout 300 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline $v 2>>$O/err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('200 steps', '$v', d['value'], d['ms_per_step'], d.get('bpp_match'))"
done
tail -3 $O/err
