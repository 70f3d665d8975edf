#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -8 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python tools/probe_k20.py 20 2>&1 | grep -v Warning | tee $O/probe_k20.txt
for i in 1 2; do
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>$O/bench20.err | tee $O/bench20_$i.json | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=20', d['value'], d['ms_per_step'], json.dumps(d.get('stages_us')), d['roofline']['frac'], d['roofline'].get('vq_alone_frac'))"
done
timeout 600 python bench.py --steps 200 --warmup 20 --no-report 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=200', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --steps 2000 --warmup 20 --no-report 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=2000', d['value'], d['ms_per_step'])"
