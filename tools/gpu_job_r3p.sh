#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3p; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "compress or codec or lane or config1" 2>&1 | tail -4 | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python tools/stress_codec.py 21 30 latency 2>&1 | tail -1
timeout 600 python tools/probe_vq_variant.py nolanes_no 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/variant.txt
for k in "20 5" "20 5" "200 20" "2000 20"; do set -- $k
timeout 600 python bench.py --steps $1 --warmup $2 --no-report 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=$1', d['value'], d['ms_per_step'])"
done | tee $O/bench.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['stages_us'], d['one_batch_in_flight']['ms_per_step'], d['single_batch']['ms_per_step'], d['bpp_match'])"
