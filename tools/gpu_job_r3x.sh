#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3x; mkdir -p $O
timeout 400 python tools/stress_vq.py 31 150 2>&1 | tail -2 | tee $O/stress_vq.txt
timeout 400 python tools/stress_codec.py 32 120 throughput 2>&1 | tail -1 | tee $O/stress_codec_tp.txt
timeout 400 python tools/stress_codec.py 33 120 latency 2>&1 | tail -1 | tee $O/stress_codec_lat.txt
for i in 1 2; do timeout 600 python bench.py --steps 6000 --warmup 20 --no-extra --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=6000', d['value'], d['ms_per_step'], 'bpp_match', d['bpp_match'])"; done | tee $O/bench6000.txt
