#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu -k "f2_f4 or mixed_stream or differentiable or codec_custom" 2>&1 | tail -25 | cut -c1-400 > $O/pytest.txt
cat $O/pytest.txt
timeout 600 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline 2>$O/bench.err | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print({k:d[k] for k in ('value','ms_per_step','rccl_ranks','per_rank_MPixels/s','histogram_allreduce_us')}, d['config'].get('distributed'))"
tail -3 $O/bench.err
