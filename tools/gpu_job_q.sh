#!/bin/bash
# quick A/B of a VQ build: the roofline command (HIP events), K=20 / K=200 / K=2000 headline, VQ parity + stress tests
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2 3; do timeout 120 python tools/run_roofline_cmd.py fused 2>&1 | grep "us per launch"; done
for K in 20 200 2000; do
  for i in 1 2; do timeout 300 python bench.py --steps $K --warmup 20 --no-report --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K=$K', d['value'], d['ms_per_step'])"; done
done
[ "$1" = "notest" ] || timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py -q -x -m gpu -k "vq or stress or filter" 2>&1 | tail -3
