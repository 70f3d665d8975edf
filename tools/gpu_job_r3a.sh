#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3a; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lane_stream" 2>&1 | tail -5 > $O/pytest_lanes.txt
cat $O/pytest_lanes.txt
timeout 600 python tools/probe_k20.py 20 2>&1 | grep -v Warning | tee $O/probe_k20.txt
timeout 600 python tools/probe_k20.py 200 2>&1 | grep -v Warning | tee $O/probe_k200.txt
