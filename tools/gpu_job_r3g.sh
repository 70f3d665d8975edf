#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3g; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -25 > $O/pytest.txt
cat $O/pytest.txt | cut -c1-300
