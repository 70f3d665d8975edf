#!/bin/bash
# session-2 baseline: full gpu test suite + the driver's bench command + a 200-step run
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $O/bench200.json 2>> $O/bench.err; echo "rc=$?"
tail -3 $O/pytest.log; head -c 600 $O/bench.json; echo; head -c 400 $O/bench200.json
