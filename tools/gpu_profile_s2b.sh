#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s2p; mkdir -p $O
for L in 1 4; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 16 --no-report --lanes $L"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats$L -o t -- $CMD) > $O/stats$L.log 2>&1
  db=$(find $O/stats$L -name '*.db' | head -1)
  python tools/rocprof_stats.py $db > $O/kernel_stats_lanes$L.md
  python tools/trace_concurrency.py $db 96 > $O/loop_lanes$L.md
  grep -o '"value": [0-9.]*\|"ms_per_step": [0-9.]*' $O/stats$L.log | head -2
done
find $O -name '*.db' -size +6M -delete
cat $O/loop_lanes1.md $O/loop_lanes4.md
