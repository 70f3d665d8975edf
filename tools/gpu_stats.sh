#!/bin/bash
# kernel-trace stats of the hot path at a given shape; usage: gpu_stats.sh name B S iters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/tools/run_hotpath.py $2 $3 $4) > $O/trace.log 2>&1
python tools/rocprof_stats.py $(find $O/trace -name '*.db' | head -1) | tee $O/stats.md
find $O -name '*.db' -size +8M -delete
