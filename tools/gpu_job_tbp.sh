#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/tbp; rm -rf $O; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/st -o t -- python $GRAFT_REPO_ROOT/tools/probes/probe_tiled_batch.py ${1:-8}) > $O/run.log 2>&1 < /dev/null
db=$(find $O/st -name '*.db' | head -1)
[ -n "$db" ] && python tools/rocprof_stats.py $db > $O/kernel_stats.md < /dev/null
grep "eager" $O/run.log
head -30 $O/kernel_stats.md < /dev/null
rm -rf $O/st
