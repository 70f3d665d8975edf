#!/bin/bash
# kernel trace of the bench: per-kernel durations; the tail of the timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-s2trace}; shift; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 16 --no-extra --no-cpu-baseline "$@") > $O/trace.log 2>&1
db=$(find $O/trace -name '*.db' | head -1)
python tools/rocprof_stats.py $db > $O/kernel_stats.md
python - $db <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
c = collections.Counter((r[3], r[4]) for r in rows)
print("(queue, stream): launches", sorted(c.items()))
PY
head -9 $O/kernel_stats.md
