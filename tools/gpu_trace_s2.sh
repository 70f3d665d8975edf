#!/bin/bash
# kernel trace of the sequential bench: per-kernel durations and the gaps between successive kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-s2trace}; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/trace -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-extra --no-cpu-baseline) > $O/trace.log 2>&1
db=$(find $O/trace -name '*.db' | head -1)
python tools/rocprof_stats.py $db > $O/kernel_stats.md
python tools/trace_overlap.py $db 60 > $O/overlap_tail.txt 2>&1
python - $db > $O/gaps.txt <<'PY'
import sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
rows = rows[-400:]
gaps = collections.defaultdict(list)
short = lambda n: n.split("(")[0].replace("void ", "").replace("cgic::", "")[:40]
for (n0, s0, e0), (n1, s1, e1) in zip(rows, rows[1:]):
    gaps[(short(n0), short(n1))].append((s1 - e0) / 1e3)
for k, v in gaps.items():
    v.sort()
    print(f"{k[0]:40s} -> {k[1]:40s} n={len(v):4d} gap med {v[len(v)//2]:6.2f} min {v[0]:6.2f} max {v[-1]:6.2f} us")
PY
find $O -name '*.db' -size +6M -delete
cat $O/kernel_stats.md; cat $O/gaps.txt; tail -12 $O/overlap_tail.txt
