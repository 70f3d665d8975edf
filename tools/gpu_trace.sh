#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c; mkdir -p $O
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/trace_pipe -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 12 --warmup 4 --no-extra --no-cpu-baseline --slots 3) > $O/trace_pipe.log 2>&1
db=$(find $O/trace_pipe -name '*.db' | head -1)
python tools/trace_overlap.py $db 400 > $O/overlap_all.txt 2>&1
grep -n "compress_streams\|decode_streams\|merge_kernel\|entropy\|vq_filter_router" $O/overlap_all.txt | head -90
