#!/bin/bash
# round-2 (second half) profile set: bench line, kernel-trace stats with one and four batches in flight, HBM PMC passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/s2p; mkdir -p $O
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench200.json 2> $O/bench.err; echo "bench rc=$?"
timeout 900 python bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline > $O/bench20.json 2>> $O/bench.err; echo "bench rc=$?"
for L in 1 4; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 16 --no-report --lanes $L"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats$L -o t -- $CMD) > $O/stats$L.log 2>&1
  db=$(find $O/stats$L -name '*.db' | head -1)
  python tools/rocprof_stats.py $db > $O/kernel_stats_lanes$L.md
  python tools/trace_concurrency.py $db 96 > $O/loop_lanes$L.md
done
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --lanes 1 --no-graph"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$C -o pmc -- $CMD) > $O/pmc_$C.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc_WRITE_SIZE -name '*.db' | head -1) > $O/pmc_hbm.md 2>&1
cp profiles/pmc_vq.json $O/pmc_vq.json
python tools/probe_lanes_kernel.py 2>&1 | grep -v amdgpu.ids > $O/lanes_kernel.txt
find $O -name '*.db' -size +6M -delete
head -12 $O/kernel_stats_lanes1.md; cat $O/loop_lanes1.md $O/loop_lanes4.md; cat $O/pmc_hbm.md; cat $O/lanes_kernel.txt; head -c 300 $O/bench200.json; echo; head -c 300 $O/bench20.json
