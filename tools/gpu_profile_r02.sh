#!/bin/bash
# round-2 profile set: bench line, kernel-trace stats of the same command, HBM PMC passes, SQ counters of the VQ kernel
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02p; mkdir -p $O
timeout 900 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-extra --schedule sequential"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats -o t -- $CMD) > $O/stats.log 2>&1
python tools/rocprof_stats.py $(find $O/stats -name '*.db' | head -1) > $O/kernel_stats.md
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$C -o pmc -- $CMD --no-graph) > $O/pmc_$C.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc_WRITE_SIZE -name '*.db' | head -1) > $O/pmc_hbm.md 2>&1
cp profiles/pmc_vq.json $O/pmc_vq.json
bash tools/gpu_pmc_vq.sh r02p/sq > /dev/null 2>&1
cp gpurun_out/r02p/sq/pmc_sq.md $O/pmc_sq_vq.md
find $O -name '*.db' -size +6M -delete
cat $O/kernel_stats.md; cat $O/pmc_hbm.md; head -c 1500 $O/bench.json; echo; tail -3 $O/bench.err
