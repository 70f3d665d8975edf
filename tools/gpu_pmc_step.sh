#!/bin/bash
# SQ instruction counters of every kernel of the step (one pass)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${1:-s2sq_step}; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-report --lanes ${2:-1} --no-graph"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc -o pmc -- $CMD) > $O/pmc.log 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc -name '*.db' | head -1) > $O/pmc_sq.md 2>&1
find $O -name '*.db' -size +8M -delete
grep -A11 "^###" $O/pmc_sq.md | grep -v "^--" 
