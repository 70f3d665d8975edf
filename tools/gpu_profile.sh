#!/bin/bash
# the round's profiling passes of the final build (kernel traces at --lanes 1 / 4, the roofline command alone, HBM and SQ counters)
# usage (on the GPU box, through gpurun): bash tools/gpu_profile.sh [r05]   -> gpurun_out/<round>p; then tools/assemble_profiles.sh <round> here
R=${1:-r06}
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/${R}p; rm -rf $O; mkdir -p $O
for L in 1 4; do
  CMD="python $GRAFT_REPO_ROOT/bench.py --steps 96 --warmup 16 --no-report --lanes $L"
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/stats$L -o t -- $CMD) > $O/stats$L.log 2>&1
  db=$(find $O/stats$L -name '*.db' | head -1)
  python tools/rocprof_stats.py $db > $O/kernel_stats_lanes$L.md
  python tools/trace_concurrency.py $db 96 $O/loop_lanes$L.json > $O/loop_lanes$L.md
done
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/alone -o t -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py fused) > $O/alone.log 2>&1
cp $(find $O/alone -name '*.db' | head -1) $O/alone.db
python tools/rocprof_stats.py $O/alone.db > $O/kernel_stats_alone.md
(cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/alonevq -o t -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py vq) > $O/alonevq.log 2>&1
python tools/rocprof_stats.py $(find $O/alonevq -name '*.db' | head -1) > $O/kernel_stats_alone_vq.md
# HBM bytes: separate passes per counter (MI355X_MICROARCH.md)
CMDP="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-report --lanes 1 --no-graph --no-dist"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$C -o pmc -- $CMDP) > $O/pmc_$C.log 2>&1
done
# ... and of the kernels of the four-lane step (throughput decoder, ONE-band merge: one staging per image)
CMDP4="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-report --lanes 4 --no-graph --no-dist"
for C in FETCH_SIZE WRITE_SIZE; do
  (cd /tmp && timeout 600 rocprofv3 --pmc $C --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc4_$C -o pmc -- $CMDP4) > $O/pmc4_$C.log 2>&1
done
python tools/pmc_summary.py $(find $O/pmc4_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc4_WRITE_SIZE -name '*.db' | head -1) > $O/pmc_hbm_lanes4.md 2>&1
CGIC_PMC_STEP_JSON=$O/pmc_step.json python tools/pmc_summary.py $(find $O/pmc4_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc4_WRITE_SIZE -name '*.db' | head -1) > /dev/null 2>&1
cp profiles/pmc_vq.json $O/pmc_vq_before.json 2>/dev/null
python tools/pmc_summary.py $(find $O/pmc_FETCH_SIZE -name '*.db' | head -1) $(find $O/pmc_WRITE_SIZE -name '*.db' | head -1) > $O/pmc_hbm.md 2>&1
cp profiles/pmc_vq.json $O/pmc_hbm.json
# SQ instruction counters of every kernel of the step
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_sq -o pmc -- $CMDP) > $O/pmc_sq.log 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc_sq -name '*.db' | head -1) > $O/pmc_sq.md 2>&1
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_BUSY_CYCLES --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_sq2 -o pmc -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py fused) > $O/pmc_sq2.log 2>&1
python tools/pmc_sq_summary.py --match vq_filter $(find $O/pmc_sq2 -name '*.db' | head -1) > $O/pmc_sq_vq.md 2>&1
# ... and of the kernels the four-lane step uses instead (throughput decoder, one-band merge), eager on four streams
CMD4="python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 10 --no-report --lanes 4 --no-graph --no-dist"
(cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_VALU --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_sq4 -o pmc -- $CMD4) > $O/pmc_sq4.log 2>&1
python tools/pmc_sq_summary.py $(find $O/pmc_sq4 -name '*.db' | head -1) > $O/pmc_sq_lanes4.md 2>&1
# matrix-core / LDS / wait counters of the shipping fused launch alone (the roofline command), three passes
PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
PMC_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"
PMC_C="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
for S in A B C; do
  eval L=\$PMC_$S
  (cd /tmp && timeout 300 rocprofv3 --pmc $L --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmcf_$S -o pmc -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py fused) > $O/pmcf_$S.log 2>&1
done
python tools/pmc_sq_summary.py --match vq_filter $(find $O/pmcf_A $O/pmcf_B $O/pmcf_C -name '*.db' | sort) > $O/pmc_sq_vq_fused.md 2>&1
python tools/pmc_sq_summary.py --json --match vq_filter $(find $O/pmcf_A $O/pmcf_B $O/pmcf_C -name '*.db' | sort) > $O/pmc_sq_vq_fused.json 2>/dev/null
# the stand-alone router launch on tie-heavy content, refinement queues on / off (tools/probes/probe_refine_queue.py: its own HIP-event figures)
(timeout 300 python tools/probes/probe_refine_queue.py smooth8 flat_edges tiles) > $O/refine_queue.log 2>&1
(timeout 300 python tools/probes/probe_entropy_err.py) > $O/entropy_err.log 2>&1
python tools/roofline_json.py $O $R > $O/roofline.log 2>&1
find $O -name '*.db' -size +6M -delete
find $O -name '*.csv' -size +2M -delete
cat $O/loop_lanes1.md $O/loop_lanes4.md; tail -5 $O/alone.log; tail -30 $O/roofline.log; du -sh $O
