#!/bin/bash
# SQ counter passes for the shipping fused VQ + router launch alone (tools/run_roofline_cmd.py fused); usage: gpu_pmc_fused.sh <outdir-name>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
PMC_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"
PMC_C="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU_MFMA_F16 SQ_INSTS_VALU_MFMA_F32 SQ_INSTS_VMEM SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
for S in A B C; do
  eval L=\$PMC_$S
  (cd /tmp && timeout 300 rocprofv3 --pmc $L --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$S -o pmc -- python $GRAFT_REPO_ROOT/tools/run_roofline_cmd.py ${2:-fused}) > $O/pmc_$S.log 2>&1
  tail -2 $O/pmc_$S.log
done
python tools/pmc_sq_summary.py --match vq_filter $(find $O -name '*.db' | sort) > $O/pmc_sq_vq.md 2>&1
find $O -name '*.db' -size +8M -delete
cat $O/pmc_sq_vq.md
