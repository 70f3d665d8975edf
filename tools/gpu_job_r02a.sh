#!/bin/bash
# round-2 GPU job A: sanity tests, SQ counter passes for the VQ filter kernel, dbg phase clocks, micro-probes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_LDS"
PMC_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_INSTS_SALU"
PMC_C="SQ_WAVES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU_MFMA_BF16 SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"
for S in A B C; do
  eval L=\$PMC_$S
  (cd /tmp && timeout 300 rocprofv3 --pmc $L --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$S -o pmc -- python $GRAFT_REPO_ROOT/tools/run_vq_only.py 20) > $O/pmc_$S.log 2>&1
  echo "pmc $S rc=$?" >> $O/pmc_$S.log
done
find $O -name '*.db' | sort > $O/dbs.txt
python tools/pmc_sq_summary.py $(cat $O/dbs.txt) > $O/pmc_sq.md 2>&1
CGIC_LIB=$PWD/control-gic_amd/libcgic_hip_dbg.so timeout 300 python tools/probe_vqk.py > $O/vqk.txt 2>&1
for p in probe_mfma probe_valu2 probe_vq; do
  [ -x tools/$p ] && timeout 120 tools/$p > $O/$p.txt 2>&1
done
find $O -name '*.db' -size +8M -delete
tail -3 $O/pytest.log
cat $O/pmc_sq.md | head -60
cat $O/vqk.txt
