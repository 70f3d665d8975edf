#!/bin/bash
# SQ counter passes for the entropy kernel alone; usage: gpu_pmc_ent.sh <outdir-name>
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/$1; mkdir -p $O
cat > /tmp/run_ent_only.py <<'PY'
import sys, os, torch
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import control_gic_amd as cg
xs = [torch.rand(64, 3, 256, 256, device="cuda") for _ in range(8)]
for i in range(24): cg.entropy_maps(xs[i % 8])
torch.cuda.synchronize()
PY
PMC_A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
PMC_B="SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVES SQ_INSTS_VALU_TRANS_F32"
for S in A B; do
  eval L=\$PMC_$S
  (cd /tmp && timeout 300 rocprofv3 --pmc $L --kernel-trace -d $GRAFT_REPO_ROOT/$O/pmc_$S -o pmc -- python /tmp/run_ent_only.py) > $O/pmc_$S.log 2>&1
done
python tools/pmc_sq_summary.py $(find $O -name '*.db' | sort) > $O/pmc_sq.md 2>&1
find $O -name '*.db' -size +8M -delete
cat $O/pmc_sq.md
