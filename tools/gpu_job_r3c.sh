#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3c; mkdir -p $O
export CGIC_LIB=$GRAFT_REPO_ROOT/control-gic_amd/libcgic_hip_dbg.so
for v in "" idx; do echo "== $v"; timeout 300 python tools/probe_vq_phases.py $v 2>&1 | grep -v "Warning\|amdgpu.ids"; done | tee $O/vq_phases.txt
