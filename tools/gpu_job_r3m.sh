#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3m; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "vq or prepared or lane_stream or fused or f2_f4 or custom_ops or nonfinite or near or stress" 2>&1 | tail -12 | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt
CGIC_LIB=$GRAFT_REPO_ROOT/control-gic_amd/libcgic_hip_dbg.so timeout 300 python tools/probe_vq_phases.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/vq_phases.txt
for i in 1 2 3; do timeout 300 python tools/run_roofline_cmd.py fused 2>&1 | grep HIP; timeout 300 python tools/run_roofline_cmd.py vq 2>&1 | grep HIP; done | tee $O/alone.txt
timeout 600 python tools/probe_vq_variant.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/variant.txt
