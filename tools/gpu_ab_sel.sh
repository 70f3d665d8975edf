#!/bin/bash
# A/B of router-select variants in ONE gpurun call: usage gpu_ab_sel.sh test_lib name [name ...]  -- tmp_libs/lib_<name>.so
cd "$GRAFT_REPO_ROOT" || exit 1
t=$1; shift
CGIC_LIB=$PWD/tmp_libs/lib_$t.so timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "router or mask or refine or flat or tie" 2>&1 | tail -2
bash tools/gpu_ab.sh "$@" | grep fused
for n in "$@"; do echo == $n; CGIC_LIB=$PWD/tmp_libs/lib_$n.so timeout 200 python tools/probes/probe_b1.py 1 2>&1 | grep -E "router_alone|fused_launch:|whole"; done
if [ -f tmp_libs/lib_${t}dbg.so ]; then CGIC_LIB=$PWD/tmp_libs/lib_${t}dbg.so timeout 300 python tools/probes/probe_b1_phases.py 1 2>&1 | grep -E "router_alone"; fi
