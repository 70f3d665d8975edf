#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3o; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu -k "decomp or decode or codec or compress or roundtrip or adversarial or tables" 2>&1 | tail -4 | cut -c1-300 > $O/pytest.txt
cat $O/pytest.txt
timeout 300 python tools/stress_codec.py 20 40 throughput 2>&1 | tail -3
CGIC_LIB=$GRAFT_REPO_ROOT/control-gic_amd/libcgic_hip_dbg.so timeout 300 python tools/probe_decode_ss.py 2>&1 | grep -v "Warning\|amdgpu.ids" | tee $O/ss.txt
bash tools/gpu_job_r3l.sh 2>&1 | grep decode_image
for k in "20 5" "20 5" "200 20"; do set -- $k
timeout 600 python bench.py --steps $1 --warmup $2 --no-report 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=$1', d['value'], d['ms_per_step'])"
done | tee $O/bench.txt
