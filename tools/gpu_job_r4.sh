#!/bin/bash
# round 4: full -m gpu suite, then the driver's bench command (+ a K=200 / K=2000 headline)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-r4}
mkdir -p $O
if [ "$2" != "notest" ]; then
  timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/gputest.log
fi
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_k20.json 2> $O/bench_k20.err
for K in 200 2000; do
  timeout 300 python bench.py --steps $K --warmup 20 --no-report --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('K=$K', d['value'], d['ms_per_step'])" >> $O/headline.txt
done
cat $O/gputest.log; python - <<PY
import json
d=json.loads(open("$O/bench_k20.json").read().strip().splitlines()[-1])
print("K=20", d["value"], d["ms_per_step"], "bpp_match", d.get("bpp_match"), "roofline", d["roofline"]["frac"], d["roofline"].get("hip_events_us"))
print(json.dumps(d.get("stages_us"), indent=0)[:1500])
mm=d.get("mask_mismatch",{})
print({k:v for k,v in mm.items() if k!="tie_heavy_content"})
for k,v in mm.get("tie_heavy_content",{}).items(): print(k, v)
print("b1", d.get("b1_latency")); print("one_batch", d.get("one_batch_in_flight"))
PY
cat $O/headline.txt
