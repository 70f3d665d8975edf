#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r02b; mkdir -p $O
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_pipe.json 2> $O/bench_pipe.err; echo "rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline > $O/bench_pipe200.json 2>> $O/bench_pipe.err; echo "rc=$?"
timeout 300 python bench.py --steps 200 --warmup 20 --no-extra --no-cpu-baseline --schedule sequential > $O/bench_seq200.json 2>> $O/bench_pipe.err; echo "rc=$?"
tail -5 $O/bench_pipe.err
python - <<'PY'
import json
for f in ("bench_pipe","bench_pipe200","bench_seq200"):
    try:
        d=json.load(open(f"gpurun_out/r02b/{f}.json"))
    except Exception as e:
        print(f, "unreadable", e); continue
    print(f, d["value"], d["ms_per_step"], d.get("bpp_match"), d["config"]["launch"][:40])
    for k in ("single_batch","stages_us","roofline","mask_mismatch","b1_latency","div2k_image","div2k_tiles","cpu_baseline"):
        if k in d: print("  ",k, json.dumps(d[k])[:400])
    if "ratio_sweep" in d:
        for r in d["ratio_sweep"]: print("   sweep", r)
PY
