#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3s; mkdir -p $O
timeout 1500 python bench.py --steps 20 --warmup 5 2>$O/bench20.err > $O/bench20.json; tail -2 $O/bench20.err
timeout 600 python bench.py 2>/dev/null --no-extra --no-cpu-baseline > $O/bench200.json
for k in "2000 20" "20 5" "20 5"; do set -- $k
timeout 600 python bench.py --steps $1 --warmup $2 --no-report 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('K=$1', d['value'], d['ms_per_step'])"
done | tee $O/bench_more.txt
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3s/bench20.json"))
print("value", d["value"], d["ms_per_step"], "rccl", d["rccl_ranks"], d["histogram_allreduce_us"])
print("roofline", {k: v for k, v in d["roofline"].items() if k not in ("note", "kernel", "duration_source")})
print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","cpu_baseline_1core")})
print("stages", d["stages_us"])
mm=d["mask_mismatch"]; print({k: v for k, v in mm.items() if k not in ("tie_heavy_content","note")})
for k, v in mm["tie_heavy_content"].items():
    if k == "reference_order_mode": print("  ref mode", {a: b for a, b in v.items() if a != "note"})
    elif k != "note": print(" ", k, {a: v[a] for a in ("differing_mask_elements","images_with_a_difference","differing_bin_files","bin_files","max_abs_entropy_diff")})
print("one_batch", d["one_batch_in_flight"]["ms_per_step"], d["single_batch"]["ms_per_step"], "bpp", d["bpp"], d["bpp_match"])
print("ratio_sweep", [(r["ratio"], r["mode"], r["MPixels/s"], r["MPixels/s_4_in_flight"], r["bpp_match"]) for r in d["ratio_sweep"]])
print("b1", d["b1_latency"]); print("div2k", {k: (v if not isinstance(v, dict) else {a: b for a, b in v.items() if a != "note"}) for k, v in d["div2k_image"].items() if k != "workload"})
print("tiles", [(t["value"], t["MPixels/s_4_in_flight"], t["bpp_match"]) for t in d["div2k_tiles"]])
print("e2e", {k: v for k, v in d["end_to_end_estimate"].items() if k != "note"})
d2=json.load(open("gpurun_out/r3s/bench200.json")); print("K=200 default", d2["value"], d2["ms_per_step"], d2["roofline"]["frac"], d2["roofline"]["frac_hip_events"])
PY
