#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r3f; mkdir -p $O
timeout 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest.txt
cat $O/pytest.txt
timeout 900 python bench.py --steps 20 --warmup 5 2>$O/bench20.err > $O/bench20.json; tail -3 $O/bench20.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r3f/bench20.json"))
print("value", d["value"], d["ms_per_step"], "roofline", {k: d["roofline"][k] for k in ("achieved","frac","vq_alone_frac","traffic")})
print("cpu", {k: d["cpu_baseline"][k] for k in ("value","cores","cpu_baseline_1core")})
print("stages", d["stages_us"])
mm=d["mask_mismatch"]; print({k: v for k, v in mm.items() if k != "tie_heavy_content"})
for k, v in mm["tie_heavy_content"].items(): print(" ", k, v if k == "note" or "error" in str(v) else {a: v[a] for a in ("differing_mask_elements","images_with_a_difference","differing_bin_files","bin_files","max_abs_entropy_diff")})
print("one_batch", d["one_batch_in_flight"], d["single_batch"])
PY
