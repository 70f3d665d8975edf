"""profiles/<round>_roofline.json from the round's profiling passes (tools/gpu_profile.sh): the ONE table bench.py's roofline line is
computed from.  usage: python tools/roofline_json.py OUTDIR ROUND  (reads OUTDIR/loop_lanes1.json, loop_lanes4.json, alone.db, pmc_hbm.json)"""
import json, os, sqlite3, sys
O = sys.argv[1]
R = sys.argv[2] if len(sys.argv) > 2 else "r05"
dom = "vq_filter_router_kernel"
out = {"kernel": dom + "<true, false> (VQ forward + the per-image router workgroups: the launch of the timed step)",
       "workload": "B=64 of 256x256: N = 262144 latent vectors, K = 1024, D = 4", "flops_per_launch": 2.0 * 262144 * 1024 * 4,
       "peak_TFLOPs": 157.3}
for L in (1, 4):
    d = json.load(open(os.path.join(O, f"loop_lanes{L}.json")))
    k = next(v for n, v in d["kernels"].items() if dom in n)
    out[f"rocprof_avg_us_lanes{L}_loop"] = round(k["avg_us"], 3)
    out[f"launches_lanes{L}_loop"] = k["launches"]
    out[f"us_per_step_under_profiler_lanes{L}"] = round(d["us_per_step_under_profiler"], 3)
    out[f"all_kernels_lanes{L}_loop_avg_us"] = {n: round(v["avg_us"], 3) for n, v in d["kernels"].items()}
db = sqlite3.connect(os.path.join(O, "alone.db"))
rows = [r for r in db.execute("select name, start, end from kernels order by start") if dom in r[0]]
rows = rows[-100:]                                   # the 5 timed replays of the 20-launch graph
out["rocprof_avg_us_alone_graph"] = round(sum(e - s for _, s, e in rows) / len(rows) / 1e3, 3)
out["launches_alone_graph"] = len(rows)
hip = [ln for ln in open(os.path.join(O, "alone.log")) if "HIP events" in ln]
if hip:
    out["hip_events_us_alone_graph_same_run"] = float(hip[-1].split(":")[-1].split("us")[0])
pm = os.path.join(O, "pmc_hbm.json")
if os.path.exists(pm):
    out.update({k: v for k, v in json.load(open(pm)).items() if k in ("hbm_bytes_per_launch", "read_bytes_x2", "write_bytes")})
ps = os.path.join(O, "pmc_step.json")
if os.path.exists(ps):
    out.update(json.load(open(ps)))
out["frac_lanes1_loop"] = round(out["flops_per_launch"] / (out["rocprof_avg_us_lanes1_loop"] * 1e-6) / 1e12 / 157.3, 4)
out["frac_alone_graph"] = round(out["flops_per_launch"] / (out["rocprof_avg_us_alone_graph"] * 1e-6) / 1e12 / 157.3, 4)
# matrix-core counters of the same command (tools/gpu_profile.sh, pmcf_* passes): busy fraction of the 1024 SIMDs' MFMA pipes
pj = os.path.join(O, "pmc_sq_vq_fused.json")
if os.path.exists(pj):
    try:
        c = next(iter(json.load(open(pj)).values()))
        busy, insts = c.get("SQ_VALU_MFMA_BUSY_CYCLES"), c.get("SQ_INSTS_MFMA")
        dur_cycles = out["rocprof_avg_us_alone_graph"] * 1e-6 * 2.4e9
        out["mfma"] = {"SQ_INSTS_MFMA": insts, "SQ_VALU_MFMA_BUSY_CYCLES": busy, "SQ_VALU_MFMA_COEXEC_CYCLES": c.get("SQ_VALU_MFMA_COEXEC_CYCLES"),
                       "SQ_INSTS_VALU": c.get("SQ_INSTS_VALU"), "SQ_WAVE_CYCLES": c.get("SQ_WAVE_CYCLES"), "SQ_WAIT_ANY": c.get("SQ_WAIT_ANY"),
                       "SQ_LDS_BANK_CONFLICT": c.get("SQ_LDS_BANK_CONFLICT"), "SQ_LDS_IDX_ACTIVE": c.get("SQ_LDS_IDX_ACTIVE"),
                       "mfma_busy_frac": round(busy / 1024.0 / dur_cycles, 4) if busy else None,
                       "note": "SQ_VALU_MFMA_BUSY_CYCLES summed over the chip / 1024 SIMDs / (launch duration x 2.4 GHz): the share of the launch "
                               "during which a SIMD's matrix pipe was busy; 32 cycles per v_mfma_f32_32x32x16_f16"}
    except Exception as e:                      # noqa: BLE001
        out["mfma"] = {"error": str(e)[:200]}
out["commands"] = {
    "lanes L loop": "rocprofv3 --kernel-trace --stats -- python bench.py --steps 96 --warmup 16 --no-report --lanes L   (tools/trace_concurrency.py: the last 96 five-launch chains)",
    "alone graph": "rocprofv3 --kernel-trace --stats -- python tools/run_roofline_cmd.py fused   (bench.graph_kernel_time: 20 launches per hipGraph, the last 100 launches)",
    "hbm": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE --kernel-trace -- python bench.py --steps 40 --warmup 10 --no-report --lanes 1 --no-graph   (tools/pmc_summary.py)"}
json.dump(out, open(os.path.join(O, f"{R}_roofline.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
