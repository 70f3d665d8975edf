"""Per-kernel mean of a PMC counter from rocprofv3 rocpd databases -> markdown + profiles/pmc_vq.json.

usage: python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_results.db gpurun_out/pmc_WRITE_SIZE/pmc_results.db
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md "HBM"); on gfx950 FETCH_SIZE counts a wide
coalesced read at HALF its bytes (128-B requests tallied at 64 B), so reads are reported both raw and x2.
"""
import json, sqlite3, sys, collections
res = collections.defaultdict(dict)
for path in sys.argv[1:]:
    db = sqlite3.connect(path)
    q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
    for k, c, v, n in db.execute(q):
        k = k.split("(")[0].replace("void ", "")
        res[k][c] = (v, n)
print("| kernel | launches | FETCH_SIZE KiB (raw) | read bytes (x2 gfx950 correction) | WRITE_SIZE KiB | write bytes |")
print("|---|---:|---:|---:|---:|---:|")
out = {}
for k, d in sorted(res.items(), key=lambda kv: -sum(x[0] for x in kv[1].values())):
    if not k.startswith("cgic::"): continue
    f = d.get("FETCH_SIZE", (0, 0)); w = d.get("WRITE_SIZE", (0, 0))
    print(f"| {k} | {max(f[1], w[1])} | {f[0]:.1f} | {f[0]*1024*2:.0f} | {w[0]:.1f} | {w[0]*1024:.0f} |")
    out[k] = {"fetch_kib_raw": f[0], "read_bytes_x2": f[0] * 2048, "write_bytes": w[0] * 1024}
import os
if os.environ.get("CGIC_PMC_STEP_JSON"):            # the whole step's counted bytes (tools/gpu_profile.sh: the four-lane passes)
    step = {k.replace("cgic::", ""): int(v["read_bytes_x2"] + v["write_bytes"]) for k, v in out.items() if "vq_prepare" not in k}
    json.dump({"step_hbm_bytes": sum(step.values()), "step_hbm_bytes_by_kernel": step}, open(os.environ["CGIC_PMC_STEP_JSON"], "w"), indent=1)
    raise SystemExit(0)
name, vq = next(((k, v) for k, v in out.items() if "vq_filter_router_kernel" in k), (None, None))      # the launch of the timed step
if vq is None:
    name, vq = next(((k, v) for k, v in out.items() if "vq_filter_kernel" in k), (None, None))
if vq is None:
    name, vq = next(((k, v) for k, v in out.items() if "vq_mfma_kernel" in k), (None, None))
if vq:
    json.dump({"kernel": name.replace("cgic::", ""), "hbm_bytes_per_launch": int(vq["read_bytes_x2"] + vq["write_bytes"]),
               "read_bytes_x2": int(vq["read_bytes_x2"]), "write_bytes": int(vq["write_bytes"]),
               "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, KiB x 1024, FETCH doubled per the gfx950 note in MI355X_MICROARCH.md; "
                       "mean over the launches of the timed step (B=64, 256x256)"}, open("profiles/pmc_vq.json", "w"), indent=1)
