"""From a rocprofv3 --kernel-trace rocpd database of bench.py: the timed loop's per-kernel durations (as dispatched on the lane
streams) and how many kernels ran at the same time.  usage: python tools/trace_concurrency.py results.db steps [out.json]"""
import json, sqlite3, sys, collections
db = sqlite3.connect(sys.argv[1]); steps = int(sys.argv[2])
rows = db.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
short = lambda n: n.split("(")[0].replace("void ", "").replace("cgic::", "")[:34]
cg = [r for r in rows if "cgic::" in r[0] and ("entropy_maps" in r[0] or "vq_filter_router" in r[0] or "compress_streams" in r[0] or "decode_" in r[0] or "merge_kernel" in r[0])]
# the timed loop = the last `steps` five-launch chains (entropy -> VQ+router -> compress -> decode -> merge on ONE stream); the stage
# breakdown that follows launches each kernel many times in a row and does not match
order = ["entropy_maps", "vq_filter_router", "compress_streams", "decode_", "merge_kernel"]
by_stream = collections.defaultdict(list)
for r in cg: by_stream[r[4]].append(r)
chains = []
for st, rs in by_stream.items():
    i = 0
    while i + 5 <= len(rs):
        if all(order[k] in rs[i + k][0] for k in range(5)):
            chains.append(rs[i:i + 5]); i += 5
        else:
            i += 1
chains.sort(key=lambda c: c[0][1])
chains = chains[-steps:]
loop = sorted((r for c in chains for r in c), key=lambda r: r[1])
t0 = loop[0][1]; t1 = max(r[2] for r in loop)
print(f"timed loop: {len(loop)} launches on streams {sorted(set(r[4] for r in loop))} (queues {sorted(set(r[3] for r in loop))}), span {(t1 - t0) / 1e3:.1f} us = {(t1 - t0) / 1e3 / steps:.2f} us per step (under the profiler)")
ev = []
for r in loop: ev.append((r[1], 1)); ev.append((r[2], -1))
ev.sort(); c = 0; last = t0; hist = collections.Counter()
for t, d in ev:
    hist[c] += t - last; last = t; c += d
tot = sum(hist.values())
print("kernels running at the same time (share of the span): " + ", ".join(f"{k}: {v / tot:.1%}" for k, v in sorted(hist.items())))
d = collections.defaultdict(list)
for r in loop: d[short(r[0])].append((r[2] - r[1]) / 1e3)
print("| kernel | launches | avg duration in the loop (us) |"); print("|---|---:|---:|")
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])): print(f"| {k} | {len(v)} | {sum(v) / len(v):.2f} |")
print(f"sum of the average durations {sum(sum(v) / len(v) for v in d.values()):.1f} us per batch")
if len(sys.argv) > 3:
    json.dump({"steps": steps, "us_per_step_under_profiler": (t1 - t0) / 1e3 / steps,
               "kernels": {k: {"launches": len(v), "avg_us": sum(v) / len(v)} for k, v in d.items()}}, open(sys.argv[3], "w"), indent=1)
