"""From a rocprofv3 --kernel-trace rocpd database: kernel intervals of the last `n` dispatches, their overlaps.
usage: python tools/trace_overlap.py results.db [n]"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
view = "kernels" if "kernels" in tabs else None
if view is None:
    print(tabs); sys.exit(1)
cols = [r[1] for r in db.execute(f"pragma table_info({view})")]
rows = db.execute(f"select name, start, end, queue_id, stream_id from {view} order by start").fetchall() if "queue_id" in cols and "stream_id" in cols else \
       [(r[0], r[1], r[2], 0, 0) for r in db.execute(f"select name, start, end from {view} order by start")]
rows = rows[-n:]
t0 = rows[0][1]
prev_end = None
for name, s, e, q, st in rows:
    name = name.split("(")[0].replace("void ", "").replace("cgic::", "")[:44]
    ov = ""
    if prev_end is not None and s < prev_end:
        ov = f"  overlaps previous by {(prev_end - s) / 1e3:.1f} us"
    print(f"{(s - t0) / 1e3:9.1f} -> {(e - t0) / 1e3:9.1f} us  ({(e - s) / 1e3:6.1f})  q{q} s{st}  {name}{ov}")
    prev_end = e if prev_end is None else max(prev_end, e)
span = (max(r[2] for r in rows) - t0) / 1e3
busy = sum(r[2] - r[1] for r in rows) / 1e3
print(f"window {span:.1f} us, sum of kernel durations {busy:.1f} us")
