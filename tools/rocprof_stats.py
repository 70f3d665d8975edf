"""Summarise a rocprofv3 rocpd database (kernel-trace) as a per-kernel stats table (markdown/CSV-ish)."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print("| kernel | calls | total_us | avg_us | pct |")
print("|---|---:|---:|---:|---:|")
for n, c, t, a, p in rows:
    n = n.split("(")[0].replace("void ", "")
    if len(n) > 70: n = n[:67] + "..."
    print(f"| {n} | {c} | {t:.1f} | {a:.2f} | {p:.2f} |")
