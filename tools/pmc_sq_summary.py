"""Per-kernel mean of every PMC counter in one or more rocprofv3 rocpd databases (markdown table).
usage: python tools/pmc_sq_summary.py [--match SUBSTR] db1 [db2 ...]
Counters are summed over all instances (XCDs/SEs/dimensions) of one dispatch, then averaged over dispatches."""
import sqlite3, sys, collections
args = sys.argv[1:]
as_json = False
if args and args[0] == "--json":
    as_json = True; args = args[1:]
match = "cgic::"
if args and args[0] == "--match":
    match = args[1]; args = args[2:]
res = collections.defaultdict(dict)
for path in args:
    db = sqlite3.connect(path)
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    disp = "dispatch_id" if "dispatch_id" in cols else None
    if disp:
        q = f"select kernel_name, counter_name, {disp}, sum(value) from counters_collection group by kernel_name, counter_name, {disp}"
        acc = collections.defaultdict(list)
        for k, c, d, v in db.execute(q):
            acc[(k, c)].append(v)
        for (k, c), vs in acc.items():
            k = k.split("(")[0].replace("void ", "")
            res[k][c] = (sum(vs) / len(vs), len(vs))
    else:
        q = "select kernel_name, counter_name, avg(value), count(*) from counters_collection group by kernel_name, counter_name"
        for k, c, v, n in db.execute(q):
            k = k.split("(")[0].replace("void ", "")
            res[k][c] = (v, n)
if as_json:
    import json
    print(json.dumps({k: {c: v[0] for c, v in d.items()} for k, d in res.items() if match in k}))
    sys.exit(0)
for k, d in res.items():
    if match not in k:
        continue
    print(f"### {k}")
    print("| counter | mean per dispatch | dispatches |")
    print("|---|---:|---:|")
    for c in sorted(d):
        print(f"| {c} | {d[c][0]:.1f} | {d[c][1]} |")
    print()
