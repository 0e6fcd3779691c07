#!/usr/bin/env python3
"""Per-kernel average of PMC counters from a rocprofv3 rocpd sqlite db.
usage: tools/rocpd_pmc.py <db> [name-substring]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
views = [r[0] for r in cur.execute("select name from sqlite_master where type='view'")]
if "counters_collection" not in views:
    raise SystemExit("no counters_collection view in db")
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
print("# columns:", cols, file=sys.stderr)
kn = "kernel_name" if "kernel_name" in cols else [c for c in cols if "kernel" in c and "name" in c][0]
cn = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
cv = "value" if "value" in cols else [c for c in cols if "value" in c][0]
rows = cur.execute(f"select {kn}, {cn}, count(*), avg({cv}), sum({cv}) from counters_collection group by {kn}, {cn}").fetchall()
print("| kernel | counter | dispatches | avg per dispatch | total |")
print("|---|---|---:|---:|---:|")
for k, c, n, a, s in rows:
    if flt in k:
        k = k if len(k) < 80 else k[:77] + "..."
        print(f"| `{k}` | {c} | {n} | {a:.4g} | {s:.6g} |")
