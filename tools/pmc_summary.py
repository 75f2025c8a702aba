#!/usr/bin/env python3
"""Summarise the PMC counters of a rocprofv3 results.db for kernels matching a substring."""
import sqlite3, sys
db, pat = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else "nutls")
c = sqlite3.connect(db)
cols = [r[1] for r in c.execute("pragma table_info(counters_collection)")]
rows = c.execute("select * from counters_collection").fetchall()
ix = {n: i for i, n in enumerate(cols)}
agg = {}
for r in rows:
    name = r[ix.get("kernel_name", ix.get("name", 0))]
    if pat not in str(name): continue
    cn, val = r[ix["counter_name"]], r[ix["value"]]
    a = agg.setdefault(cn, [0, 0.0]); a[0] += 1; a[1] += val
for k, (n, v) in sorted(agg.items()):
    print("%-28s dispatches=%4d  mean=%.4g" % (k, n, v / n))
if not agg:
    print("columns:", cols); print(rows[:2])
