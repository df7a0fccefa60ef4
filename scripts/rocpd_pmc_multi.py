#!/usr/bin/env python
"""Per-kernel averages of EVERY counter in one rocprofv3 rocpd database (one --pmc pass):
    python scripts/rocpd_pmc_multi.py <results.db> [min_dispatches]  -> CSV kernel,dispatches,avg_us,<COUNTER>...
Counter values are summed over their dimensions (XCC / SE / instance) per dispatch, then averaged over dispatches."""
import re
import sqlite3
import sys
from collections import defaultdict

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
min_disp = int(sys.argv[2]) if len(sys.argv) > 2 else 1
cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
namec = "kernel_name" if "kernel_name" in cols else "name"
cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
valc = "value" if "value" in cols else [c for c in cols if "value" in c][0]
dispc = "dispatch_id" if "dispatch_id" in cols else None
durc = "(end - start)" if "start" in cols and "end" in cols else "0"
if dispc:   # sum the dimensions of a counter inside a dispatch first
    q = (f"select {namec}, {cname}, {dispc}, sum({valc}), max({durc}) from counters_collection "
         f"group by {namec}, {cname}, {dispc}")
else:
    q = f"select {namec}, {cname}, 0, {valc}, {durc} from counters_collection"
acc = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for n, c, d, v, t in cur.execute(q):
    short = re.sub(r"\(.*", "", n)[:90]
    acc[short][c].append(float(v))
    dur[short].append((t or 0) / 1e3)
counters = sorted({c for k in acc for c in acc[k]})
print("kernel,dispatches,avg_us," + ",".join(counters))
for k in sorted(acc, key=lambda k: -sum(dur[k])):
    nd = max(len(v) for v in acc[k].values())
    if nd < min_disp:
        continue
    row = [f'"{k}"', str(nd), f"{sum(dur[k]) / max(1, len(dur[k])):.2f}"]
    for c in counters:
        v = acc[k].get(c, [])
        row.append(f"{sum(v) / len(v):.1f}" if v else "")
    print(",".join(row))
