#!/usr/bin/env python
"""Per-kernel average of one PMC counter from a rocprofv3 rocpd database:
    python scripts/rocpd_pmc.py <results.db> <COUNTER>   -> CSV kernel,dispatches,avg_value,avg_us"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
want = sys.argv[2]
views = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
if "counters_collection" in views:
    cols = [r[1] for r in cur.execute("pragma table_info(counters_collection)")]
    print("# counters_collection columns:", cols, file=sys.stderr)
    namec = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else cols[0])
    cname = "counter_name" if "counter_name" in cols else [c for c in cols if "counter" in c and "name" in c][0]
    valc = "value" if "value" in cols else [c for c in cols if "value" in c][0]
    durc = "(end - start)" if "start" in cols and "end" in cols else "0"
    q = (f"select {namec}, count(*), avg({valc}), avg({durc}) from counters_collection where {cname} = ? "
         f"group by {namec} order by 2*3 desc")
    rows = cur.execute(q, (want,)).fetchall()
else:
    rows = []
print("kernel,dispatches,avg_value,avg_us")
for n, c, v, d in rows:
    short = re.sub(r"\(.*", "", n)[:100]
    print('"%s",%d,%.1f,%.2f' % (short, c, v, (d or 0) / 1e3))
