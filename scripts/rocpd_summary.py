#!/usr/bin/env python
"""Summarise a rocprofv3 (rocpd SQLite) kernel trace: per-kernel calls / total / avg / % - the
`--stats` table - as CSV on stdout.   python scripts/rocpd_summary.py <results.db> [min_pct]"""
import re
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
name_col = "name" if "name" in cols else [c for c in cols if "name" in c][0]
rows = cur.execute(f"select {name_col}, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start) "
                   f"from kernels group by {name_col} order by 3 desc").fetchall()
total = sum(r[2] for r in rows)
print("kernel,calls,total_ms,avg_us,min_us,max_us,pct")
for n, c, t, a, mn, mx in rows:
    short = re.sub(r"\(.*", "", n)[:110]
    print(f"\"{short}\",{c},{t / 1e6:.3f},{a / 1e3:.2f},{mn / 1e3:.2f},{mx / 1e3:.2f},{100.0 * t / total:.2f}")
print(f"\"TOTAL\",{sum(r[1] for r in rows)},{total / 1e6:.3f},,,,100.00")
