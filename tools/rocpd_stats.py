#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd SQLite result (kernel-trace): per-kernel calls / total / average / min / max duration.
usage: rocpd_stats.py results.db [> summary.csv]"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
disp = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "kernel_symbol" in t][0]
cols = [r[1] for r in cur.execute("pragma table_info(%s)" % disp)]
scols = [r[1] for r in cur.execute("pragma table_info(%s)" % sym)]
name_col = "display_name" if "display_name" in scols else ("kernel_name" if "kernel_name" in scols else scols[1])
q = ("select s.%s, count(*), sum(d.end-d.start), avg(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s on d.kernel_id=s.id "
     "group by s.%s order by 3 desc") % (name_col, disp, sym, name_col)
rows = cur.execute(q).fetchall()
tot = sum(r[2] for r in rows) or 1
print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage")
for r in rows:
    print("\"%s\",%d,%d,%.0f,%d,%d,%.2f" % (r[0][:90], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / tot))
