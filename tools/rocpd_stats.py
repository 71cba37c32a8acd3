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

# --busy: how much of the traced span had at least one kernel running, and how many ran side by side (to stderr, so the CSV stays clean):
# the gaps between kernels longer than 1 ms are listed with what ran before and after them
if "--busy" in sys.argv:
    iv = cur.execute("select d.start, d.end, s.%s from %s d join %s s on d.kernel_id=s.id order by d.start" % (name_col, disp, sym)).fetchall()
    if iv:
        t0, t1 = iv[0][0], max(e for _, e, _ in iv)
        busy = 0
        cs, ce, last = iv[0][0], iv[0][1], iv[0][2]
        gaps = []
        for s_, e_, n_ in iv[1:]:
            if s_ > ce:
                busy += ce - cs
                if s_ - ce > 1e6: gaps.append((ce - t0, s_ - ce, last, n_))
                cs, ce = s_, e_
            else:
                ce = max(ce, e_)
            last = n_
        busy += ce - cs
        print("traced span %.1f ms, some kernel running %.1f ms (%.1f %%), sum of kernel time %.1f ms = %.2f kernels side by side while busy" %
              ((t1 - t0) / 1e6, busy / 1e6, 100.0 * busy / (t1 - t0), tot / 1e6, tot / busy), file=sys.stderr)
        for at, g, a, b in sorted(gaps, key=lambda x: -x[1])[:40]:
            print("  idle %.1f ms at %.1f ms: after %s, before %s" % (g / 1e6, at / 1e6, a[:50], b[:50]), file=sys.stderr)
