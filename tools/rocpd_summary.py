#!/usr/bin/env python3
"""Per-kernel summary (and optionally the timeline of the last repetition) from a rocprofv3 rocpd database.
usage: python tools/rocpd_summary.py results.db [--timeline first_kernel_substring]"""
import re
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "info_kernel_symbol" in t][0]
rows = list(cur.execute("select s.kernel_name, d.start, d.end from %s d join %s s on d.kernel_id = s.id order by d.start" % (kd, ks)))


def short(n):
    return re.sub(r"\(.*", "", n).replace("void ", "").replace("jcdf::", "")[:60]


agg = {}
for n, s, e in rows:
    a = agg.setdefault(short(n), [0, 0.0])
    a[0] += 1
    a[1] += (e - s) / 1e3
print("%-60s %7s %10s %9s" % ("kernel", "calls", "total us", "avg us"))
for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("%-60s %7d %10.1f %9.2f" % (n, c, t, t / c))
if "--timeline" in sys.argv:
    key = sys.argv[sys.argv.index("--timeline") + 1]
    idx = [i for i, r in enumerate(rows) if key in r[0]]
    a = idx[-1]
    t0 = rows[a][1]
    prev_end = t0
    for n, s, e in rows[a:]:
        print("%9.1f  gap %6.1f  dur %8.1f  %s" % ((s - t0) / 1e3, (s - prev_end) / 1e3, (e - s) / 1e3, short(n)))
        prev_end = e
