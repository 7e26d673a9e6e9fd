#!/usr/bin/env python3
"""Per-kernel time of ONE steady SCF step from a rocprofv3 kernel trace of tools/scf_steps.py (the step between the last two
k_exchange_W launches).  usage: step_breakdown.py <dir with *kernel_trace.csv>"""
import collections, csv, glob, sys
d = sys.argv[1]
f = (glob.glob(d + "/*kernel_trace.csv") + glob.glob(d + "/*/*kernel_trace.csv"))[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_exchange_W" in r["Kernel_Name"]]
a, b = idx[-2], idx[-1]
one = rows[a:b]
t0 = int(one[0]["Start_Timestamp"])
agg = collections.OrderedDict()
for r in one:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "").replace("jcdf::", "")[:60]
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    c = agg.setdefault(n, [0, 0.0, (int(r["Start_Timestamp"]) - t0) / 1e3])
    c[0] += 1
    c[1] += dur
print("step %.1f us, %d launches" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, len(one)))
for n, (cnt, tot, first) in agg.items():
    print("%9.1f first at | %4d x | %9.1f us total | %s" % (first, cnt, tot, n))
