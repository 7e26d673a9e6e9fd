#!/usr/bin/env python3
"""Effective shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration) of the Fock-build kernels inside bench.py's SCF loop, from
one `rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace` pass.  usage: python tools/loop_clock.py <rocprof output dir>"""
import collections, csv, glob, sys
d = sys.argv[1]
cc = glob.glob(d + "/*/*counter_collection.csv")[0]
kt = glob.glob(d + "/*/*kernel_trace.csv")[0]
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(list)
for r in csv.DictReader(open(cc)):
    k = r["Kernel_Name"].split("(")[0]
    if ("k_exchange" in k or "k_coulomb" in k or "sytrd" in k) and r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        t = dur.get(r["Dispatch_Id"])
        if t:
            agg[k[:60]].append((float(r["Counter_Value"]) / 8.0 / t, t / 1e3))
for k, v in agg.items():
    v = v[3:]
    print("%-62s n=%d  clock %.3f GHz  dur %.1f us" % (k, len(v), sum(x[0] for x in v) / len(v), sum(x[1] for x in v) / len(v)))
