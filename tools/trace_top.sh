#!/usr/bin/env bash
# rocprofv3 kernel trace of a python tool run (GPU box): tools/trace_top.sh <tag> <tool.py> [args...]; prints the top kernels by total time
cd /tmp && export TMPDIR=/tmp
TAG=$1; shift
OUT=$GRAFT_REPO_ROOT/gpurun_out/trace_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o a -- python3 $GRAFT_REPO_ROOT/"$@" > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_stats.csv')
rows = list(csv.DictReader(open(f[0])))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("%-72s %7s %10s %9s %6s" % ("kernel", "calls", "total ms", "avg us", "%"))
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"]))[:28]:
    print("%-72s %7s %10.2f %9.1f %6.1f" % (r["Name"].split("(")[0].replace("void ", "")[:72], r["Calls"], float(r["TotalDurationNs"]) / 1e6,
                                          float(r["AverageNs"]) / 1e3, 100 * float(r["TotalDurationNs"]) / tot))
PY
