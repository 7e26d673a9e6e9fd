#!/usr/bin/env bash
# Kernel timeline of one SCF step of the bench (run on the GPU box through gpurun): tools/step_timeline.sh [eigh|sp2]
cd /tmp && export TMPDIR=/tmp
S=${1:-eigh}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/timeline_$S -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --density-solver $S --steps 20 --no-w50 --no-real > $GRAFT_REPO_ROOT/gpurun_out/timeline_$S.log 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/timeline_$S <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_exchange_W' in r['Kernel_Name']]
a, b = idx[-8], idx[-3]
print("per step %.1f us" % ((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / 5))
one = rows[idx[-4]:idx[-3]]
t0 = int(one[0]['Start_Timestamp'])
skip = 0
for r in one:
    n = r['Kernel_Name'].split('(')[0].replace('void ', '')[:70]
    if 'k_sp2_fused' in n or 'k_dc_' in n:
        skip += 1
        continue
    print("%8.1f  dur %7.1f  %s" % ((int(r['Start_Timestamp']) - t0) / 1e3, (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, n))
print("(%d k_sp2_fused / k_dc_* launches not listed)" % skip)
PY
