#!/usr/bin/env bash
# Kernel timeline of one SCF step of the bench (run on the GPU box through gpurun): tools/step_timeline.sh [eigh|sp2]
# (eigh: a step of the timed main loop; sp2: a step of the bench's alt loop with the spectral-projection solver)
cd /tmp && export TMPDIR=/tmp
S=${1:-eigh}
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/timeline_$S -o a -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --steps 20 --no-w50 --no-real > $GRAFT_REPO_ROOT/gpurun_out/timeline_$S.log 2>&1
python3 - $GRAFT_REPO_ROOT/gpurun_out/timeline_$S $S <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*kernel_trace.csv')[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'k_exchange_W' in r['Kernel_Name']]
# bench: 3 warm-up + 20 timed steps (eigh), 3 stand-alone J builds, then 6 + 20 steps of the alt (sp2) loop
sel = (12, 17) if sys.argv[2] == "eigh" else (len(idx) - 8, len(idx) - 3)
a, b = idx[sel[0]], idx[sel[1]]
print("per step %.1f us" % ((int(rows[b]['Start_Timestamp']) - int(rows[a]['Start_Timestamp'])) / 1e3 / 5))
one = rows[idx[sel[0] + 1]:idx[sel[0] + 2]]
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
