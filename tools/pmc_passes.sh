#!/usr/bin/env bash
# PMC passes over tools/prof_fock.py (one counter group per run; never combined with
# tracing domains other than --kernel-trace).  usage: tools/pmc_passes.sh <outdir> [config]
set -u
OUT=${1:-gpurun_out/pmc}; CFG=${2:-C20H42}
export TMPDIR=/tmp
export JCDF_PROF_NO_OVERLAP=1      # J one after K (not beside it on the side stream): the cycle counters of a launch belong to that launch
mkdir -p "$OUT"
run() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 tools/prof_fock.py "$CFG" 3 > "$OUT/$name.log" 2>&1 || echo "pass $name failed"; }
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_WAIT_INST_LDS
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS
run tcc1 FETCH_SIZE GRBM_GUI_ACTIVE
run tcc2 WRITE_SIZE TCC_HIT TCC_MISS
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "jcdf::" not in k: continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
with open(out + "/summary.txt", "w") as fo:
    for k in sorted(agg):
        fo.write(k + "\n")
        for c in sorted(agg[k]):
            v = agg[k][c]
            fo.write("   %-34s n=%-3d mean=%.6g\n" % (c, len(v), sum(v) / len(v)))
print(open(out + "/summary.txt").read())
PY
