#!/usr/bin/env bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench / Fock build;
# summaries land in gpurun_out/<tag>/ and are copied into profiles/ by the developer.
set -u
TAG=${1:-r02}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-w50 --no-real > "$OUT/trace_bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fock" -- python3 tools/prof_fock.py C20H42 10 > "$OUT/trace_fock.log" 2>&1
tools/pmc_passes.sh "$OUT/pmc" C20H42 > "$OUT/pmc.log" 2>&1
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
def stats(d):
    f = glob.glob(d + "/*/*kernel_stats.csv")
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return [r for r in rows if "jcdf::" in r["Name"]] , rows
j, allrows = stats(out + "/trace_fock")
with open(out + "/kernel_stats_fock.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats -- python3 tools/prof_fock.py C20H42 10   (jcdf kernels only)\n")
    for r in j:
        fo.write("%-60s calls %4s  avg %10.1f us  min %10.1f  max %10.1f\n" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
jb, allb = stats(out + "/trace_bench")
with open(out + "/kernel_stats_bench.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-w50 --no-real  (top 25 by total time)\n")
    for r in sorted(allb, key=lambda r: -float(r["TotalDurationNs"]))[:25]:
        fo.write("%-70s calls %5s  total %9.2f ms  avg %10.1f us\n" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
# PMC traffic per launch (KB counters; FETCH_SIZE x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH HBM section)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "jcdf::" in k: agg[k.replace("jcdf::", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the record is only valid for the kernel sources and the shape it was measured on: bench.py checks both
tr = {"csrc_sha256_16": bench.csrc_hash(), "shape": [510, 1950, 81], "command": "tools/pmc_passes.sh (tools/prof_fock.py C20H42 3)"}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        key = "k_exchange_W" if k.startswith("k_exchange_W") else k       # the W pass: register- or DMA-staged kernel
        tr[key] = {"kernel_name": k, "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                 "hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                 "note": "2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of 16-B/lane coalesced reads)",
                 "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "grbm_gui_active": m.get("GRBM_GUI_ACTIVE"),
                 "lds_bank_conflict": m.get("SQ_LDS_BANK_CONFLICT"), "tcc_hit": m.get("TCC_HIT"), "tcc_miss": m.get("TCC_MISS")}
json.dump(tr, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/kernel_stats_fock.txt").read()); print(json.dumps(tr, indent=1)[:1500])
PY
