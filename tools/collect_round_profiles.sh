#!/usr/bin/env bash
# Run on the GPU box (via gpurun): kernel-trace stats + PMC passes of the bench / Fock build and the per-step kernel record;
# summaries land in gpurun_out/<tag>/ and are copied into profiles/<tag>_* by the developer (tools/collect_round_profiles.sh r04
# && cp gpurun_out/r04/{kernel_stats_*.txt,pmc_traffic.json,pmc_summary.txt,step_kernels.json,step_timeline_eigh.txt} profiles/).
set -u
TAG=${1:-r04}
OUT=gpurun_out/$TAG
export TMPDIR=/tmp
mkdir -p "$OUT"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_bench" -o a -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-w50 --no-real > "$OUT/trace_bench.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_fock" -o a -- python3 tools/prof_fock.py C20H42 10 > "$OUT/trace_fock.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_gly10" -o a -- python3 tools/scf_steps.py gly10_vtz 0.30 5 > "$OUT/trace_gly10.log" 2>&1
tools/pmc_passes.sh "$OUT/pmc" C20H42 > "$OUT/pmc.log" 2>&1
cp "$OUT/pmc/summary.txt" "$OUT/pmc_summary.txt" 2>/dev/null
python3 - "$OUT" "$TAG" <<'PY'
import csv, glob, json, os, re, sys, collections
out, tag = sys.argv[1], sys.argv[2]
sys.path.insert(0, os.getcwd())
import bench
def stats(d):
    f = glob.glob(d + "/*kernel_stats.csv") + glob.glob(d + "/*/*kernel_stats.csv")
    rows = list(csv.DictReader(open(f[0]))) if f else []
    return [r for r in rows if "jcdf::" in r["Name"]], rows
j, allrows = stats(out + "/trace_fock")
with open(out + "/kernel_stats_fock.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats -- python3 tools/prof_fock.py C20H42 10   (jcdf kernels only; back-to-back Fock builds)\n")
    for r in j:
        fo.write("%-60s calls %4s  avg %10.1f us  min %10.1f  max %10.1f\n" % (r["Name"].split("(")[0][:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
jb, allb = stats(out + "/trace_bench")
with open(out + "/kernel_stats_bench.txt", "w") as fo:
    fo.write("# rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-w50 --no-real  (top 30 by total time)\n")
    for r in sorted(allb, key=lambda r: -float(r["TotalDurationNs"]))[:30]:
        fo.write("%-70s calls %5s  total %9.2f ms  avg %10.1f us\n" % (r["Name"].split("(")[0][:70], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e3))
# ---- one timed step of the bench: kernel list, launches by family (step_kernels.json, step_timeline_eigh.txt) ----------------
tf = glob.glob(out + "/trace_bench/*kernel_trace.csv") + glob.glob(out + "/trace_bench/*/*kernel_trace.csv")
rows = sorted(csv.DictReader(open(tf[0])), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_exchange_W" in r["Kernel_Name"]]
# bench: 3 warm-up + 20 timed steps of the main (eigh) loop come first
a, b = idx[12], idx[13]
one = rows[a:b]
t0 = int(one[0]["Start_Timestamp"])
fam = collections.Counter()
small_ms, lines = 0.0, []
BIG = ("k_exchange_W", "k_exchange_K", "k_coulomb_J", "k_fock_assemble", "k_sytrd", "k_sytd2_tail", "k_q_tail_reflect", "k_dc_", "k_reduce_V", "k_prep_C")
for r in one:
    n = r["Kernel_Name"].split("(")[0].replace("void ", "")
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    f = "library" if "jcdf::" in n else ("vendor" if re.search(r"rocblas|rocsolver|Cijk_|hipblaslt|Tensile", n) else "torch")
    fam[f] += 1
    if not any(k in n for k in BIG):
        small_ms += dur / 1e3
    if "k_dc_" not in n:
        lines.append("%8.1f  dur %7.1f  %s" % ((int(r["Start_Timestamp"]) - t0) / 1e3, dur, n[:70]))
step_us = (int(rows[idx[17]]["Start_Timestamp"]) - int(rows[idx[12]]["Start_Timestamp"])) / 1e3 / 5
with open(out + "/step_timeline_eigh.txt", "w") as fo:
    fo.write("per step %.1f us  (rocprofv3 --kernel-trace of bench.py, one step of the timed loop; k_dc_* launches not listed: %d)\n" % (
        step_us, sum(1 for r in one if "k_dc_" in r["Kernel_Name"])))
    fo.write("\n".join(lines) + "\n")
json.dump({"csrc_sha256_16": bench.csrc_hash(), "shape": [510, 1950, 81], "command": "rocprofv3 --kernel-trace -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-w50 --no-real",
           "step_us": step_us, "launches_per_step": sum(fam.values()), "library_kernels_per_step": fam["library"], "torch_kernels_per_step": fam["torch"],
           "vendor_kernels_per_step": fam["vendor"], "small_launch_ms_per_step": small_ms,
           "small_launch_note": "everything outside W / K / J / reduce_V / prep_C / assemble / sytrd (+ tail) / D&C",
           "kernels": [r["Kernel_Name"].split("(")[0].replace("void ", "")[:80] for r in one]}, open(out + "/step_kernels.json", "w"), indent=1)
# ---- one SCF step of the gly10 shape (BASELINE config 5: N = 1915, above the size whose Q fits the tridiagonalisation kernel) -----
tg = glob.glob(out + "/trace_gly10/*kernel_trace.csv") + glob.glob(out + "/trace_gly10/*/*kernel_trace.csv")
if tg:
    rows5 = sorted(csv.DictReader(open(tg[0])), key=lambda r: int(r["Start_Timestamp"]))
    idx5 = [i for i, r in enumerate(rows5) if "k_exchange_W" in r["Kernel_Name"]]
    one5 = rows5[idx5[-2]:idx5[-1]]                      # the second-to-last step, Fock build to Fock build
    fam5, tot5 = collections.Counter(), collections.Counter()
    for r in one5:
        n = r["Kernel_Name"].split("(")[0].replace("void ", "")
        f = "library" if "jcdf::" in n else ("vendor" if re.search(r"rocblas|rocsolver|Cijk_|hipblaslt|Tensile", n) else "torch")
        fam5[f] += 1
        tot5[n.split("<")[0]] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    step5 = (int(rows5[idx5[-1]]["Start_Timestamp"]) - int(rows5[idx5[-2]]["Start_Timestamp"])) / 1e3
    json.dump({"csrc_sha256_16": bench.csrc_hash(), "shape": [1915, 5261, 155], "command": "rocprofv3 --kernel-trace -- python3 tools/scf_steps.py gly10_vtz 0.30 5",
               "step_us": step5, "launches_per_step": sum(fam5.values()), "library_kernels_per_step": fam5["library"], "torch_kernels_per_step": fam5["torch"],
               "vendor_kernels_per_step": fam5["vendor"], "kernel_us_per_step": dict(sorted(tot5.items(), key=lambda kv: -kv[1])[:24])},
              open(out + "/step_kernels_gly10.json", "w"), indent=1)
    print(open(out + "/step_kernels_gly10.json").read()[:1800])
# ---- PMC per launch (KB counters; FETCH_SIZE x2 on gfx950 for wide coalesced reads, MI355X_MICROARCH HBM section) ----------
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc/*/*/*counter_collection.csv") + glob.glob(out + "/pmc/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "jcdf::" in k: agg[k.replace("jcdf::", "").split("<")[0]][r["Counter_Name"]].append(float(r["Counter_Value"]))
# the record is only valid for the kernel sources and the shape it was measured on: bench.py checks both
tr = {"csrc_sha256_16": bench.csrc_hash(), "shape": [510, 1950, 81], "command": "tools/pmc_passes.sh (tools/prof_fock.py C20H42 3, JCDF_PROF_NO_OVERLAP=1: J after K, every kernel alone)"}
for k, c in agg.items():
    m = {n: sum(v) / len(v) for n, v in c.items()}
    if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
        key = "k_exchange_W" if k.startswith("k_exchange_W") else k       # the W pass under its record name
        tr[key] = {"kernel_name": k, "FETCH_SIZE_KB": m["FETCH_SIZE"], "WRITE_SIZE_KB": m["WRITE_SIZE"],
                 "hbm_bytes_per_launch": (2.0 * m["FETCH_SIZE"] + m["WRITE_SIZE"]) * 1024.0,
                 "note": "2*FETCH_SIZE + WRITE_SIZE (gfx950: FETCH_SIZE counts half of 16-B/lane coalesced reads)",
                 "mfma_busy_cycles": m.get("SQ_VALU_MFMA_BUSY_CYCLES"), "grbm_gui_active": m.get("GRBM_GUI_ACTIVE"),
                 "lds_bank_conflict": m.get("SQ_LDS_BANK_CONFLICT"), "lds_idx_active": m.get("SQ_LDS_IDX_ACTIVE"),
                 "tcc_hit": m.get("TCC_HIT"), "tcc_miss": m.get("TCC_MISS")}
json.dump(tr, open(out + "/pmc_traffic.json", "w"), indent=1)
print(open(out + "/kernel_stats_fock.txt").read()); print(open(out + "/step_timeline_eigh.txt").read()); print(json.dumps(tr, indent=1)[:2500])
PY
