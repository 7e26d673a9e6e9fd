#!/usr/bin/env python3
"""Prints the headline fields of a bench.py JSON line.  usage: python tools/show_bench.py file.json"""
import json, sys
d = json.load(open(sys.argv[1]))
for k in ["value", "ms_per_step", "fock_build_ms", "replicated_ms", "allreduce_ms", "fock_build_useful_tflops", "fock_build_useful_pct_fp64_mfma_peak",
          "fock_build_tflops_dense_formula", "fock_build_pct_fp64_mfma_peak_dense_formula", "kernels_ms", "kernels_executed_tflops"]:
    print(k, d.get(k))
r = d["roofline"]
print("roofline", {k: r[k] for k in ["frac", "achieved", "launch_ms", "executed_tflops", "traffic", "traffic_source"]}, "J", r["hbm_stream"]["GBs"])
if d.get("alt"):
    print("alt", d["alt"]["value"], d["alt"].get("replicated_ms"))
for k, v in (d.get("scaling_w50") or {}).items():
    if isinstance(v, dict):
        print(k, {kk: v.get(kk) for kk in ["value", "ms_per_step", "fock_build_ms", "replicated_ms", "allreduce_ms", "device_GB_rank0", "kernels_ms", "sp2_steps", "sp2_fallbacks"]})
print("real", d.get("real_molecule"))
cb = d.get("cpu_baseline")
if cb:
    print("cpu", cb["value"], cb["cores"], cb["blas"], "dense", cb["dense"]["fock_build_s"], cb["dense"]["steps_s"], "screened", cb["screened"]["fock_build_s"],
          cb["screened"]["steps_s"], "loop", cb["loop_body_s"], "setup", cb["setup_s"])
    print("speedup it", d["speedup_vs_cpu_iteration"], "fock", d["speedup_vs_cpu_fock_build"])
