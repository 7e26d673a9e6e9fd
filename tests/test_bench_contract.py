"""bench.py prints ONE JSON line with the fields the driver and the judge read (metric contract of the task)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_bench_prints_the_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "3", "--warmup", "1", "--no-cpu-baseline"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    # ... and NOTHING else on stdout (RCCL prints a version banner to descriptor 1 when a communicator is created: it must end on stderr)
    assert [l for l in r.stdout.splitlines() if l.strip()] == lines, r.stdout[:600]
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1 and d["dtype"] == "f64" and d["vs_baseline"] is None
    assert d["higher_is_better"] is True and d["data"] == "synthetic" and "workload" in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    rf = d["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0.3 < rf["frac"] < 1.0 and d["value"] > 50.0
    # traffic is either a PMC record of exactly this kernel source and shape, or null with the reason
    assert "traffic_source" in rf and (rf["traffic"] is None or rf["traffic"] > 0.5 * rf["alg_bytes_per_launch"])
    # the Fock-build rate is quoted twice, labelled: useful flops (K symmetric, W on the kept pairs) and SURVEY 8d's dense formula
    for k in ("fock_build_useful_tflops", "fock_build_useful_pct_fp64_mfma_peak", "fock_build_tflops_dense_formula",
              "fock_build_pct_fp64_mfma_peak_dense_formula", "replicated_ms", "allreduce_ms", "fock_build_ms", "vendor_kernels_per_step"):
        assert k in d, k
    # launches per step by family: a hash-checked rocprofv3 record of this very source, or null with the reason — never a constant
    assert "step_kernels_source" in d and (d["vendor_kernels_per_step"] is None or d["vendor_kernels_per_step"] == 0)
    assert (d["vendor_kernels_per_step"] is None) == ("mismatch" in d["step_kernels_source"] or "missing" in d["step_kernels_source"])
    # the longest kernel of the step is named with its share (it is not the roofline kernel) and the K build has its own entry
    lk = d["longest_kernel"]
    assert lk["kernel"].startswith("k_sytrd") and 0.2 < lk["share_of_ms_per_step"] < 0.7 and 1.0 < lk["us_per_column"] < 20.0
    kb = rf["k_build"]
    assert kb["kernel"] == "k_exchange_K64" and 0.3 < kb["frac_useful"] < kb["frac_executed"] < 1.0
    assert "mfma_busy_frac_pmc" in kb and "pmc_source" in kb
    assert "fock_build_tflops" not in d and d["fock_build_useful_tflops"] < d["fock_build_tflops_dense_formula"]
    assert d["allreduce_ms"] == 0.0 and abs(d["replicated_ms"] + d["fock_build_ms"] - d["ms_per_step"]) < 1e-9
    # the strong-scaling workload of north_star ((H2O)50 shape) measured in the same run, never `value`
    w50 = d["scaling_w50"]
    assert w50["screened_13pct_sp2"]["density_solver"] == "sp2" and w50["screened_13pct_sp2"]["replicated_ms"] < w50["screened_13pct"]["replicated_ms"]
    for kind in ("screened_13pct", "dense_map"):
        for k in ("value", "ms_per_step", "fock_build_ms", "allreduce_ms", "replicated_ms", "kernels_ms", "device_GB_rank0"):
            assert k in w50[kind], (kind, k)
    assert 0.11 < w50["screened_13pct"]["kept_pair_fraction"] < 0.16 and w50["dense_map"]["kept_pair_fraction"] == 1.0
    assert w50["screened_13pct"]["device_GB_rank0"] < 0.4 * w50["dense_map"]["device_GB_rank0"]
    assert w50["screened_13pct"]["kernels_ms"]["k_exchange_W"] < 0.3 * w50["dense_map"]["kernels_ms"]["k_exchange_W"]
    for kind in ("screened_13pct", "dense_map", "screened_13pct_sp2", "dense_map_sp2"):
        pj = w50[kind]["projected_8gpu"]
        assert "projection" in pj["note"] and 1.0 < pj["speedup_over_1gpu"] < 8.0
        assert abs(pj["ms_per_step"] - (pj["fock_build_ms"] + pj["replicated_ms"] + pj["allreduce_ms_assumed"])) < 1e-9
    # a real molecule through the same path beside the synthetic fixed point
    assert d["real_molecule"].get("converged") is True, d["real_molecule"]
    xs = d["real_molecule"]["exchange_screen"]
    assert xs["converged"] and xs["k_blocks_computed_fraction"] < 0.9 and xs["k_exchange_K_ms"] < 1.05 * xs["k_exchange_K_ms_unscreened"]
    assert abs(xs["energy_minus_unscreened"]) < 1e-5                  # the screened K blocks hold no pair above the Schwarz threshold
    # the optional spectral-projection density solver is reported beside, on the same problem, with the same energy
    assert d["density_solver"]["name"] == "eigh"
    assert d["alt"]["density_solver"] == "sp2" and d["alt"]["value"] > 50.0 and abs(d["alt"]["energy_minus_eigh"]) < 1e-6
    # the self-verification record of a (here: one-rank) run, and the host boundary of the C ABI (PCIe inclusive)
    ds = d["distributed"]
    assert ds["world_size"] == 1 and ds["ranks_seen"] == 1 and ds["aux_rows"] == [1950] and ds["consistent"] is True
    assert ds["allreduce_ms"] == 0.0 and ds["bcast_ms"] == 0.0 and ds["rccl_version"]
    assert d["in_process_group"] is None and d["config"]["transport"] == "torch.distributed"
    hb = d["host_boundary"]
    assert hb["jcdf_fock_build_ms"] > 0.9 * hb["device_fock_ms"] > 1.0      # (median of 8 host-timed calls vs the device time of one)
    for tr in ("peer", "rccl"):
        assert hb["group_1dev_" + tr]["bit_equal_to_handle"] is True, hb
        assert hb["group_1dev_" + tr]["transport"].startswith(tr)


@pytest.mark.gpu
def test_bench_gpus_2_starts_its_own_ranks():
    """A plain `python bench.py --gpus 2` (no launcher around it) is a 2-rank job: bench.py starts torch.distributed.run as a
    child before touching torch / HIP.  On the one-GPU box both ranks share the card and the collectives are host-staged
    (JCDF_BENCH_BACKEND=gloo): timings mean nothing here, the record's shape does."""
    env = dict(os.environ, JCDF_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-real"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["allreduce_ms"] > 0.0 and d["scaling"] == "strong"
    w50 = d["scaling_w50"]
    assert w50 is not None and w50["screened_13pct"]["allreduce_ms"] > 0.0
    assert w50["screened_13pct"]["aux_rows_rank0"] < 4800          # rank 0 holds a shard of the aux index, not all of it
    assert "cpu_baseline" not in d and "real_molecule" not in d
    # what makes an N-GPU record checkable from the line alone
    ds = d["distributed"]
    assert ds["world_size"] == 2 and ds["collective_backend"] == "gloo" and ds["ranks_seen"] == 2 and ds["consistent"] is True
    assert len(ds["aux_rows"]) == 2 and sum(ds["aux_rows"]) == 1950 and ds["aux_row_start"] == [0, ds["aux_rows"][0]]
    assert len(ds["fock_build_ms"]) == 2 and all(x > 0 for x in ds["fock_build_ms"]) and len(ds["device_index"]) == 2
    assert ds["allreduce_ms"] > 0.0 and ds["bcast_ms"] > 0.0 and abs(ds["allreduce_ms"] + ds["bcast_ms"] - d["allreduce_ms"]) < 1e-9
    # the one-time B exchange is the lower triangle: rank 0 sends its block to rank 1 and receives nothing
    P = 510 * 510
    assert ds["b_exchange_doubles_sent"] == [ds["aux_rows"][0] * P, 0] and ds["b_exchange_doubles_received"] == [0, ds["aux_rows"][0] * P]
    assert ds["rccl_version"] and d["host_boundary"] is None and d["in_process_group"] is None


@pytest.mark.gpu
def test_bench_in_process_group_rehearsal():
    """`bench.py --gpus 2 --in-process`: ONE process, the devices behind the C ABI's multi-device group (C fetched device-to-device,
    F reduced on the devices, the SCF loop on device 0).  On the one-GPU box the two members share the card
    (JCDF_BENCH_SHARE_DEVICE=1, "peer" transport): timings mean nothing here, the record's shape and the energy do."""
    env = dict(os.environ, JCDF_BENCH_SHARE_DEVICE="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--in-process", "--steps", "3", "--warmup", "1",
                        "--no-cpu-baseline", "--no-real"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["transport"] == "in-process group" and d["allreduce_ms"] == 0.0
    g = d["in_process_group"]
    assert g["transport"].startswith("peer") and g["devices"] == [0, 0] and sum(g["aux_rows"]) == 1950
    assert len(g["member_fock_ms"]) == 2 and all(x > 0 for x in g["member_fock_ms"]) and g["reduce_ms"] > 0 and g["gather_ms"] > 0
    assert abs(d["fock_build_ms"] - (g["bcast_ms"] + g["build_ms"] + g["reduce_ms"] + g["gather_ms"])) < 1e-9
    assert d["distributed"]["world_size"] == 1 and d["host_boundary"] is None
    w50 = d["scaling_w50"]["screened_13pct"]
    assert w50["n_gpus_measured"] == 2 and w50["group"]["transport"].startswith("peer") and w50["aux_rows_rank0"] < 4800
    # the same SCF as the one-rank-per-GPU path: the sp2 run reproduces the eigensolver's energy through the group as well
    assert abs(d["alt"]["energy_minus_eigh"]) < 1e-6


def test_bench_refuses_a_world_that_is_not_gpus():
    """--gpus 2 inside a 1-rank world (or the reverse) exits non-zero before anything is measured (no GPU needed)."""
    for gpus, world in (("2", "1"), ("1", "2")):
        env = dict(os.environ, WORLD_SIZE=world, RANK="0", LOCAL_RANK="0")
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", gpus, "--steps", "1", "--warmup", "0"],
                           capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
        assert r.returncode == 2, (r.returncode, r.stderr[-500:])
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
        assert "WORLD_SIZE" in r.stderr


def test_bench_child_failure_is_relayed():
    """launch_ranks hands the child's return code on and prints no JSON line when the ranks fail (here: no GPU, so every
    rank exits non-zero; on a GPU box the test_bench_gpus_2 case covers the success side)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("failure side is exercised on the GPU-less box")
    env = dict(os.environ, JCDF_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline",
                        "--no-real", "--no-w50"], capture_output=True, text=True, timeout=300, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_bench_cpu_baseline_object_shape():
    """The cpu_baseline leg (the oracle on the host cores) on a tiny shape: keys of the contract."""
    sys.path.insert(0, ROOT)
    import bench
    cb = bench.cpu_baseline(24, 40, 4, budget_s=0.5)
    for k in ("value", "unit", "cores", "kind", "sample", "blas", "dense", "screened"):
        assert k in cb, k
    assert cb["kind"] == "port" and cb["value"] > 0 and cb["cores"] >= 1
    assert cb["blas"]["vendor"] and cb["blas"]["threads"] == cb["cores"] and any(k.startswith("dgemm_") for k in cb["blas"])
    assert cb["dense"]["fock_build_s"] > 0 and cb["screened"]["fock_build_s"] > 0 and 0.3 < cb["screened"]["kept_pair_fraction"] < 0.6


def test_cpu_baseline_port_matches_the_numpy_oracle():
    """oracle/c/jcdf_cpu_baseline.c (the reference's two CPU modes on the host BLAS) against oracle/df_fock.py."""
    sys.path.insert(0, ROOT)
    import numpy as np
    from oracle import cpu_baseline as cbm, df_fock as orc
    from juliachem_jl_amd import synthetic
    base = cbm.CpuBaseline(calibrate_n=256)
    for (N, Q, o, kept) in [(60, 40, 7, None), (130, 50, 9, 0.4), (257, 48, 9, 0.3)]:
        s = synthetic.make(N, Q, o, seed=3, kept_fraction=kept)
        B = orc.calculate_B(s.J2c, s.T)
        Co = s.C[:, :o]
        ref = s.H + orc.df_rhf_fock_build_BLAS(B, Co)
        F, _ = base.fock_dense(np.asfortranarray(B), Co, s.H)
        assert np.abs(F - ref).max() < 1e-12 * np.abs(ref).max()
        sd = orc.get_screening_metadata(s.mask if s.mask is not None else np.ones((N, N), bool))
        Bp = orc.pack_three_center(B, sd)
        ref2 = s.H + orc.df_rhf_fock_build_screened(Bp, Co, sd)
        F2, _ = base.fock_screened(Bp, sd, Co, s.H)
        assert np.abs(F2 - ref2).max() < 1e-12 * np.abs(ref2).max()
