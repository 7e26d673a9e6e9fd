#!/usr/bin/env python3
"""bench.py — DF-RHF SCF iterations/s + Fock-build TFLOP/s on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W; for N > 1 it
is launched with torch.distributed.run, one rank per GPU (RCCL).  Rank 0 prints
ONE JSON line.

Workload = BASELINE.json metric config: C20H42 / cc-pVDZ (+ cc-pVDZ-RIFIT),
N = 510 AO, Q = 1950 aux, 81 occupied, synthetic tensors of that
exact shape (no Julia/Libint/basis data in the image), fp64.  A "step" is ONE
full SCF iteration = the loop body of scf_cycles_kernel (SCF.jl:399-573): DF Fock
build on the GPU (W, K, V, J, assemble), all-reduce of F over the aux shards
(N > 1), DIIS, damping, X F X, eigensolve, density, energy — nothing skipped.
B = L^-1 (Q|pq) is resident in HBM when the timed region starts (it is formed
once per SCF, like the reference's iteration-1 setup).

Extra objects on the JSON line (tier contract 4): "roofline" for the dominant
kernel (k_exchange_W, fp64 MFMA) from HIP events recorded around each kernel
launch on the launch stream inside the timed region; "cpu_baseline" = the CPU
oracle (numpy restatement of the reference's dense CPU mode, multithreaded host
BLAS) timed on this box's host cores on the same workload, rank 0, N = 1 only.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FP64_MFMA_PEAK_TFLOPS = 78.6      # MI355X datasheet fp64 matrix (= 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
HBM_PEAK_GBS = 8000.0             # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured achievable)


def fock_alg_flops(N, Q, o):
    """SURVEY 8d / BASELINE.md: F_alg = 4 Q N^2 o + 4 Q N^2 + 2 N^2 o."""
    return 4.0 * Q * N * N * o + 4.0 * Q * N * N + 2.0 * N * N * o


def cpu_baseline(N, Q, o, budget_s=25.0):
    """The oracle's dense CPU Fock build + the SCF loop body around it, timed on
    the host cores (checker code used as the reported CPU baseline only)."""
    from oracle import df_fock as orc, scf as oscf
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        threads = len(os.sched_getaffinity(0))
    rng = np.random.default_rng(1)
    t0 = time.perf_counter()
    B = rng.standard_normal((Q, N, N))
    B += B.transpose(0, 2, 1).copy()
    B *= 0.05
    C, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
    X = np.eye(N)
    gen = time.perf_counter() - t0
    times, fock_times = [], []
    F_old = H.copy()
    D = 2.0 * C[:, :o] @ C[:, :o].T
    S = np.eye(N)
    start = time.perf_counter()
    it = 0
    while it < 2 or (time.perf_counter() - start < budget_s and it < 8):
        t1 = time.perf_counter()
        F = H + orc.df_rhf_fock_build_BLAS(B, C[:, :o])
        t2 = time.perf_counter()
        FDS = (F @ D) @ S
        e = FDS - FDS.T
        F = 0.5 * F + 0.5 * F_old + 1e-12 * e          # same op count as DIIS mix + damping
        F_old = F
        _, _, C, D = oscf.iteration(F, H, X, o)
        t3 = time.perf_counter()
        if it > 0:                                       # first pass warms the BLAS threads
            times.append(t3 - t1); fock_times.append(t2 - t1)
        it += 1
    it_s = float(np.mean(times))
    return {"value": 1.0 / it_s, "unit": "SCF iterations/s", "cores": int(threads), "kind": "port",
            "sample": "%d full-size SCF iterations (N=%d,Q=%d,o=%d) after 1 warm-up; dense CPU mode "
                      "(DensityFitting.jl:185-224 restated in numpy, host BLAS threads=%d)" % (len(times), N, Q, o, threads),
            "fock_build_s": float(np.mean(fock_times)), "iteration_s": it_s,
            "fock_build_tflops": fock_alg_flops(N, Q, o) / float(np.mean(fock_times)) / 1e12,
            "setup_s": gen}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C20H42")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--density-solver", default="eigh", choices=["eigh", "sp2"],
                    help="eigh: the reference's eigensolve per iteration (default, what `value` is quoted on); sp2: spectral projection")
    args = ap.parse_args()

    import torch
    import juliachem_jl_amd as jc
    from juliachem_jl_amd import synthetic
    from juliachem_jl_amd.engine import DeviceFockBuilder, DeviceSCF

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JCDF_BENCH_BACKEND", "nccl")      # "gloo": host-staged rehearsal on a 1-GPU box
        if backend == "gloo":
            local = local % max(1, torch.cuda.device_count())
            torch.cuda.set_device(local)
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product path has no CPU fallback")

    N, Q, o = synthetic.CONFIGS[args.config]
    dev = torch.device("cuda", local)
    rng = np.random.default_rng(synthetic.SEED)
    M = rng.standard_normal((Q, Q))
    J2c = M @ M.T + Q * np.eye(Q)
    C0, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
    S = np.eye(N)
    shells = synthetic.aux_shells(Q, rng)

    t_setup = time.perf_counter()
    fb = DeviceFockBuilder(N, Q, o, shells, device=local)
    fb.set_metric(J2c)
    fb.set_core_hamiltonian(H)
    # this rank's three-centre block, generated on the device: [p][q][a] contiguous == (rows, N*N) column-major
    g = torch.Generator(device=dev); g.manual_seed(synthetic.SEED + 17 * rank)
    R = len(fb.rows)
    A = torch.randn((N, N, R), dtype=torch.float64, device=dev, generator=g) * 0.1
    T_own = (0.5 * (A + A.transpose(0, 1))).contiguous().reshape(-1)
    del A
    fb.exchange_three_center(T_own)
    del T_own
    torch.cuda.empty_cache()
    scf = DeviceSCF(fb, H, S, 0.0, density_solver=args.density_solver)
    torch.cuda.synchronize(dev)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        scf.step()
    kstats = {}
    fock_s = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        scf.step()
        # HIP-event timings of the launches of this step (events already complete: step() syncs on E)
        for ks in fb.h.kernel_stats():
            d = kstats.setdefault(ks["name"], dict(seconds=0.0, n=0, flops=ks["flops"], alg_flops=ks["alg_flops"],
                                                   alg_bytes=ks["alg_bytes"]))
            d["seconds"] += ks["seconds"]; d["n"] += 1
        fock_s.append(fb.h.synchronize().fock_time)
    barrier()
    elapsed = time.perf_counter() - t0
    # outside the timed region: stand-alone duration of the HBM-streaming J pass (in the timed steps it runs beside the
    # MFMA-bound K pass on a side stream and takes longer while it shares the device)
    fb.h.set_overlap(False)
    j_alone = []
    for _ in range(3):
        fb.build(scf.Co_t)
        j_alone.append([ks["seconds"] for ks in fb.h.kernel_stats() if ks["name"] == "k_coulomb_J"][0])
    fb.h.set_overlap(True)
    j_alone_s = float(np.median(j_alone))
    # also outside the timed region: the same SCF with the optional spectral-projection density solver (no eigensolve
    # per iteration) — reported beside, never as `value`
    alt = None
    if args.density_solver == "eigh":
        scf2 = DeviceSCF(fb, H, S, 0.0, density_solver="sp2")
        for _ in range(6):
            scf2.step()
        barrier()
        ta = time.perf_counter()
        alt_fock, alt_k = [], {}
        for _ in range(args.steps):
            scf2.step()
            for ks in fb.h.kernel_stats():
                alt_k.setdefault(ks["name"], []).append(ks["seconds"])
            alt_fock.append(fb.h.synchronize().fock_time)
        barrier()
        alt_s = time.perf_counter() - ta
        if world > 1:
            tt = torch.tensor([alt_s], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
            torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
            alt_s = float(tt.item())
        alt = {"density_solver": "sp2", "value": args.steps / alt_s, "unit": "SCF iterations/s", "ms_per_step": alt_s / args.steps * 1e3,
               "steps": args.steps, "fock_build_ms": float(np.mean(alt_fock)) * 1e3,
               "kernels_ms": {k: float(np.mean(v)) * 1e3 for k, v in alt_k.items()}, "sp2_steps": scf2.sp2_steps, "sp2_fallbacks": scf2.sp2_fallbacks,
               "energy_minus_eigh": scf2.trail[-1][1] - scf.trail[-1][1],
               "note": "optional scf flag density_solver=sp2: occupied-space projector by matrix squarings (jcdf_sp2_device) instead "
                       "of the per-iteration eigensolve; same energies; not the default, not `value`"}
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        f_alg = fock_alg_flops(N, Q, o)
        fock_ms = float(np.mean(fock_s)) * 1e3
        w = kstats["k_exchange_W"]
        w_avg = w["seconds"] / w["n"]
        w_alg = w["alg_flops"]                  # algorithmic flops of ONE launch (this rank's aux shard)
        achieved = w_alg / w_avg / 1e12
        # HBM bytes per launch from the PMC passes (separate rocprofv3 --pmc runs, tools/collect_round_profiles.sh;
        # (2*FETCH_SIZE + WRITE_SIZE) KB, the gfx950 correction of MI355X_MICROARCH.md): valid for the 1-GPU workload
        traffic, traffic_src = None, None
        tf = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
        if world == 1 and args.config == "C20H42" and os.path.exists(tf):
            try:
                traffic = json.load(open(tf))["k_exchange_W"]["hbm_bytes_per_launch"]
                traffic_src = "profiles/r01_pmc_traffic.json (rocprofv3 --pmc FETCH_SIZE/WRITE_SIZE passes of tools/prof_fock.py, same kernel and shape)"
            except Exception:
                traffic = None
        out = {
            "metric": "SCF iterations/sec (DF-RHF, C20H42/cc-pVDZ shape); Fock-build TFLOP/s in fock_build_tflops",
            "value": args.steps / elapsed, "unit": "SCF iterations/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s/cc-pVDZ + cc-pVDZ-RIFIT shaped DF-RHF SCF iteration: N=%d AO, Q=%d aux, n_occ=%d, "
                                   "dense pq map, aux index sharded over %d GPU(s), F all-reduce over RCCL"
                                   % (args.config, N, Q, o, world)},
            "density_solver": {"name": scf.density_solver, "sp2_steps": scf.sp2_steps, "sp2_fallbacks": scf.sp2_fallbacks,
                               "sp2_fallback_reasons": scf.sp2_reasons, "trail_tail": [list(t) for t in scf.trail[-3:]]},
            "alt": alt,
            "fock_build_ms": fock_ms,
            "fock_build_tflops": f_alg / (fock_ms * 1e-3) / 1e12,          # whole job (all shards)
            "fock_build_pct_fp64_mfma_peak": 100.0 * f_alg / (fock_ms * 1e-3) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * world),
            "setup_s": t_setup,
            "kernels_ms": {k: v["seconds"] / v["n"] * 1e3 for k, v in kstats.items()},
            "roofline": {"kernel": "k_exchange_W", "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src, "alg_bytes_per_launch": w["alg_bytes"],
                         "launch_ms": w_avg * 1e3, "alg_flops_per_launch": w_alg,
                         "executed_tflops": w["flops"] / w_avg / 1e12,
                         "alg_hbm_GBs": w["alg_bytes"] / w_avg / 1e9,
                         "hbm_stream": {"kernel": "k_coulomb_J", "stand_alone_ms": j_alone_s * 1e3,
                                        "GBs": kstats["k_coulomb_J"]["alg_bytes"] / j_alone_s / 1e9,
                                        "peak_GBs": HBM_PEAK_GBS,
                                        "note": "stand-alone launch after the timed loop; in the timed steps J overlaps K (kernels_ms)"}},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, Q, o)
            out["speedup_vs_cpu_iteration"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_fock_build"] = out["cpu_baseline"]["fock_build_s"] / (fock_ms * 1e-3)
        print(json.dumps(out))
    fb.close()
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
