#!/usr/bin/env python3
"""bench.py — DF-RHF SCF iterations/s + Fock-build TFLOP/s on MI355X.

Contract (driver): python bench.py --gpus N --steps K --warmup W; for N > 1 the
driver launches it with torch.distributed.run, one rank per GPU (RCCL) — and a plain
`python bench.py --gpus N` starts those N ranks itself as a child process before it
touches torch or HIP (launch_ranks).  WORLD_SIZE != --gpus is fatal (exit 2): a line
whose n_gpus differs from what was asked for is never printed.  Rank 0 prints ONE
JSON line.

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
launch on the launch stream inside the timed region; "cpu_baseline" = the reference's
two CPU Fock-build modes (dense BLAS mode and the default screened / blocked mode)
restated in C on the best host BLAS of the box (oracle/cpu_baseline.py; vendor, thread
count and a DGEMM calibration printed), timed on this box's host cores on the same
workload, rank 0, N = 1 only.  "scaling_w50": the (H2O)50 / cc-pVDZ shape of BASELINE
config 4 measured in the same run on the same ranks (the strong-scaling workload of
north_star; `value` stays on C20H42 so that the N = 1 line equals the plain bench).
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
    """SURVEY 8d / BASELINE.md: F_alg = 4 Q N^2 o + 4 Q N^2 + 2 N^2 o  — the DENSE FORMULA: K counted as the full
    2 Q o N^2 although only its lower triangle has to be computed."""
    return 4.0 * Q * N * N * o + 4.0 * Q * N * N + 2.0 * N * N * o


def fock_useful_flops(N, Q, o, P):
    """What a Fock build has to execute at least: W on the kept pairs 2 Q P o (GPUDF.jl:637-667), K symmetric
    Q o N (N+1), V fused 2 Q N o, J on the kept lower pairs 2 Q (P+N)/2."""
    return 2.0 * Q * P * o + Q * o * N * (N + 1.0) + 2.0 * Q * N * o + Q * (P + N)


def csrc_hash():
    """sha256 over the kernel sources: a committed PMC record is only valid for the code it was measured on."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "juliachem.jl_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hpp", ".hip", ".cpp", ".h")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


PMC_RECORD = os.path.join("profiles", "r04_pmc_traffic.json")
STEP_RECORD = os.path.join("profiles", "r04_step_kernels.json")


def checked_record(rel, shape=None):
    """A committed profile record, only if it was measured on THIS kernel source (and shape): (record, None) or (None, why)."""
    f = os.path.join(ROOT, rel)
    if not os.path.exists(f):
        return None, "%s missing" % rel
    try:
        rec = json.load(open(f))
    except Exception as e:
        return None, "%s unreadable: %r" % (rel, e)
    if rec.get("csrc_sha256_16") != csrc_hash():
        return None, "%s was measured on other kernel sources (hash mismatch): re-run tools/collect_round_profiles.sh" % rel
    if shape is not None and list(rec.get("shape", [])) != list(shape):
        return None, "%s holds shape %s" % (rel, rec.get("shape"))
    return rec, None


def pmc_kernel(kernel, shape, world):
    """Per-launch PMC figures of `kernel` from the committed passes (separate rocprofv3 --pmc runs of tools/prof_fock.py,
    tools/pmc_passes.sh): HBM bytes = 2 FETCH_SIZE + WRITE_SIZE (the gfx950 correction of MI355X_MICROARCH.md), MFMA-busy
    fraction = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCD x 1024 SIMD), LDS bank-conflict share."""
    if world != 1:
        return None, "PMC records are single-GPU"
    rec, why = checked_record(PMC_RECORD, shape)
    if rec is None:
        return None, why
    k = rec.get(kernel)
    if not k:
        return None, "%s has no entry %s" % (PMC_RECORD, kernel)
    out = {"kernel_name": k.get("kernel_name"), "hbm_bytes_per_launch": k.get("hbm_bytes_per_launch"), "source": PMC_RECORD}
    if k.get("mfma_busy_cycles") and k.get("grbm_gui_active"):
        out["mfma_busy_frac"] = k["mfma_busy_cycles"] / (k["grbm_gui_active"] / 8.0 * 1024.0)
    if k.get("lds_bank_conflict") is not None and k.get("lds_idx_active"):
        out["lds_bank_conflict_frac"] = k["lds_bank_conflict"] / k["lds_idx_active"]
    return out, None


def pmc_traffic(kernel, shape, world):
    """HBM bytes per launch of `kernel` from the committed PMC passes (separate rocprofv3 --pmc runs,
    tools/pmc_passes.sh; (2 FETCH_SIZE + WRITE_SIZE), the gfx950 correction of MI355X_MICROARCH.md) — only if the
    record was taken on THIS kernel source and THIS shape; otherwise null."""
    k, why = pmc_kernel(kernel, shape, world)
    if k is None:
        return None, why
    return k["hbm_bytes_per_launch"], "%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of tools/prof_fock.py; kernel %s, shape and csrc hash checked)" % (PMC_RECORD, k["kernel_name"])


def cpu_baseline(N, Q, o, budget_s=12.0):
    """both CPU Fock-build modes of the reference (checker code, used here as the reported CPU
    baseline only) on the host cores, plus the SCF loop body around the faster one."""
    from oracle import cpu_baseline as cb, df_fock as orc, scf as oscf
    from juliachem_jl_amd import synthetic
    t0 = time.perf_counter()
    base = cb.CpuBaseline()
    rng = np.random.default_rng(1)

    def fortran_random(shape):
        # uniform numbers straight into a column-major array (normal variates + a symmetrisation pass over 4 GB cost a
        # minute of host time and change nothing for BLAS timings)
        a = np.empty(shape, order="F")
        flat = a.reshape(-1, order="A")
        for i0 in range(0, flat.size, 1 << 26):
            rng.random(out=flat[i0:i0 + (1 << 26)])
        flat -= 0.5
        return a
    B = fortran_random((Q, N, N))
    C, _ = np.linalg.qr(rng.standard_normal((N, N)))
    Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
    X = np.eye(N)
    # the packed tensor of the screened mode: the 47 %-kept band map SURVEY 8d reads off the reference's C20H42 picture
    mask = synthetic.band_mask(N, 0.47, np.random.default_rng(2))
    sd = orc.get_screening_metadata(mask)
    Bp = fortran_random((Q, int(sd.screened_indices_count)))
    gen = time.perf_counter() - t0

    def run(fn, budget):
        ts, parts = [], []
        start = time.perf_counter()
        it = 0
        while it < 2 or (time.perf_counter() - start < budget and it < 8):
            t1 = time.perf_counter()
            F, part = fn()
            if it > 0:                                   # first pass warms the BLAS threads
                ts.append(time.perf_counter() - t1); parts.append(part)
            it += 1
        return F, float(np.mean(ts)), {k: float(np.mean([p[k] for p in parts])) for k in parts[0]}, len(ts)

    Fd, t_dense, parts_dense, n_dense = run(lambda: base.fock_dense(B, C[:, :o], H), budget_s)
    Fs, t_scr, parts_scr, n_scr = run(lambda: base.fock_screened(Bp, sd, C[:, :o], H), budget_s)
    P = int(sd.screened_indices_count)
    # the rest of an SCF iteration on the host (DIIS error, mix, X F X, eigensolve, density, energy), numpy / LAPACK
    F_old, D, S = H.copy(), 2.0 * C[:, :o] @ C[:, :o].T, np.eye(N)
    rest = []
    for it in range(3):
        t1 = time.perf_counter()
        FDS = (Fd @ D) @ S
        e = FDS - FDS.T
        F = 0.5 * Fd + 0.5 * F_old + 1e-12 * e          # same op count as DIIS mix + damping
        _, _, C2, D = oscf.iteration(F, H, X, o)
        if it > 0:
            rest.append(time.perf_counter() - t1)
    t_rest = float(np.mean(rest))
    dense = {"fock_build_s": t_dense, "steps_s": parts_dense, "samples": n_dense,
             "fock_build_tflops_dense_formula": fock_alg_flops(N, Q, o) / t_dense / 1e12,
             "reference": "df_rhf_fock_build_BLAS! (DensityFitting.jl:111-125,185-224), dense (Q,N,N) tensor, all BLAS threads"}
    screened = {"fock_build_s": t_scr, "steps_s": parts_scr, "samples": n_scr, "kept_pair_fraction": P / float(N * N),
                "fock_build_useful_tflops": fock_useful_flops(N, Q, o, P) / t_scr / 1e12,
                "reference": "df_rhf_fock_build_screened! (ScreenedDF.jl:80-132,242-378,548-641), the reference's default CPU mode: "
                             "packed (Q,P) tensor with a 47 %-kept band map, p-parallel single-threaded gemm for W, 10x10 lower-triangle K blocks"}
    best = min(t_dense, t_scr)
    it_s = best + t_rest
    return {"value": 1.0 / it_s, "unit": "SCF iterations/s", "cores": int(base.threads), "kind": "port",
            "sample": "%d + %d full-size Fock builds (N=%d,Q=%d,o=%d; dense and screened CPU modes of the reference restated in C, "
                      "oracle/c/jcdf_cpu_baseline.c) after 1 warm-up each, + 2 host SCF-loop bodies; value = 1 / (faster Fock build + loop body)"
                      % (n_dense, n_scr, N, Q, o),
            "blas": {"vendor": base.blas, "threads": int(base.threads), "dgemm_%d_gflops" % base.calibrate_n: base.dgemm_gflops,
                     "candidates_gflops": base.calibration},
            "dense": dense, "screened": screened, "loop_body_s": t_rest,
            "fock_build_s": best, "iteration_s": it_s, "setup_s": gen,
            "note": "the reference itself (Julia + Libint) cannot run here; this is its algorithm on this box's host cores"}


LAST_COLLECTIVES = {}      # per-step ms of the broadcast of C and the all-reduce of F in the most recent run_scf_steps


def run_scf_steps(scf, fb, steps, warmup, barrier):
    """W untimed + K timed SCF iterations.  No host call but scf.step() inside the timed loop: the library sums the
    HIP-event times of every build's launches itself (jcdf_kernel_stats_total), the collectives are timed by device events."""
    for _ in range(warmup):
        scf.step()
    fb.time_collectives = True
    fb.collective_events = []
    barrier()
    fb.h.kernel_stats_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(steps):
        scf.step()
    barrier()
    elapsed = time.perf_counter() - t0
    recs, nb, fock_sum = fb.h.kernel_stats_total(reset=True)
    nb = max(nb, 1)
    kstats = {r["name"]: dict(seconds=r["seconds"], n=nb, flops=r["flops"], alg_flops=r["alg_flops"], alg_bytes=r["alg_bytes"]) for r in recs}
    parts = fb.collective_ms(split=True)
    fb.time_collectives = False
    LAST_COLLECTIVES.clear()
    LAST_COLLECTIVES.update({k: v / steps for k, v in parts.items()})
    return elapsed, kstats, fock_sum / nb * 1e3, sum(parts.values()) / steps


def max_over_ranks(x, world, dev):
    if world == 1:
        return x
    import torch
    tt = torch.tensor([x], dtype=torch.float64, device=dev if torch.distributed.get_backend() == "nccl" else "cpu")
    torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
    return float(tt.item())


def measure_w50(args, world, rank, local, dev, barrier, kept, density_solver=None, config="w50", steps_cap=10):
    """BASELINE config 4: the (H2O)50 / cc-pVDZ shape (1250 / 4800 / 250), aux index sharded over the ranks, one F
    all-reduce per iteration.  kept = None: unscreened map (60 GB of B in all); kept = 0.13: a scattered 3-D-cluster map
    with the kept fraction of the real cluster (profiles/r02_w50_real_run.txt) — what the reference's adaptive rule runs
    at this size (N >= 800: screened path, DensityFitting.jl:78-90).  B is synthetic, symmetric in (p,q), generated on
    the device in column blocks."""
    import torch
    import juliachem_jl_amd as jc
    from juliachem_jl_amd import synthetic
    from juliachem_jl_amd.engine import DeviceSCF
    N, Q, o = synthetic.CONFIGS[config]
    rng = np.random.default_rng(synthetic.SEED + 50)
    pq = (None, None)
    if kept is None:
        p = np.repeat(np.arange(N, dtype=np.int64), N); q = np.tile(np.arange(N, dtype=np.int64), N)
    else:
        sd = jc.get_screening_metadata(synthetic.cluster_mask(N, kept, rng))
        pq = jc.packed_pq_lists(sd)
        p, q = pq
    P = len(p)
    shells = synthetic.aux_shells(Q, rng)
    Hs = rng.standard_normal((N, N)); H = 0.5 * (Hs + Hs.T)
    t_setup = time.perf_counter()
    fb = make_builder(args, N, Q, o, shells, local, pq)
    fb.set_core_hamiltonian(H)
    for shard, h, sdev in builder_shards(fb, rank):
        # shard `shard` of the synthetic tensor, generated on the device that holds it (same numbers whichever process owns it)
        g = torch.Generator(device=sdev); g.manual_seed(synthetic.SEED + 1000 + shard)
        R = len(fb.ranges[shard])
        g1 = torch.randn((R, N), dtype=torch.float64, device=sdev, generator=g) * 0.05
        g2 = torch.randn((R, N), dtype=torch.float64, device=sdev, generator=g) * 0.05
        pd, qd = torch.as_tensor(p, device=sdev), torch.as_tensor(q, device=sdev)
        for c0 in range(0, P, 8192):
            c1 = min(P, c0 + 8192)
            blk = (g1[:, pd[c0:c1]] * g2[:, qd[c0:c1]] + g1[:, qd[c0:c1]] * g2[:, pd[c0:c1]]).t().contiguous()
            torch.cuda.synchronize(sdev)
            h.set_B_columns_device(c0, c1, blk.data_ptr())
        del g1, g2, blk
    torch.cuda.empty_cache()
    R = len(fb.ranges[rank])
    scf = DeviceSCF(fb, H, np.eye(N), 0.0, density_solver=density_solver or args.density_solver)
    torch.cuda.synchronize(dev)
    t_setup = time.perf_counter() - t_setup
    steps = max(3, min(args.steps, steps_cap))
    elapsed, kstats, fock_ms, coll_ms = run_scf_steps(scf, fb, steps, 8 if density_solver == "sp2" else 2, barrier)
    elapsed = max_over_ranks(elapsed, world, dev)
    nbytes = fb.h.device_bytes()
    rep = scf.solver_report()
    ms = elapsed / steps * 1e3
    Ql = R
    eig_ms = None
    if scf.density_solver == "eigh" and scf.eigh.ok:            # outside the timed steps: the stages of the replicated eigensolve
        scf.eigh.timing = True
        for _ in range(2):
            scf.step()
        torch.cuda.synchronize(dev)
        a_ms, b_ms = scf.eigh.stage_ms()
        scf.eigh.timing = False
        eig_ms = {"tridiagonalisation_ms": a_ms, "tridiagonal_solver_ms": b_ms,
                  "back_transformation": "one GEMM with the Q accumulated in the kernel" if scf.eigh.with_q else
                  "blocked compact-WY from the stored reflectors (jcdf_ormtr_device)"}
    n_shards = len(fb.ranges)
    if args.in_process:
        # member 0's events cover its shard only: the build of the whole group (fetch of C, longest member, reduce, gather) is
        # what the step waits for — measured once more on a drained device
        torch.cuda.synchronize(dev)
        fb.build_ld(scf.Cop, scf.Fbuf[scf.fi ^ 1])
        gtm = fb.group_timings()
        fock_ms = gtm["bcast_ms"] + gtm["build_ms"] + gtm["reduce_ms"] + gtm["gather_ms"]
    out = {"value": steps / elapsed, "unit": "SCF iterations/s", "ms_per_step": ms, "steps": steps, "n_gpus_measured": n_shards,
           "fock_build_ms": fock_ms, "allreduce_ms": coll_ms, "replicated_ms": ms - fock_ms - coll_ms,
           "kernels_ms": {k: v["seconds"] / v["n"] * 1e3 for k, v in kstats.items()},
           "kept_pair_fraction": P / float(N * N), "aux_rows_rank0": Ql, "device_GB_rank0": nbytes / 1e9,
           "fock_build_useful_tflops": fock_useful_flops(N, Q, o, P) / (fock_ms * 1e-3) / 1e12,
           "fock_build_tflops_dense_formula": fock_alg_flops(N, Q, o) / (fock_ms * 1e-3) / 1e12,
           "setup_s": t_setup, "eigensolver": rep, "eigensolver_stages": eig_ms, "density_solver": scf.density_solver,
           "sp2_steps": scf.sp2_steps, "sp2_fallbacks": scf.sp2_fallbacks, "sp2_basis_retries": scf.sp2_basis_retries,
           "sp2_accelerated_steps": scf.sp2_accelerated, "sp2_reference_refreshes": scf.sp2_refreshes}
    if scf.sp2 is not None:
        si = scf.sp2.info.cpu().tolist()                     # the last projection of the run
        out["sp2_last_projection"] = {"squarings": si[0], "finished": si[1], "accelerated": si[6], "delta_to_reference": si[7],
                                      "spectral_bounds": [si[4], si[5]], "squarings_enqueued": scf.sp2.iterations,
                                      "last_density_change": scf.trail[-1][3]}
        if si[6] != 1.0:
            out["sp2_last_projection"]["note"] = ("not accelerated: the distance to the last diagonalised matrix exceeds half its HOMO-LUMO gap "
                                                  "(a synthetic SCF that has not settled — see last_density_change — keeps it there; "
                                                  "profiles/r04_sp2_w50_trend.txt; the real (H2O)50 with this map: 1.7 ms of replicated work "
                                                  "per accelerated iteration, profiles/r04_w50_real_dense_sp2.txt)")
    if args.in_process:
        out["group"] = dict(gtm, transport=fb.g.transport())
    fb.close()
    out["projected_8gpu"] = projected_8gpu(out, 8.0 * N * N)
    return out


def measure_host_boundary(jc, N, Q, o, J2c, H, T_dev, C_occ, device, builds=8):
    """One Fock build through the host entry points a JuliaChem caller binds (julia/JCDFHip.jl): `jcdf_fock_build` on one
    handle and `jcdf_group_fock_build` on a one-device group with either reduce transport — C_occ from host memory, F into
    host memory, so H2D / D2H over PCIe and the call's synchronisation are inside the time.  Same tensor as the timed loop."""
    import torch
    out = {"note": "host C_occ -> host F through the C ABI, PCIe inclusive; ms per build, median of %d" % builds}
    T_dev = T_dev.contiguous()
    torch.cuda.synchronize()

    def run(fn):
        fn()
        ts = []
        for _ in range(builds):
            t0 = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t0) * 1e3)
        return float(np.median(ts))
    h = jc.JCDFHandle(device)
    h.configure(N, Q, 0, Q, o)
    h.set_metric(np.tril(J2c))
    h.push_three_center_device(0, Q, T_dev.data_ptr())
    h.set_core_hamiltonian(H)
    F_ref, t = h.fock_build(C_occ)
    out["jcdf_fock_build_ms"] = run(lambda: h.fock_build(C_occ))
    out["device_fock_ms"] = t.fock_time * 1e3
    out["copy_ms"] = t.copy_time * 1e3
    h.close()
    for transport in ("peer", "rccl"):
        try:
            g = jc.JCDFGroup([device])
            g.set_transport(transport)
            g.configure(N, Q, [0, Q], o)
            g.set_metric(np.tril(J2c))
            g.push_three_center_device(0, Q, T_dev.data_ptr())
            g.set_core_hamiltonian(H)
            F, _, gt = g.fock_build(C_occ)
            out["group_1dev_%s" % transport] = {"ms": run(lambda: g.fock_build(C_occ)), "transport": g.transport(),
                                                "bit_equal_to_handle": bool(np.array_equal(F, F_ref)),
                                                "bcast_ms": gt.bcast_time * 1e3, "reduce_ms": gt.reduce_time * 1e3, "d2h_ms": gt.d2h_time * 1e3}
            g.close()
        except Exception as e:                               # informational object: never fail the bench line for it
            out["group_1dev_%s" % transport] = {"error": repr(e)}
    return out


def distributed_record(world, rank, dev, fb, fock_ms, coll_parts, b_exchange):
    """What makes a multi-GPU line checkable from the line alone (every rank takes part, rank 0 keeps the result): the
    collective backend and its version, the number of ranks an all-reduce of ones actually sees, every rank's aux rows,
    device and Fock-build time (all-gathered), the broadcast of C and the all-reduce of F timed separately, and the
    doubles each rank sent / received in the one-time B exchange."""
    import torch
    rec = {"world_size": world, "collective_backend": None, "rccl_version": None, "ranks_seen": 1,
           "aux_rows": [len(fb.rows)] if world == 1 else None, "fock_build_ms": [fock_ms], "device_index": [dev.index],
           "bcast_ms": coll_parts.get("bcast", 0.0), "allreduce_ms": coll_parts.get("allreduce", 0.0),
           "b_exchange_doubles_sent": [int(b_exchange.get("sent", 0))], "b_exchange_doubles_received": [int(b_exchange.get("received", 0))]}
    try:
        v = torch.cuda.nccl.version()
        rec["rccl_version"] = ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception as e:
        rec["rccl_version"] = "unavailable: %r" % (e,)
    if world > 1:
        dist = torch.distributed
        backend = dist.get_backend()
        cdev = dev if backend == "nccl" else "cpu"
        rec["collective_backend"] = backend
        ones = torch.ones(1, dtype=torch.float64, device=cdev)
        dist.all_reduce(ones)
        rec["ranks_seen"] = int(round(float(ones.item())))
        mine = torch.tensor([float(len(fb.rows)), float(fock_ms), float(dev.index), float(b_exchange.get("sent", 0)),
                             float(b_exchange.get("received", 0)), float(fb.rows.start)], dtype=torch.float64, device=cdev)
        allv = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(allv, mine)
        rows = [[float(x) for x in t.cpu().tolist()] for t in allv]
        rec["aux_rows"] = [int(r[0]) for r in rows]
        rec["fock_build_ms"] = [r[1] for r in rows]
        rec["device_index"] = [int(r[2]) for r in rows]
        rec["b_exchange_doubles_sent"] = [int(r[3]) for r in rows]
        rec["b_exchange_doubles_received"] = [int(r[4]) for r in rows]
        rec["aux_row_start"] = [int(r[5]) for r in rows]
        rec["consistent"] = bool(rec["ranks_seen"] == world and sum(rec["aux_rows"]) == fb.Q_total and dist.get_world_size() == world)
    else:
        rec["consistent"] = True
    return rec


def make_builder(args, N, Q, o, shells, local, pq=(None, None)):
    """one rank per GPU (torch.distributed, default) or ONE process over all --gpus devices (--in-process: jcdf_group_*)"""
    from juliachem_jl_amd.engine import DeviceFockBuilder, GroupFockBuilder
    if not args.in_process:
        return DeviceFockBuilder(N, Q, o, shells, device=local, pq=pq)
    import torch
    ndev = torch.cuda.device_count()
    if os.environ.get("JCDF_BENCH_SHARE_DEVICE") == "1":         # rehearsal on a one-GPU box: the members share the device
        devices = [i % max(1, ndev) for i in range(args.gpus)]
    elif args.gpus > ndev:
        sys.stderr.write("bench.py --in-process: %d devices asked for, the node shows %d (JCDF_BENCH_SHARE_DEVICE=1 shares one GPU)\n"
                         % (args.gpus, ndev))
        raise SystemExit(2)
    else:
        devices = list(range(args.gpus))
    return GroupFockBuilder(N, Q, o, shells, devices, pq=pq, transport=None if args.transport == "auto" else args.transport)


def builder_shards(fb, rank):
    """(shard index, handle, torch device) of every aux shard this process holds"""
    import torch
    if hasattr(fb, "g"):
        return [(i, m, torch.device("cuda", d)) for i, (m, d) in enumerate(zip(fb.g.members, fb.devices))]
    return [(rank, fb.h, fb.device)]


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", default="C20H42")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-w50", action="store_true", help="skip the scaling_w50 object")
    ap.add_argument("--no-gly10", action="store_true", help="skip the gly10_vtz object (BASELINE config 5 shape, 30 %-kept map)")
    ap.add_argument("--no-real", action="store_true", help="skip the real_molecule object (profiling runs)")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host_boundary object (host C in, host F out through the C ABI)")
    ap.add_argument("--in-process", action="store_true",
                    help="ONE process drives all --gpus devices through the C ABI's multi-device group (jcdf_group_*: C fetched "
                         "device-to-device, F reduced on the devices, the SCF loop on device 0 only) instead of one rank per GPU over "
                         "torch.distributed; same workload, same metric — the two transports of an 8-GPU node side by side")
    ap.add_argument("--transport", default="auto", choices=["auto", "rccl", "peer"], help="--in-process: the group's reduce transport")
    ap.add_argument("--density-solver", default="eigh", choices=["eigh", "sp2"],
                    help="eigh: the reference's eigensolve per iteration (default, what `value` is quoted on); sp2: spectral projection")
    return ap.parse_args(argv)


def launch_ranks(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves, as a CHILD process
    (torch.distributed.run, one rank per GPU) — this process has not imported torch or touched HIP, and it never execs.
    Rank 0's JSON line is relayed on stdout, everything else on stderr; the child's return code is ours.  torchrun tears
    the other ranks down when one of them fails."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    lines = 0
    for line in child.stdout:
        if line.startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
            lines += 1
        else:
            sys.stderr.write(line)
    rc = child.wait()
    if rc == 0 and lines != 1:
        sys.stderr.write("bench.py: the %d-rank child printed %d JSON lines (expected 1)\n" % (args.gpus, lines))
        rc = 3
    return rc


XGMI_ALLREDUCE_GBS = 7 * 153.0 / 2.0      # ring all-reduce over 7 xGMI links of ~153 GB/s: ~2 (n-1)/n x bytes over the per-GPU link sum; rough


def projected_8gpu(line, nbytes):
    """PROJECTION, not a measurement: 8-GPU time of an SCF iteration from the parts measured on THIS run's ranks — the
    Fock build shards over the aux index (per-rank time = whole-job Fock-build time / 8), the replicated part does not
    shard, and the N x N all-reduce is priced at the xGMI ring rate.  ratio = 1-GPU-equivalent time / projected time."""
    n = line.get("n_gpus_measured", 1)
    fock_all = line["fock_build_ms"] * n                       # per-rank Fock build x ranks = the whole job's
    t1 = fock_all + line["replicated_ms"]
    ar = 2.0 * 7.0 / 8.0 * nbytes / (XGMI_ALLREDUCE_GBS * 1e9) * 1e3 + 0.03
    t8 = fock_all / 8.0 + line["replicated_ms"] + ar
    return {"ms_per_step": t8, "speedup_over_1gpu": t1 / t8, "fock_build_ms": fock_all / 8.0, "replicated_ms": line["replicated_ms"],
            "allreduce_ms_assumed": ar, "note": "projection from measured parts (Amdahl): (F + R) / (F / 8 + R + all-reduce)"}


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("bench.py: --gpus must be >= 1")
    if args.in_process:
        if int(os.environ.get("WORLD_SIZE", "1")) != 1:
            sys.stderr.write("bench.py --in-process is ONE process over all devices: start it without a launcher\n")
            raise SystemExit(2)
        world = 1                                              # torch sees a single-rank job; the devices are the group's members
    elif "WORLD_SIZE" not in os.environ:
        if args.gpus > 1:
            raise SystemExit(launch_ranks(args, argv))
        world = 1
    else:
        world = int(os.environ["WORLD_SIZE"])
    if world != args.gpus and not args.in_process:
        # a record that says n_gpus = 1 for a run asked to use 8 (or the reverse) must never exist
        sys.stderr.write("bench.py: --gpus %d but WORLD_SIZE=%d: launch with --nproc-per-node %d (or drop the launcher: "
                         "bench.py starts its own ranks)\n" % (args.gpus, world, args.gpus))
        raise SystemExit(2)

    # ONE JSON line on stdout and nothing else: libraries that print to file descriptor 1 (RCCL writes its version banner there when
    # a communicator is created) are sent to stderr for the whole run; the line goes out through the saved descriptor at the end
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import juliachem_jl_amd as jc
    from juliachem_jl_amd import synthetic
    from juliachem_jl_amd.engine import DeviceSCF

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import datetime
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("JCDF_BENCH_BACKEND", "nccl")      # "gloo": host-staged rehearsal on a 1-GPU box
        # a rank that dies must not leave the others waiting in a collective for ever
        limit = datetime.timedelta(seconds=int(os.environ.get("JCDF_BENCH_TIMEOUT_S", "900")))
        ndev = torch.cuda.device_count()
        if backend == "gloo":
            local = local % max(1, ndev)
            torch.cuda.set_device(local)
            dist.init_process_group("gloo", timeout=limit)
        else:
            if local >= ndev:
                sys.stderr.write("bench.py: rank %d needs GPU %d but the node shows %d (JCDF_BENCH_BACKEND=gloo shares one GPU)\n"
                                 % (rank, local, ndev))
                raise SystemExit(2)
            torch.cuda.set_device(local)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local), timeout=limit)
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
    fb = make_builder(args, N, Q, o, shells, local)
    n_shards = len(fb.ranges)
    shard_rows = [len(r) for r in fb.ranges]
    fb.set_metric(J2c)
    fb.set_core_hamiltonian(H)

    def shard_block(i):
        # aux shard i's three-centre block, generated on this process's device: [p][q][a] contiguous == (rows, N*N) column-major
        # (seeded by the shard, so one rank per GPU and one process over all GPUs contract the same tensor)
        g = torch.Generator(device=dev); g.manual_seed(synthetic.SEED + 17 * i)
        A = torch.randn((N, N, shard_rows[i]), dtype=torch.float64, device=dev, generator=g) * 0.1
        return (0.5 * (A + A.transpose(0, 1))).contiguous().reshape(-1)
    if args.in_process:
        b_exchange = {"sent": 0, "received": 0, "recv_buffer": 0}
        for i, r in enumerate(fb.ranges):                      # every shard is in this process: pushed block by block
            fb.push_three_center_device(r.start, r.stop, shard_block(i))
    else:
        T_own = shard_block(rank)
        b_exchange = fb.exchange_three_center(T_own)
        del T_own
    torch.cuda.empty_cache()
    scf = DeviceSCF(fb, H, S, 0.0, density_solver=args.density_solver)
    torch.cuda.synchronize(dev)
    t_setup = time.perf_counter() - t_setup

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize(dev)

    elapsed, kstats, fock_ms, coll_ms = run_scf_steps(scf, fb, args.steps, args.warmup, barrier)
    coll_parts = dict(LAST_COLLECTIVES)
    member0_fock_ms = fock_ms
    group_obj = None
    if args.in_process:
        # member 0's events cover its shard; the step waits for the whole group build: fetch of C, longest member, reduce, gather
        torch.cuda.synchronize(dev)
        fb.build_ld(scf.Cop, scf.Fbuf[scf.fi ^ 1])
        gtm = fb.group_timings()
        fock_ms = gtm["bcast_ms"] + gtm["build_ms"] + gtm["reduce_ms"] + gtm["gather_ms"]
        group_obj = dict(gtm, transport=fb.g.transport(), devices=fb.devices, member0_fock_ms_timed_loop=member0_fock_ms,
                         aux_rows=[len(r) for r in fb.ranges])
    dist_obj = distributed_record(world, rank, dev, fb, fock_ms, coll_parts, b_exchange)
    # outside the timed region: stand-alone duration of the HBM-streaming J pass (in the timed steps it runs beside the
    # MFMA-bound K pass on a side stream and takes longer while it shares the device)
    fb.h.set_overlap(False)
    j_alone = []
    for _ in range(3):
        fb.build(scf.Co_t)
        j_alone.append([ks["seconds"] for ks in fb.h.kernel_stats() if ks["name"] == "k_coulomb_J"][0])
    fb.h.set_overlap(True)
    j_alone_s = float(np.median(j_alone))
    # also outside the timed region (the events would sit on the timed stream): the two stages of the replicated eigensolve,
    # device events around them over 5 more iterations of the same loop
    sytrd_ms = stedc_ms = None
    if args.density_solver == "eigh" and scf.eigh.ok:
        scf.eigh.timing = True
        for _ in range(5):
            scf.step()
        torch.cuda.synchronize(dev)
        sytrd_ms, stedc_ms = scf.eigh.stage_ms()
        scf.eigh.timing = False
    # also outside the timed region: the same SCF with the optional spectral-projection density solver (no eigensolve
    # per iteration) — reported beside, never as `value`
    alt = None
    if args.density_solver == "eigh":
        scf2 = DeviceSCF(fb, H, S, 0.0, density_solver="sp2")
        alt_s, alt_k, alt_fock_ms, alt_coll = run_scf_steps(scf2, fb, args.steps, 6, barrier)
        alt_s = max_over_ranks(alt_s, world, dev)
        alt = {"density_solver": "sp2", "value": args.steps / alt_s, "unit": "SCF iterations/s", "ms_per_step": alt_s / args.steps * 1e3,
               "steps": args.steps, "fock_build_ms": alt_fock_ms, "allreduce_ms": alt_coll,
               "replicated_ms": alt_s / args.steps * 1e3 - alt_fock_ms - alt_coll,
               "kernels_ms": {k: v["seconds"] / v["n"] * 1e3 for k, v in alt_k.items()}, "sp2_steps": scf2.sp2_steps, "sp2_fallbacks": scf2.sp2_fallbacks,
               "energy_minus_eigh": scf2.trail[-1][1] - scf.trail[-1][1],
               "sp2_last_projection": (lambda si: {"squarings": si[0], "finished": si[1], "accelerated": si[6], "delta_to_reference": si[7],
                                                   "spectral_bounds": [si[4], si[5]], "squarings_enqueued": scf2.sp2.iterations})(scf2.sp2.info.cpu().tolist())
               if scf2.sp2 is not None else None,
               "sp2_basis_retries": scf2.sp2_basis_retries, "sp2_accelerated_steps": scf2.sp2_accelerated, "sp2_reference_refreshes": scf2.sp2_refreshes,
               "note": "optional scf flag density_solver=sp2: occupied-space projector by matrix squarings (jcdf_sp2_device) instead "
                       "of the per-iteration eigensolve, basis by Newton-Schulz (jcdf_lowdin_rows_device): every product on the "
                       "library's own cores; same energies; not the default, not `value`"}
    elapsed = max_over_ranks(elapsed, world, dev)
    solver_report = scf.solver_report()
    rows0 = len(fb.ranges[0])
    onehop_max_n = int(jc._lib.load().jcdf_sytrd_max_n(1))
    Co_host = scf.Co_t.t().contiguous().cpu().numpy()           # (N, n_occ): what a host caller hands over
    fb.close()
    del scf, fb
    torch.cuda.empty_cache()
    # the C ABI as the reference's Julia caller uses it: host C_occ in, host F out (PCIe inclusive; never `value`)
    host_boundary = None
    if n_shards == 1 and world == 1 and not args.no_host_boundary:
        host_boundary = measure_host_boundary(jc, N, Q, o, J2c, H, shard_block(0), np.asfortranarray(Co_host), local)
        torch.cuda.empty_cache()

    # the strong-scaling workload of north_star, same ranks, same run (never `value`)
    w50 = None
    if not args.no_w50:
        w50 = {"workload": "(H2O)50 / cc-pVDZ + cc-pVDZ-RIFIT shaped DF-RHF SCF iteration: N=1250 AO, Q=4800 aux, n_occ=250, aux index "
                           "sharded over %d GPU(s), %s per iteration"
                           % (n_shards, "one process (jcdf_group): C fetched device-to-device, F reduced on the devices" if args.in_process
                              else "C broadcast + F all-reduce over RCCL"),
               "screened_13pct": measure_w50(args, world, rank, local, dev, barrier, 0.13),
               "dense_map": measure_w50(args, world, rank, local, dev, barrier, None),
               # the same screened problem with the optional spectral-projection density solver (no eigensolve per
               # iteration): what the replicated part costs strong scaling — informational, like `alt`
               "screened_13pct_sp2": measure_w50(args, world, rank, local, dev, barrier, 0.13, density_solver="sp2"),
               "dense_map_sp2": measure_w50(args, world, rank, local, dev, barrier, None, density_solver="sp2")}

    # BASELINE config 5: the glycine-oligomer / cc-pVTZ shape (N = 1915: above the size whose Q fits the tridiagonalisation
    # kernel — two-kernel tridiagonalisation + compact-WY back-transformation, no vendor kernel), 30 %-kept map; never `value`
    gly10 = None
    if not args.no_gly10 and not args.no_w50:
        gly10 = measure_w50(args, world, rank, local, dev, barrier, 0.30, config="gly10_vtz", steps_cap=4)
        gly10["workload"] = "glycine-oligomer / cc-pVTZ shaped DF-RHF SCF iteration (BASELINE config 5): N=1915 AO, Q=5261 aux, n_occ=155, 30 %-kept pair map"
        rec5, why5 = checked_record(os.path.join("profiles", "r04_step_kernels_gly10.json"), (1915, 5261, 155)) if n_shards == 1 else (None, "single-GPU record")
        gly10["vendor_kernels_per_step"] = rec5.get("vendor_kernels_per_step") if rec5 else None
        gly10["step_kernels_source"] = ("profiles/r04_step_kernels_gly10.json (rocprofv3 --kernel-trace of tools/scf_steps.py gly10_vtz; csrc hash checked)"
                                        if rec5 else why5)

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        P = N * N
        f_alg = fock_alg_flops(N, Q, o)
        f_use = fock_useful_flops(N, Q, o, P)
        w = kstats["k_exchange_W"]
        w_avg = w["seconds"] / w["n"]
        w_alg = w["alg_flops"]                  # algorithmic flops of ONE launch (this rank's aux shard)
        achieved = w_alg / w_avg / 1e12
        traffic, traffic_src = pmc_traffic("k_exchange_W", (N, Q, o), n_shards)
        step_rec, step_why = checked_record(STEP_RECORD, (N, Q, o)) if n_shards == 1 else (None, "single-GPU record")
        k_pmc, k_why = pmc_kernel("k_exchange_K64", (N, Q, o), n_shards)
        w_pmc, _ = pmc_kernel("k_exchange_W", (N, Q, o), n_shards)
        kk = kstats["k_exchange_K"]
        k_avg = kk["seconds"] / kk["n"]
        k_useful = rows0 * o * N * (N + 1.0)                                      # this rank's (member 0's) aux shard
        longest = None
        if sytrd_ms is not None:
            longest = {"kernel": ("k_sytrd_onehop" if N <= onehop_max_n else "k_sytrd_lower") + " (columns 0 .. N-129) + k_sytd2_tail (last 128, one workgroup) + k_q_tail_reflect",
                       "role": "replicated eigensolve, tridiagonalisation + Q (caller side, SCF.jl:1083)",
                       "ms": sytrd_ms, "us_per_column": sytrd_ms * 1e3 / N, "share_of_ms_per_step": sytrd_ms / ms,
                       "bound": "latency: one chip-wide hand-off per column (4.6-7 us), 1.7 us per column inside the one-workgroup tail; not on a flop or byte roofline",
                       "flops": 4.0 / 3.0 * N ** 3 * 2.0, "frac_fp64_peak": 4.0 / 3.0 * N ** 3 * 2.0 / (sytrd_ms * 1e-3) / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                       "stedc_ms": stedc_ms, "measured": "device events around the launch over 5 iterations after the timed loop"}
        out = {
            "metric": "SCF iterations/sec (DF-RHF, C20H42/cc-pVDZ shape); Fock-build TFLOP/s in fock_build_useful_tflops / fock_build_tflops_dense_formula",
            "value": args.steps / elapsed, "unit": "SCF iterations/s", "n_gpus": n_shards, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True,
            "scaling": "strong", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "%s/cc-pVDZ + cc-pVDZ-RIFIT shaped DF-RHF SCF iteration: N=%d AO, Q=%d aux, n_occ=%d, "
                                   "dense pq map, aux index sharded over %d GPU(s), %s"
                                   % (args.config, N, Q, o, n_shards,
                                      "ONE process, jcdf_group: C fetched device-to-device, F reduced on the devices, SCF loop on device 0"
                                      if args.in_process else "one rank per GPU, C broadcast + F all-reduce over RCCL"),
                       "transport": "in-process group" if args.in_process else "torch.distributed"},
            # what an N-GPU record can be checked with: backend, RCCL version, ranks an all-reduce of ones saw, every rank's
            # aux rows / device / Fock-build time, the broadcast of C and the all-reduce of F timed separately, B-exchange traffic
            "distributed": dist_obj,
            "in_process_group": group_obj,
            "host_boundary": host_boundary,
            "density_solver": {"name": scf_name(args), "trail_note": "the timed steps sit on the converged fixed point of the synthetic problem "
                               "(all work executed; DIIS takes its singular branch); real-molecule iterations: real_molecule",
                               "eigensolver": solver_report},
            "alt": alt,
            "fock_build_ms": fock_ms,
            "allreduce_ms": coll_ms,
            "replicated_ms": ms - fock_ms - coll_ms,          # DIIS + damping + X F X + eigensolve + density + energy (per rank, not sharded)
            # launches of ONE timed step by family, from the committed rocprofv3 kernel trace of this bench — only if that
            # record was taken on these kernel sources; otherwise null with the reason (never a constant)
            "vendor_kernels_per_step": step_rec.get("vendor_kernels_per_step") if step_rec else None,
            "launches_per_step": step_rec.get("launches_per_step") if step_rec else None,
            "small_launch_ms_per_step": step_rec.get("small_launch_ms_per_step") if step_rec else None,
            "step_kernels_source": STEP_RECORD + " (rocprofv3 --kernel-trace of bench.py, tools/collect_round_profiles.sh; csrc hash checked)"
                                   if step_rec else step_why,
            # useful = what must be executed (K symmetric, W on the kept pairs); dense_formula = SURVEY 8d's F_alg (K counted twice over)
            "fock_build_useful_tflops": f_use / (fock_ms * 1e-3) / 1e12,
            "fock_build_useful_pct_fp64_mfma_peak": 100.0 * f_use / (fock_ms * 1e-3) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * n_shards),
            "fock_build_tflops_dense_formula": f_alg / (fock_ms * 1e-3) / 1e12,          # whole job (all shards)
            "fock_build_pct_fp64_mfma_peak_dense_formula": 100.0 * f_alg / (fock_ms * 1e-3) / 1e12 / (FP64_MFMA_PEAK_TFLOPS * n_shards),
            "setup_s": t_setup,
            "kernels_ms": {k: v["seconds"] / v["n"] * 1e3 for k, v in kstats.items()},
            "kernels_executed_tflops": {k: v["flops"] / (v["seconds"] / v["n"]) / 1e12 for k, v in kstats.items() if v["flops"] > 0},
            "roofline": {"kernel": "k_exchange_W", "bound": "mfma", "achieved": achieved, "peak": FP64_MFMA_PEAK_TFLOPS,
                         "unit": "TFLOP/s", "frac": achieved / FP64_MFMA_PEAK_TFLOPS, "traffic": traffic,
                         "traffic_source": traffic_src, "alg_bytes_per_launch": w["alg_bytes"],
                         "launch_ms": w_avg * 1e3, "alg_flops_per_launch": w_alg,
                         "executed_flops_per_launch": w["flops"], "executed_tflops": w["flops"] / w_avg / 1e12,
                         "alg_hbm_GBs": w["alg_bytes"] / w_avg / 1e9,
                         "mfma_busy_frac_pmc": w_pmc.get("mfma_busy_frac") if w_pmc else None,
                         # north_star: "MFMA utilisation on the K-build GEMMs": the exchange-K SYRK of the same build
                         "k_build": {"kernel": "k_exchange_K64", "launch_ms": k_avg * 1e3, "useful_flops_per_launch": k_useful,
                                     "useful_tflops": k_useful / k_avg / 1e12, "frac_useful": k_useful / k_avg / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                     "executed_tflops": kk["flops"] / k_avg / 1e12, "frac_executed": kk["flops"] / k_avg / 1e12 / FP64_MFMA_PEAK_TFLOPS,
                                     "mfma_busy_frac_pmc": k_pmc.get("mfma_busy_frac") if k_pmc else None,
                                     "lds_bank_conflict_frac_pmc": k_pmc.get("lds_bank_conflict_frac") if k_pmc else None,
                                     "pmc_source": k_pmc["source"] if k_pmc else k_why,
                                     "note": "in the timed steps K runs beside the HBM-bound J pass (kernels_ms)"},
                         "hbm_stream": {"kernel": "k_coulomb_J", "stand_alone_ms": j_alone_s * 1e3,
                                        "GBs": kstats["k_coulomb_J"]["alg_bytes"] / j_alone_s / 1e9,
                                        "peak_GBs": HBM_PEAK_GBS,
                                        "note": "stand-alone launch after the timed loop; in the timed steps J overlaps K (kernels_ms)"}},
            "longest_kernel": longest,
            "scaling_w50": w50,
            "gly10_vtz": gly10,
        }
        if n_shards == 1 and not args.no_real:
            out["real_molecule"] = real_molecule()
        if n_shards == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, Q, o)
            out["speedup_vs_cpu_iteration"] = out["value"] / out["cpu_baseline"]["value"]
            out["speedup_vs_cpu_fock_build"] = out["cpu_baseline"]["fock_build_s"] / (fock_ms * 1e-3)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if world > 1:
        torch.distributed.destroy_process_group()


def scf_name(args):
    return (os.environ.get("JCDF_DENSITY_SOLVER") or args.density_solver).lower()


def real_molecule():
    """Beside the synthetic fixed point: per-iteration wall time of a real SCF through the same path — n-eicosane C20H42
    in the basis pair the reference's logs hold carbon tables for (6-31G(2df,p) / cc-pVTZ-JKFIT: 956 AO, 3390 aux, 81
    occupied; tools/run_c20h42.py), Schwarz-screened packed layout, core guess, 1e-6.  Informational, never `value`."""
    try:
        from juliachem_jl_amd.synthetic import n_alkane
        from juliachem_jl_amd import rhf
        b = json.load(open(os.path.join(ROOT, "tests", "golden", "s22_10_benzene_methane_631g2dfp_jkfit.json")))
        flags = {"dele": 1e-6, "rmsd": 1e-6, "niter": 50, "df_use_adaptive": False}
        t0 = time.perf_counter()
        res = rhf.run(n_alkane(20), b["charges"], b["basis"], b["aux_basis"], flags)
        wall = time.perf_counter() - t0
        it = res["Iteration Times"]
        ks = {k["name"]: k["seconds"] * 1e3 for k in res["Kernel Stats"]}
        out = {"molecule": "n-C20H42, 6-31G(2df,p) / cc-pVTZ-JKFIT, N=%d" % res["Overlap"].shape[0], "converged": bool(res["Converged?"]),
               "iterations": int(res["Iterations"]), "energy": float(res["Energy"]), "wall_s": wall,
               "ms_per_iteration_median": float(np.median(it[1:]) * 1e3), "ms_per_iteration_first": float(it[0] * 1e3),
               "last_fock_build_kernels_ms": ks, "device_GB": res["Device Bytes"] / 1e9,
               "kept_pair_fraction": float(res["Timings"].non_timing_data.get("screened_indices_count", 0)) / res["Overlap"].shape[0] ** 2}
        # the same run with the reference's block-screened exchange (scf flag df_exchange_screen, ScreenedDF.jl:431-447,459-545;
        # off by default there and here): K blocks without a kept pair are not computed
        res2 = rhf.run(n_alkane(20), b["charges"], b["basis"], b["aux_basis"], dict(flags, df_exchange_screen=True))
        ks2 = {k["name"]: k["seconds"] * 1e3 for k in res2["Kernel Stats"]}
        kf = {k["name"]: k["flops"] for k in res["Kernel Stats"]}
        kf2 = {k["name"]: k["flops"] for k in res2["Kernel Stats"]}
        out["exchange_screen"] = {"df_exchange_screen": True, "n_blocks": int(res2["Timings"].non_timing_data.get("df_exchange_screen_blocks", 0)),
                                  "k_exchange_K_ms": ks2.get("k_exchange_K"), "k_exchange_K_ms_unscreened": ks.get("k_exchange_K"),
                                  "k_blocks_computed_fraction": kf2.get("k_exchange_K", 0.0) / max(kf.get("k_exchange_K", 1.0), 1.0),
                                  "converged": bool(res2["Converged?"]), "iterations": int(res2["Iterations"]),
                                  "energy_minus_unscreened": float(res2["Energy"]) - float(res["Energy"]),
                                  "ms_per_iteration_median": float(np.median(res2["Iteration Times"][1:]) * 1e3)}
        return out
    except Exception as e:                                   # informational object: never fail the bench line for it
        return {"error": repr(e)}


if __name__ == "__main__":
    main()
