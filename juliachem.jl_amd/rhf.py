"""`JCRHF.Energy.run` for the density-fitted GPU path, end to end inside this package (reference:
src/rhf/energy/Energy.jl:35-85 -> rhf_energy / rhf_kernel, SCF.jl:60-260): host integrals (include/jcint.h),
Schwarz screening + packed layout (SchwarzScreening.jl:9-83), B formation and the Fock build on the device(s)
(include/jcdf.h), SCF loop on the device (engine.py).  One process per GPU; under torch.distributed every rank
computes the three-centre integrals of its own auxiliary shard only (GPUDF.jl:51-57).

The basis is passed as data (symbol -> shells), see integrals.py; everything else follows the reference's
`keywords.scf` flags (SCFOptions.jl) and returns the reference's result dictionary (SCF.jl:251-258)."""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .df import (JCTC, create_jctiming, create_scf_options, exchange_screen_blocks, get_screening_metadata, packed_pq_lists)
from .integrals import HostIntegralEngine


def spatial_order(atoms: Sequence[Dict], cell: float = 4.0) -> List[int]:
    """Atom order along a Z-order (Morton) curve through cells of `cell` bohr.  The device tensor is the reference's
    packed (Q_d, P) matrix, so memory and work follow the kept pairs in ANY order; with neighbours adjacent the kept q
    of a p come in a few long runs, i.e. the W kernel's row gather reads long contiguous stretches of B and of C, and
    the aux shards of neighbouring ranks' shells stay spatially compact.  Energies do not depend on the order;
    results are returned in the caller's order."""
    R = np.asarray([a["center"] for a in atoms], dtype=np.float64)
    q = np.floor((R - R.min(axis=0)) / cell).astype(np.int64)

    def morton(v):
        code = 0
        for bit in range(16):
            for d in range(3):
                code |= ((int(v[d]) >> bit) & 1) << (3 * bit + d)
        return code
    keys = [morton(v) for v in q]
    return sorted(range(len(atoms)), key=lambda i: (keys[i], i))


def run(atoms: Sequence[Dict], charges: Dict[str, float], basis: Dict[str, List[Dict]], aux_basis: Dict[str, List[Dict]],
        scf_flags: Optional[Dict[str, Any]] = None, molecular_charge: int = 0, output: int = 0,
        device: Optional[int] = None) -> Dict[str, Any]:
    import time
    import torch
    from .engine import DeviceFockBuilder, DeviceSCF

    flags = dict(scf_flags or {})
    flags.setdefault("scf_type", "df")
    flags.setdefault("contraction_mode", "GPU")
    opts = create_scf_options(flags)
    if not opts.density_fitting or opts.contraction_mode not in ("GPU", "HIP", "denseGPU"):
        raise ValueError("rhf.run: only scf_type 'df' with a GPU contraction mode is implemented (no CPU path in this package)")
    if opts.guess != "hcore":
        raise ValueError("rhf.run: guess 'hcore' only (SAD / conventional guesses stay with the reference host code)")
    jc_timing = create_jctiming()
    t_all = time.perf_counter()
    # spatial atom order for the screened layout (see spatial_order); `ao_back` maps the internal AO order to the caller's
    user_atoms = list(atoms)
    order = spatial_order(user_atoms) if flags.get("reorder_atoms", True) else list(range(len(user_atoms)))
    atoms = [user_atoms[i] for i in order]
    nb_atom = [sum((sh["l"] + 1) * (sh["l"] + 2) // 2 for sh in basis[a["symbol"]]) for a in user_atoms]
    start = np.concatenate([[0], np.cumsum(nb_atom)])
    ao_internal = np.concatenate([np.arange(start[i], start[i + 1]) for i in order]).astype(np.int64)   # internal k -> user AO
    ao_back = np.argsort(ao_internal)                                                                 # user AO -> internal k
    eng = HostIntegralEngine(atoms, basis, aux_basis, charges)
    N, Q = eng.prim.nbf, eng.aux.nbf
    nels = int(round(float(np.sum(eng.Z)))) - int(molecular_charge)
    if nels % 2:
        raise ValueError("rhf.run: closed-shell RHF needs an even number of electrons")
    n_occ = nels // 2
    S, T, V = eng.one_electron()
    H = T + V
    E_nuc = eng.nuclear_repulsion()
    t0 = time.perf_counter()
    J2c = eng.calculate_two_center_intgrals()                              # GPUDF.jl:43
    jc_timing.timings[JCTC.two_eri_time] = time.perf_counter() - t0
    # dense below 800 functions on a single rank, packed/screened otherwise (DensityFitting.jl:78-90)
    world = torch.distributed.get_world_size() if torch.distributed.is_available() and torch.distributed.is_initialized() else 1
    dense = opts.contraction_mode == "denseGPU" or opts.df_force_dense or (opts.df_use_adaptive and N < 800 and world == 1)
    sd, pq = None, (None, None)
    if not dense:
        t0 = time.perf_counter()
        mask = eng.schwarz_mask(opts.df_screening_sigma, float(np.max(np.diag(J2c))))     # ScreenedDF.jl:16-77
        sd = get_screening_metadata(mask)
        pq = packed_pq_lists(sd)
        jc_timing.timings[JCTC.screening_time] = time.perf_counter() - t0
        jc_timing.non_timing_data[JCTC.screened_indices_count] = str(int(mask.sum()))
    # df_exchange_screen (ScreenedDF.jl:431-447, 459-545): meaningful on the packed map only (a dense map keeps every block)
    xs = exchange_screen_blocks(opts) if not dense else 0
    jc_timing.non_timing_data["df_exchange_screen_blocks"] = str(xs)
    if opts.num_devices > 1 and world == 1:
        # scf flag num_devices of the reference (one rank, several GPUs: GPUDF.jl:188-277): ONE process, the devices behind the
        # C ABI's multi-device group — the SCF loop on the first device, every Fock build sharded over all of them and reduced
        # on the devices (engine.GroupFockBuilder)
        from .df import _physical_device
        from .engine import GroupFockBuilder
        first = 0 if device is None else int(device)
        fb = GroupFockBuilder(N, Q, n_occ, eng.aux.shell_nbas, [_physical_device(first + d) for d in range(opts.num_devices)], pq=pq,
                              exchange_screen_blocks=xs, transport=flags.get("group_transport"))
        jc_timing.non_timing_data["GPU_reduce_transport"] = fb.g.transport()
    else:
        fb = DeviceFockBuilder(N, Q, n_occ, eng.aux.shell_nbas, device=device, pq=pq, exchange_screen_blocks=xs)
    jc_timing.non_timing_data[JCTC.GPU_num_devices] = str(opts.num_devices if world == 1 else 1)
    fb.set_metric(J2c)
    fb.set_core_hamiltonian(H)
    t_eri = 0.0
    if fb.world == 1:
        # one rank: stream the auxiliary shells through in blocks of <= `block` functions, so the host never holds
        # more than one (block x P) slab of the three-centre tensor (ThreeCenterIntegralsScreened.jl:8-85 fills the
        # whole (A, P) array; (H2O)50: 9 GB packed, 60 GB dense)
        pos = eng.aux.shell_pos
        block = int(flags.get("three_center_block", 512))
        s0 = 0
        while s0 < eng.aux.nshell:
            s1 = s0 + 1
            while s1 < eng.aux.nshell and pos[s1 + 1] - pos[s0] <= block:
                s1 += 1
            t0 = time.perf_counter()
            Tb = eng.calculate_three_center_integrals(range(int(pos[s0]), int(pos[s1])), sd)
            t_eri += time.perf_counter() - t0
            fb.push_three_center_device(int(pos[s0]), int(pos[s1]), torch.as_tensor(np.ravel(Tb, order="K"), device=fb.device))
            torch.cuda.synchronize(fb.device)
            s0 = s1
    else:
        t0 = time.perf_counter()
        T_own = eng.calculate_three_center_integrals(fb.rows, sd)          # (rows, P) column-major, this rank's shard
        t_eri = time.perf_counter() - t0
        fb.exchange_three_center(torch.as_tensor(np.ravel(T_own, order="K"), device=fb.device))
        del T_own
    jc_timing.timings[JCTC.three_eri_time] = t_eri
    scf = DeviceSCF(fb, H, S, E_nuc, density_solver=flags.get("density_solver"))
    converged = False
    E = 0.0
    it = 0
    it_times = []
    for it in range(1, opts.df_max_iterations + 1):                         # SCF.jl:399-573, 596-604
        t_it = time.perf_counter()
        E, dE, drms = scf.step()
        it_times.append(time.perf_counter() - t_it)                         # step() ends with its one host sync
        if output >= 2 and fb.rank == 0:
            print("%d      %.10f      %.10f      %.10f" % (it, E, dE, drms))
        if abs(dE) <= opts.df_energy_convergence and drms <= opts.df_density_convergence:
            converged = True
            break
    scf.canonical_orbitals()                                               # no-op with the default eigensolver
    eps = scf.eps.cpu().numpy()
    C = scf.C.cpu().numpy()
    Cocc = C[:, :n_occ]
    W = 2.0 * (Cocc * eps[:n_occ][None, :]) @ Cocc.T                         # energy-weighted density, SCF.jl:262-275
    jc_timing.run_time = time.perf_counter() - t_all
    jc_timing.converged, jc_timing.scf_energy = converged, E          # set_converge_properties!, SCF.jl:588
    jc_timing.non_timing_data[JCTC.contraction_algorithm] = "dense hip" if dense else "screened hip"
    u = ao_back                                                            # back to the caller's AO order
    sym = lambda M: np.ascontiguousarray(M[np.ix_(u, u)])
    out = {"Fock": sym(scf.F.cpu().numpy()), "Density": sym(scf.D.cpu().numpy()), "Energy-Weighted Density": sym(W),
           "MO Coeff": np.ascontiguousarray(C[u, :]), "Overlap": sym(S), "Energy": E, "Converged?": converged, "Timings": jc_timing,
           "Orbital Energies": eps, "Iterations": it, "Nuclear Repulsion": E_nuc, "Trail": list(scf.trail),
           "Kernel Stats": fb.h.kernel_stats(), "Device Bytes": fb.h.device_bytes(), "Iteration Times": it_times,
           "Eigensolver": scf.solver_report(),
           "Density Solver": {"name": scf.density_solver, "sp2_steps": scf.sp2_steps, "sp2_fallbacks": scf.sp2_fallbacks, "sp2_basis_retries": scf.sp2_basis_retries, "sp2_accelerated_steps": scf.sp2_accelerated, "sp2_reference_refreshes": scf.sp2_refreshes,
                              "sp2_fallback_reasons": dict(scf.sp2_reasons)}}
    fb.close()
    eng.close()
    return out
