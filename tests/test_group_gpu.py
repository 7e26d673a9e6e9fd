"""GPU tests of the multi-device group of the C ABI (include/jcdf.h, jcdf_group_*): all devices of one process behind
one call, the partial Fock matrices summed ON THE DEVICES (reference: a task, an H2D and a D2H per device and a host
axpy!, GPUDF.jl:188-193, 206, 267-277).  The box has ONE GPU: a group of one device runs both transports ("peer" and
"rccl" — ncclCommInitAll / ncclReduceScatter of librccl.so.1 with one rank), and groups whose members share the device
run the whole multi-member logic (one upload of C, the metric and the T blocks, concurrent builds, the fixed-order slice
sums, the slice-wise D2H) with the "peer" transport.  More than one PHYSICAL device cannot be exercised here: RCCL
refuses duplicate devices (asserted below) — that tier is the driver's 8-GPU run."""
import ctypes

import numpy as np
import pytest

import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from oracle import df_fock as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-11


def _rel(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def _shards(s, n):
    offs = orc.shard_offsets(s.aux_shell_nbas, n)
    return [int(x) for x in offs]


def _setup_group(s, N, Q, o, devices, transport=None, pq=None, sd=None):
    g = jc.JCDFGroup(devices)
    if transport:
        g.set_transport(transport)
    offs = _shards(s, len(devices))
    g.configure(N, Q, offs, o, *(pq or (None, None)))
    g.set_metric(np.tril(s.J2c))
    T = np.asfortranarray(s.T.reshape(Q, N * N, order="F")) if sd is None else orc.pack_three_center(s.T, sd)
    for b in range(len(devices)):                                    # every block once, through the group
        g.push_three_center(offs[b], offs[b + 1], np.asfortranarray(T[offs[b]:offs[b + 1]]))
    g.set_core_hamiltonian(s.H)
    return g, offs, T


def _single_handles(s, N, Q, o, offs, T, pq=None):
    hs = []
    for r in range(len(offs) - 1):
        h = jc.JCDFHandle(0)
        h.configure(N, Q, offs[r], offs[r + 1], o, *(pq or (None, None)))
        h.set_metric(np.tril(s.J2c))
        for b in range(len(offs) - 1):
            h.push_three_center(offs[b], offs[b + 1], np.asfortranarray(T[offs[b]:offs[b + 1]]))
        h.set_core_hamiltonian(s.H if r == 0 else None)
        hs.append(h)
    return hs


@pytest.mark.parametrize("transport", ["auto", "peer", "rccl"])
def test_group_of_one_device_is_bit_equal_to_the_handle(transport):
    """Done-criterion of VERDICT r03 item 1: a 1-device group == jcdf_fock_build, bit for bit — with the hand-written
    transport and through RCCL itself (one rank: communicator, stream and in-place reduce-scatter are real)."""
    N, Q, o = 70, 120, 9
    s = synthetic.make(N, Q, o, seed=123)
    g, offs, T = _setup_group(s, N, Q, o, [0], transport)
    F, t, gt = g.fock_build(s.C[:, :o])
    F2, _, _ = g.fock_build(s.C[:, :o])                               # second build: the event chain of the first is reused
    hs = _single_handles(s, N, Q, o, offs, T)
    Fh, th = hs[0].fock_build(s.C[:, :o])
    ref = s.H + orc.df_rhf_fock_build_BLAS(orc.calculate_B(s.J2c, s.T), s.C[:, :o])
    assert np.array_equal(F, Fh) and np.array_equal(F2, Fh)
    assert _rel(F, ref) < RTOL
    name = g.transport()
    assert name.startswith("rccl " if transport == "rccl" else "peer"), name
    if transport == "rccl":
        assert "1 ranks" in name and "librccl" in name
    assert len(t) == 1 and t[0].fock_time > 0 and t[0].W_time > 0 and gt.total_time > 0 and gt.build_time > 0
    hs[0].close()
    g.close()


@pytest.mark.parametrize("n,kept", [(2, None), (3, None), (3, 0.5), (8, None)])
def test_group_with_members_sharing_the_device_sums_in_member_order(n, kept):
    """n aux shards as n members on the one GPU, "peer" transport: the group's F equals ((F_0 + F_1) + F_2) + ... of n
    independent handles BIT FOR BIT (fixed member order, one writer per element), and the oracle to 1e-11."""
    N, Q, o = 96, 157, 11
    s = synthetic.make(N, Q, o, seed=21, kept_fraction=kept)
    sd = orc.get_screening_metadata(s.mask) if kept else None
    pq = (sd.pq_p, sd.pq_q) if kept else None
    g, offs, T = _setup_group(s, N, Q, o, [0] * n, None, pq, sd)
    assert g.transport().startswith("peer") and "shared" in g.transport()
    Co = s.C[:, :o]
    F, t, gt = g.fock_build(Co)
    hs = _single_handles(s, N, Q, o, offs, T, pq)
    total = None
    for i, h in enumerate(hs):
        Fh, _ = h.fock_build(Co)
        total = Fh if total is None else total + Fh
        # the members hold the B of independent handles: metric factored once, L^-1 rows and T blocks fetched device-to-device
        assert np.array_equal(g.members[i].get_B(), h.get_B())
        h.close()
    assert np.array_equal(F, total)
    if kept:
        Bp = orc.pack_three_center(orc.calculate_B(s.J2c, s.T), sd)
        ref = s.H + orc.df_rhf_fock_build_screened(Bp, Co, sd)
    else:
        ref = orc.df_rhf_fock_build([orc.calculate_B(s.J2c, s.T)], s.C, o, s.H)
    assert _rel(F, ref) < RTOL
    assert len(t) == n and all(x.fock_time > 0 for x in t)
    assert gt.reduce_time > 0 and gt.bcast_time > 0 and gt.d2h_time >= 0
    F2, _, _ = g.fock_build(Co)
    assert np.array_equal(F2, F)
    g.close()


def test_group_device_entry_writes_the_reduced_matrix_with_a_leading_dimension():
    """jcdf_group_fock_build_device_ld: C and F of a device-resident caller (zero padded, on a stream of its own)."""
    import torch
    N, Q, o = 70, 113, 9
    s = synthetic.make(N, Q, o, seed=5)
    g, offs, T = _setup_group(s, N, Q, o, [0, 0, 0])
    Co = s.C[:, :o]
    F_host, _, _ = g.fock_build(Co)
    dev = torch.device("cuda", 0)
    ldc, ldf = 96, 128
    Cp = torch.zeros((32, ldc), dtype=torch.float64, device=dev)
    Cp[:o, :N] = torch.as_tensor(Co.T.copy(), device=dev)
    Fp = torch.full((ldf, ldf), 7.0, dtype=torch.float64, device=dev)
    side = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(side):
        g.fock_build_device_ld(Cp.data_ptr(), ldc, Fp.data_ptr(), ldf, side.cuda_stream)
        out = Fp.clone()                                              # ordered behind the build on the caller's stream
    side.synchronize()
    t, gt = g.synchronize()
    got = out[:N, :N].cpu().numpy().T                                 # column p at d_F + ldf p
    assert np.array_equal(got, F_host)
    assert float(out[N:, :].sub(7.0).abs().max()) == 0.0 and float(out[:, N:].sub(7.0).abs().max()) == 0.0
    assert len(t) == 3 and gt.build_time > 0
    g.close()


def test_group_errors_are_loud_and_there_is_no_host_fallback():
    N, Q, o = 40, 60, 4
    s = synthetic.make(N, Q, o, seed=1)
    g = jc.JCDFGroup([0, 0])
    with pytest.raises(jc.JCDFError) as e:
        g.set_transport("rccl")                                       # RCCL needs distinct devices
    assert e.value.code == 1 and "share a device" in str(e.value)
    with pytest.raises(jc.JCDFError):
        g.set_transport("mpi")
    g.set_transport("peer")
    with pytest.raises(jc.JCDFError):
        g.fock_build(np.zeros((N, o)))                                # not configured
    offs = _shards(s, 2)
    with pytest.raises(jc.JCDFError):
        g.configure(N, Q, [0, 0, Q], o)                               # empty shard
    g.configure(N, Q, offs, o)
    with pytest.raises(jc.JCDFError) as e:
        g.fock_build(s.C[:, :o])                                      # B not set
    assert "B not set" in str(e.value)
    with pytest.raises(jc.JCDFError) as e:
        g.set_metric(-np.eye(Q))
    assert e.value.code == 5
    g.close()
    with pytest.raises(jc.JCDFError):
        jc.JCDFGroup([0, 99])                                         # no such device
    with pytest.raises(jc.JCDFError):
        jc.JCDFGroup([])


def test_operator_with_num_devices_2_goes_through_the_group(monkeypatch):
    """df_rhf_fock_build with num_devices = 2 (GPUDF.jl:188-277): the devices are one JCDFGroup; the reduce transport is
    recorded in the timing object."""
    monkeypatch.setenv("JCDF_ALLOW_DEVICE_WRAP", "1")
    N, Q, o = 70, 113, 9
    s = synthetic.make(N, Q, o, seed=8)
    bs = jc.CalculationBasisSets(jc.basis_from_shell_sizes([N], nels=2 * o), jc.basis_from_shell_sizes(s.aux_shell_nbas))
    eng = jc.TensorIntegralEngine(s.J2c, s.T)
    opts = jc.create_scf_options({"scf_type": "df", "contraction_mode": "HIP", "num_devices": 2})
    scf_data = jc.SCFData(jc.get_default_gpu_data_hip())
    tm = jc.create_jctiming()
    for it in (1, 2):
        F = jc.df_rhf_fock_build(scf_data, eng, None, bs, s.C, it, opts, s.H, tm)
    ref = s.H + orc.df_rhf_fock_build_BLAS(orc.calculate_B(s.J2c, s.T), s.C[:, :o])
    assert _rel(F, ref) < RTOL
    assert scf_data.gpu_data.group is not None and len(scf_data.gpu_data.handles) == 2
    assert tm.non_timing_data["GPU_reduce_transport"].startswith("peer")
    assert tm.non_timing_data["contraction_algorithm"] == "dense hip"       # adaptive rule: N < 800 on one rank
    assert "GPU_2_K_time-2" in tm.timings and tm.timings["fock_gpu_cpu_copy_reduce_time-2"] > 0
    scf_data.gpu_data.close()


def _operator_rank(rank, world, port, out, num_devices):
    import os
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", JCDF_ALLOW_DEVICE_WRAP="1")
    import torch, torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        N, Q, o = 60, 97, 7
        s = synthetic.make(N, Q, o, seed=12, kept_fraction=0.6)
        bs = jc.CalculationBasisSets(jc.basis_from_shell_sizes([N], nels=2 * o), jc.basis_from_shell_sizes(s.aux_shell_nbas))
        eng = jc.TensorIntegralEngine(s.J2c, s.T, mask=s.mask)
        opts = jc.create_scf_options({"scf_type": "df", "contraction_mode": "GPU", "num_devices": num_devices})
        scf_data = jc.SCFData(jc.get_default_gpu_data_hip())
        tm = jc.create_jctiming()
        F = jc.df_rhf_fock_build(scf_data, eng, None, bs, s.C, 1, opts, s.H, tm)
        sd = orc.get_screening_metadata(s.mask)
        Bp = orc.pack_three_center(orc.calculate_B(s.J2c, s.T), sd)
        ref = s.H + orc.df_rhf_fock_build_screened(Bp, s.C[:, :o], sd)
        rows = [len(r) for r in scf_data.gpu_data.device_Q_indices]
        out.put((rank, float(_rel(F, ref)), tm.non_timing_data["contraction_algorithm"], int(tm.non_timing_data["B_exchange_doubles_sent"]),
                 int(tm.non_timing_data["B_exchange_doubles_received"]), rows, int(sd.screened_indices_count),
                 tm.non_timing_data.get("GPU_reduce_transport", "")))
        scf_data.gpu_data.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("num_devices", [1, 2])
def test_reference_shaped_operator_on_two_ranks(num_devices):
    """df_rhf_fock_build on two processes (gloo rehearsal on the one GPU), one or two devices per rank (global device id =
    rank * num_devices + dev, GPUDF.jl:1026-1056; two devices per rank = one jcdf_group per rank): multi-rank runs take the
    screened layout (DensityFitting.jl:78-90); the one-time B formation inside calculate_B_GPU is the point-to-point lower
    triangle (rank 0 sends its block to rank 1 and receives nothing), F is all-reduced across the ranks."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_operator_rank, args=(r, 2, port, out, num_devices)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(240)
        assert p.exitcode == 0
    got = sorted(out.get(timeout=5) for _ in range(2))
    for rank, err, algo, sent, recv, rows, P, transport in got:
        assert err < RTOL and algo == "screened hip" and len(rows) == 2 * num_devices and sum(rows) == 97
        own0 = sum(rows[:num_devices])
        assert (sent, recv) == ((own0 * P, 0) if rank == 0 else (0, own0 * P))
        assert transport.startswith("peer") == (num_devices == 2)


@pytest.mark.parametrize("solver", ["eigh", "sp2"])
def test_rhf_run_with_num_devices_3_is_one_process_over_a_group(solver, monkeypatch):
    """rhf.run with the reference's scf flag num_devices (one rank, several GPUs; here three members on the one GPU): the device
    SCF loop on the first device, every Fock build through jcdf_group_fock_build_device_ld — same energy and iteration count as
    the one-device run to 1e-10 Eh, with either density solver."""
    import json, os
    from juliachem_jl_amd import rhf
    from water_case import GOLDEN, FIXTURES
    monkeypatch.setenv("JCDF_ALLOW_DEVICE_WRAP", "1")
    g = json.load(open(os.path.join(GOLDEN, FIXTURES["ccpvdz"])))
    atoms = list(g["atoms"]) + [{"symbol": a["symbol"], "center": [a["center"][0] + 0.3, a["center"][1] + 7.0, a["center"][2] + 1.1]}
                                for a in g["atoms"]]
    f = {"dele": 1e-8, "rmsd": 1e-8, "niter": 60, "density_solver": solver}
    one = rhf.run(atoms, g["charges"], g["basis"], g["aux_basis"], f)
    three = rhf.run(atoms, g["charges"], g["basis"], g["aux_basis"], dict(f, num_devices=3))
    assert three["Converged?"] and three["Iterations"] == one["Iterations"]
    assert abs(three["Energy"] - one["Energy"]) < 1e-10
    assert three["Timings"].non_timing_data["GPU_num_devices"] == "3"
    assert three["Timings"].non_timing_data["GPU_reduce_transport"].startswith("peer")
    assert np.abs(three["Density"] - one["Density"]).max() < 1e-8
