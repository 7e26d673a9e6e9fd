"""What keeps the shader clock up between Fock builds?  Each build is preceded by ~3 ms of: nothing; sleeping waves on every CU
(occupancy, no issue); an fp64 FMA chain per wave; fp64 MFMAs; the replicated eigensolve alone; the eigensolve with each of
the three keep-alive kernels beside it on a second stream (tools/gap_test2.py has the GEMM / HBM-stream variants)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder
from juliachem_jl_amd.eigh import DeviceEigh
N, Q, o = synthetic.CONFIGS["C20H42"]
rng = np.random.default_rng(1); dev = torch.device("cuda", 0)
fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0)
fb.h.set_metric_inverse(np.eye(Q)); fb.set_core_hamiltonian(np.eye(N))
g = torch.Generator(device=dev); g.manual_seed(7)
for s0 in range(0, Q, 256):
    s1 = min(Q, s0 + 256)
    A = torch.randn((N, N, s1 - s0), dtype=torch.float64, device=dev, generator=g) * 0.1
    fb.push_three_center_device(s0, s1, (0.5 * (A + A.transpose(0, 1))).contiguous().reshape(-1))
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)
eig = DeviceEigh(N, dev)
S = torch.randn((N, N), dtype=torch.float64, device=dev); S = S + S.T
side = torch.cuda.Stream(device=dev)
lib = jc._lib.load()
sink = torch.zeros(8, dtype=torch.float64, device=dev)
stop = torch.zeros(4, dtype=torch.int32, device=dev)


def keep(stream, us, mode, wg=512, thr=256, pause=0, use_stop=False):
    rc = lib.jcdf_keepalive_device(ctypes.c_void_p(stream.cuda_stream), wg, thr, float(us), mode, pause,
                                   ctypes.c_void_p(stop.data_ptr()) if use_stop else None, ctypes.c_void_p(sink.data_ptr()))
    assert rc == 0


eig_ms = []


def pre(mode):
    main = torch.cuda.current_stream(dev)
    if mode == "idle 3 ms":
        torch.cuda.synchronize(); time.sleep(3e-3)
    elif mode.startswith("keepalive"):
        keep(main, 3000.0, int(mode.split()[1]))
    elif mode == "eigensolve":
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eig(S); e1.record()
        eig_ms.append((e0, e1))
    elif mode.startswith("eig, burst"):
        # how long must the activity last?  eigensolve, then FMA chains on every CU for so many microseconds, then the build
        eig(S)
        keep(main, float(mode.split()[-1]), 1)
    elif mode.startswith("eig+keep"):
        # keep-alive waves beside the eigensolve until it is done: mode, workgroups, threads, pause
        _, m, wg, thr, pause = mode.split()
        stop.zero_()
        ev = torch.cuda.Event(); ev.record()
        side.wait_event(ev)
        keep(side, 20000.0, int(m), int(wg), int(thr), int(pause), True)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); eig(S); e1.record()
        eig_ms.append((e0, e1))
        stop.fill_(1)                                               # (a fill kernel on the main stream, behind the eigensolve)
        main.wait_stream(side)


modes = ["back-to-back", "idle 3 ms", "keepalive 0", "keepalive 1", "keepalive 2", "eigensolve"]
modes += ["eig, burst %d" % us for us in (50, 100, 200, 400, 800, 1600)]
if os.environ.get("GAP3_SHORT"):
    modes = ["back-to-back", "eigensolve"] + ["eig, burst %d" % us for us in (50, 100, 200, 400, 800, 1600)]
for m in (1, 2):
    for wg, thr in ((256, 64), (512, 64), (256, 256), (512, 256)):
        for pause in (0, 16):
            if not os.environ.get("GAP3_SHORT"):
                modes.append("eig+keep %d %d %d %d" % (m, wg, thr, pause))
modes.append("back-to-back")
for mode in modes:
    for _ in range(5): fb.build(Ct)
    torch.cuda.synchronize()
    fb.h.kernel_stats_total(reset=True)
    eig_ms.clear()
    t0 = time.perf_counter()
    for _ in range(30):
        pre(mode)
        fb.build(Ct)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30 * 1e3
    recs, nb, fs = fb.h.kernel_stats_total(reset=True)
    ks = {r["name"]: r["seconds"] / max(nb, 1) * 1e3 for r in recs}
    em = np.median([a.elapsed_time(b) for a, b in eig_ms]) if eig_ms else float("nan")
    print("%-28s W %.3f ms  K %.3f ms  J %.3f  build %.3f ms  eigensolve %.3f ms  (cycle %.2f ms)"
          % (mode, ks["k_exchange_W"], ks["k_exchange_K"], ks["k_coulomb_J"], fs / max(nb, 1) * 1e3, em, wall), flush=True)
