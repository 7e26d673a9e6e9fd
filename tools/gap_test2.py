"""What does the Fock build need from the phase before it?  Each build is preceded by ~3 ms of (a) nothing (host sleep),
(b) an HBM stream (device copy), (c) fp64 GEMMs from L2, (d) the replicated eigensolver itself (the SCF loop's case),
(e) the eigensolver with an HBM stream beside it on a second stream."""
import os, sys, time
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
src = torch.randn(1 << 28, dtype=torch.float64, device=dev)       # 2 GB
dst = torch.empty_like(src)
Ma = torch.randn((2048, 2048), dtype=torch.float64, device=dev); Mb = torch.randn((2048, 2048), dtype=torch.float64, device=dev)
eig = DeviceEigh(N, dev)
S = torch.randn((N, N), dtype=torch.float64, device=dev); S = S + S.T
side = torch.cuda.Stream(device=dev)


def pre(mode):
    if mode == "sleep 3 ms":
        torch.cuda.synchronize(); time.sleep(3e-3)
    elif mode == "HBM stream 3 ms":
        for _ in range(3): dst.copy_(src)
    elif mode == "fp64 GEMM 3 ms":
        for _ in range(10): torch.mm(Ma, Mb)
    elif mode == "eigensolve":
        eig(S)
    elif mode.startswith("eigensolve + fp64 GEMMs beside it"):
        n = int(mode.split()[-1])
        ev = torch.cuda.Event(); ev.record()
        eig(S)
        with torch.cuda.stream(side):
            side.wait_event(ev)
            for _ in range(n): torch.mm(Ma, Mb)
        torch.cuda.current_stream().wait_stream(side)
    elif mode == "eigensolve + HBM stream beside it":
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2): dst.copy_(src)
        eig(S)
        torch.cuda.current_stream().wait_stream(side)


for mode in ("back-to-back", "sleep 3 ms", "HBM stream 3 ms", "fp64 GEMM 3 ms", "eigensolve", "eigensolve + HBM stream beside it",
             "eigensolve + fp64 GEMMs beside it 4", "eigensolve + fp64 GEMMs beside it 8", "back-to-back"):
    for _ in range(5): fb.build(Ct)
    torch.cuda.synchronize()
    fb.h.kernel_stats_total(reset=True)
    t0 = time.perf_counter()
    for _ in range(30):
        pre(mode)
        fb.build(Ct)
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / 30 * 1e3
    recs, nb, fs = fb.h.kernel_stats_total(reset=True)
    ks = {r["name"]: r["seconds"] / max(nb, 1) * 1e3 for r in recs}
    print("%-36s W %.3f ms  K %.3f ms  J %.3f  build %.3f ms  (cycle %.2f ms, %d builds)" % (mode, ks["k_exchange_W"], ks["k_exchange_K"], ks["k_coulomb_J"], fs / max(nb, 1) * 1e3, wall, nb), flush=True)
