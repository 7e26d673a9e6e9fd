"""Is a Fock build slower when it starts from an idle GPU?  back-to-back builds vs a host sync (+ optional sleep) before each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import juliachem_jl_amd as jc
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder
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
for mode, gap in (("back-to-back", None), ("sync before each", 0.0), ("sync + 1 ms sleep", 1e-3), ("sync + 3 ms sleep", 3e-3), ("back-to-back", None)):
    for _ in range(5): fb.build(Ct)
    torch.cuda.synchronize()
    W, K, J = [], [], []
    for _ in range(30):
        if gap is not None:
            torch.cuda.synchronize()
            if gap: time.sleep(gap)
        fb.build(Ct)
        if gap is not None:
            ks = {k["name"]: k["seconds"] * 1e3 for k in fb.h.kernel_stats()}
            W.append(ks["k_exchange_W"]); K.append(ks["k_exchange_K"])
    torch.cuda.synchronize()
    if gap is None:
        ks = {k["name"]: k["seconds"] * 1e3 for k in fb.h.kernel_stats()}; W, K = [ks["k_exchange_W"]], [ks["k_exchange_K"]]
    print("%-20s W %.3f ms  K %.3f ms" % (mode, np.median(W), np.median(K)), flush=True)
