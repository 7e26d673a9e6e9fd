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
M = torch.randn((N, N), dtype=torch.float64, device=dev); M = M + M.T
big = torch.empty(64 * 1024 * 1024, dtype=torch.float64, device=dev)   # 512 MB
def run(label, between):
    ks = {}
    for it in range(8):
        between()
        fb.build(Ct); torch.cuda.synchronize()
        for s in fb.h.kernel_stats():
            ks.setdefault(s["name"], []).append(s["seconds"] * 1e3)
    print(label, {k: round(float(np.median(v[2:])), 3) for k, v in ks.items() if "ex" in k or "cou" in k})
run("back-to-back      ", lambda: None)
run("eigh between      ", lambda: torch.linalg.eigh(M))
run("sleep 10 ms       ", lambda: time.sleep(0.01))
run("512 MB memset btw ", lambda: big.zero_())
run("back-to-back again", lambda: None)
