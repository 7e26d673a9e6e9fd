#!/usr/bin/env python3
"""Where the waves of the W kernel spend their cycles: s_memtime stamps around the five segments of a
phase in a diagnostic build of k_exchange_W_dma (template bit 32), C20H42 shape, 6 MFMA row tiles.
usage (GPU box): JCDF_W_ABLATE=32 JCDF_W_REM=0 python tools/w_stall.py
W_STALL_GAP_MS=3: every build is preceded by that many ms of idle device (the state a build meets inside the SCF loop)."""
import os, sys
os.environ.setdefault("JCDF_W_ABLATE", "32"); os.environ.setdefault("JCDF_W_REM", "0")
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes
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
import time
gap = float(os.environ.get("W_STALL_GAP_MS", "0")) * 1e-3
for _ in range(10):
    if gap > 0:
        torch.cuda.synchronize()
        time.sleep(gap)
    fb.build(Ct)
torch.cuda.synchronize()
if gap > 0:
    print("(every build after %.1f ms of idle device)" % (gap * 1e3))
ks = {k["name"]: k["seconds"] * 1e3 for k in fb.h.kernel_stats()}
lib = jc._lib.load()
buf = np.zeros((200000, 6), dtype=np.uint64)
n = int(lib.jcdf_w_stall_cycles(fb.h._h, buf.ctypes.data, buf.shape[0]))
if n == 0:
    raise SystemExit("diagnostic build not active (needs JCDF_W_ABLATE=32 JCDF_W_REM=0 and 81..96 occupied orbitals)")
b = buf[:n].astype(np.float64)
b = b[b[:, 5] > 0]
per = b[:, :5] / b[:, 5:6]
tot = per.sum(axis=1)
names = ["DMA issue (addresses + 4 global_load_lds)", "operand ds_reads + MFMA issue", "index loads + end-of-p epilogue", "counted vmcnt wait", "barrier"]
print("k_exchange_W_dma<6,1,2> with s_memtime stamps, C20H42 shape: %.3f ms per launch (stamps cost cycles: 1.70 without)" % ks["k_exchange_W"])
print("%d waves, %.0f phases each (median); shader cycles per phase per wave, median over waves [10 %% .. 90 %%]:" % (len(b), np.median(b[:, 5])))
for k, nm in enumerate(names):
    print("  %-46s %7.0f  [%6.0f .. %6.0f]   %4.1f %%" % (nm, np.median(per[:, k]), np.percentile(per[:, k], 10), np.percentile(per[:, k], 90),
                                                        100.0 * np.median(per[:, k]) / np.median(tot)))
print("  %-46s %7.0f   (24 MFMAs of 64 cycles = 1536 issue cycles per wave and phase; 3 waves share a SIMD)" % ("phase", np.median(tot)))
fb.close()
