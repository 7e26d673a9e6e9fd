#!/usr/bin/env python3
"""The J || K phase of a Fock build (C20H42 shape) against J's workgroup budget while it runs beside K: per setting the
times of K (event 3 -> 5), J (3 -> 4), the window max(J, K) and the whole build; first the two kernels alone.
usage: jk_sweep.py [config]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from juliachem_jl_amd import synthetic
from juliachem_jl_amd.engine import DeviceFockBuilder

cfg = sys.argv[1] if len(sys.argv) > 1 else "C20H42"
N, Q, o = synthetic.CONFIGS[cfg]
rng = np.random.default_rng(1)
dev = torch.device("cuda", 0)
C, _ = np.linalg.qr(rng.standard_normal((N, N)))
Ct = torch.as_tensor(np.ascontiguousarray(C[:, :o].T), device=dev)


def builder(tuning):
    fb = DeviceFockBuilder(N, Q, o, [1] * Q, device=0, tuning=tuning)
    fb.h.set_metric_inverse(np.eye(Q))
    g = torch.Generator(device=dev); g.manual_seed(7)
    for s0 in range(0, Q, 256):
        s1 = min(Q, s0 + 256)
        A = torch.randn((N, N, s1 - s0), dtype=torch.float64, device=dev, generator=g) * 0.1
        fb.push_three_center_device(s0, s1, (0.5 * (A + A.transpose(0, 1))).contiguous().reshape(-1))
    torch.cuda.synchronize()
    return fb


def measure(fb, reps=12):
    for _ in range(3):
        fb.build(Ct)
    ts = []
    for _ in range(reps):
        fb.build(Ct)
        t = fb.h.synchronize()
        ts.append((t.fock_time, t.W_time, t.J_time, t.K_time))
    a = np.median(np.array(ts), axis=0) * 1e3
    return a


fb = builder({})
fb.h.set_overlap(False)
a = measure(fb)
print("one after the other: build %.3f ms  W %.3f  J %.3f  K %.3f" % tuple(a), flush=True)
fb.close()
for kfirst in [int(x) for x in os.environ.get('JK_KFIRST', '0,1').split(',')]:
    for wg in [int(x) for x in os.environ.get('JK_WG', '0,32,64,96,128,192,256,384,512,768,1024').split(',')]:
        fb = builder({"j_workgroups": wg, "k_first": kfirst})
        a = measure(fb)
        print("k_first=%d j_workgroups=%4d: build %.3f ms  W %.3f  J(3->4) %.3f  K(3->5) %.3f  window %.3f" % (kfirst, wg, a[0], a[1], a[2], a[3], max(a[2], a[3])),
              flush=True)
        fb.close()
