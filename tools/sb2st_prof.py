#!/usr/bin/env python3
"""Where the waves of the bulge-chasing kernel spend their cycles (diagnostic build with -DJCDF_SB2ST_PROFILE:
  hipcc ... -DJCDF_SB2ST_PROFILE -o juliachem.jl_amd/lib/libjcdf_hip_prof.so;  JCDF_LIB_PATH=that python tools/sb2st_prof.py [n])."""
import ctypes as C
import os
import sys
import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import juliachem_jl_amd as jc   # noqa: E402

lib = jc._lib.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 510
dev = torch.device("cuda", 0)
rng = np.random.default_rng(n)
A = rng.standard_normal((n, n)); A = 0.5 * (A + A.T)
f64 = dict(dtype=torch.float64, device=dev)
wb = int(lib.jcdf_sytrd2_workspace_bytes(n))
work = torch.zeros(wb // 8 + 8, **f64)
D = torch.zeros(n, **f64); E = torch.zeros(n, **f64); Q = torch.zeros((n, n), **f64)
p = lambda t: C.c_void_p(t.data_ptr())
st = torch.cuda.current_stream(dev).cuda_stream
out = (C.c_ulonglong * 128)()
for rep in range(3):
    dA = torch.as_tensor(A, device=dev)
    torch.cuda.synchronize()
    lib.jcdf_sb2st_profile(None, 1)
    lib.jcdf_sytrd2_device(C.c_void_p(st), n, p(dA), n, p(D), p(E), p(Q), n, p(work), wb)
    torch.cuda.synchronize()
lib.jcdf_sb2st_profile(out, 0)
a = np.array(list(out), dtype=np.float64).reshape(16, 8)
print("n = %d; s_memtime ticks per wave (100 MHz constant clock? compare 'kernel' with the rocprof duration)" % n)
print("wave   waiting    in steps   steps   per step   prologues    kernel   ring wait   (two-wave kernel: even waves = chain, odd = update)")
for w in range(16):
    print("%4d %10.0f %10.0f %7.0f %9.1f %10.0f %10.0f %10.0f" % (w, a[w, 0], a[w, 1], a[w, 2], a[w, 1] / max(a[w, 2], 1), a[w, 3], a[w, 4], a[w, 5]))
