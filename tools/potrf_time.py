"""Time the metric factorisation: library device path vs scipy LAPACK vs library host path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import juliachem_jl_amd as jc
for n in (1950, 4800):
    rng = np.random.default_rng(1)
    M = rng.standard_normal((n, n)); A = M @ M.T + n * np.eye(n)
    jc.device_potrf_trtri(np.tril(A[:256, :256]))                 # warm the context
    t0 = time.perf_counter(); X = jc.device_potrf_trtri(np.tril(A)); t1 = time.perf_counter()
    Xl = jc.lapack_potrf_trtri(np.tril(A)); t2 = time.perf_counter()
    print("n=%d device %.1f ms (incl. H2D/D2H + alloc)  lapack %.1f ms  maxdiff %.2e" % (n, 1e3 * (t1 - t0), 1e3 * (t2 - t1), np.abs(X - Xl).max() / np.abs(Xl).max()), flush=True)
    if "--host" in sys.argv:
        t0 = time.perf_counter(); jc.host_potrf_trtri(np.tril(A)); print("  host lib %.1f ms" % (1e3 * (time.perf_counter() - t0)))
