import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, scipy.linalg as sla
import juliachem_jl_amd as jc
for Q in (1950, 4800):
    rng = np.random.default_rng(0); M = rng.standard_normal((Q, Q)); A = M @ M.T + Q * np.eye(Q)
    t0 = time.perf_counter(); X = jc.host_potrf_trtri(A); t1 = time.perf_counter()
    L = sla.cholesky(A, lower=True); Li = sla.solve_triangular(L, np.eye(Q), lower=True); t2 = time.perf_counter()
    print("Q=%d  library host potrf+trtri %.2f s   scipy %.2f s   max diff %.1e" % (Q, t1 - t0, t2 - t1, np.abs(X - Li).max()))
