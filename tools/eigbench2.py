import ctypes as C, time, os, numpy as np, torch
tl = os.path.join(os.path.dirname(torch.__file__), "lib")
rb = C.CDLL(os.path.join(tl, "librocblas.so")); rs = C.CDLL(os.path.join(tl, "librocsolver.so"))
h = C.c_void_p(); assert rb.rocblas_create_handle(C.byref(h)) == 0
rb.rocblas_set_stream(h, C.c_void_p(torch.cuda.current_stream().cuda_stream))
EVECT_ORIGINAL = 211  # rocblas_evect_original
FILL_LOWER = 122; ESORT_ASC = 231 # rocblas_esort_ascending
for N in (240, 510, 1250):
    rng = np.random.default_rng(0); A0 = rng.standard_normal((N, N)); A0 = A0 + A0.T
    Ag = torch.from_numpy(A0).cuda()
    W = torch.empty(N, dtype=torch.float64, device="cuda"); E = torch.empty(N, dtype=torch.float64, device="cuda")
    info = torch.zeros(1, dtype=torch.int32, device="cuda"); res = torch.zeros(1, dtype=torch.float64, device="cuda"); nsw = torch.zeros(1, dtype=torch.int32, device="cuda")
    def run(name, f, n=5):
        A = Ag.clone(); f(A); torch.cuda.synchronize()
        ts = []
        for _ in range(n):
            A = Ag.clone(); torch.cuda.synchronize(); t0 = time.perf_counter(); rc = f(A); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        w = W.cpu().numpy(); ref = np.linalg.eigvalsh(A0)
        print(N, name, "rc", rc, "ms %.2f" % np.median(ts), "eval err %.1e" % np.abs(np.sort(w) - ref).max(), "info", int(info.item()))
    p = lambda t: C.c_void_p(t.data_ptr())
    run("syevd", lambda A: rs.rocsolver_dsyevd(h, EVECT_ORIGINAL, FILL_LOWER, N, p(A), N, p(W), p(E), p(info)))
    run("syev", lambda A: rs.rocsolver_dsyev(h, EVECT_ORIGINAL, FILL_LOWER, N, p(A), N, p(W), p(E), p(info)))
    try:
        run("syevj", lambda A: rs.rocsolver_dsyevj(h, ESORT_ASC, EVECT_ORIGINAL, FILL_LOWER, N, p(A), N, C.c_double(0.0), p(res), 100, p(nsw), p(W), p(info)))
        print("   sweeps", int(nsw.item()))
    except Exception as e: print("syevj failed", e)
    try:
        run("syevdj", lambda A: rs.rocsolver_dsyevdj(h, EVECT_ORIGINAL, FILL_LOWER, N, p(A), N, p(W), p(info)))
    except Exception as e: print("syevdj failed", e)
