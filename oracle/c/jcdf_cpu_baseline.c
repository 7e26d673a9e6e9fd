/* CPU baseline of the DF-RHF Fock build: plain-C restatement of the reference's two CPU modes on top of the best host
 * BLAS found at run time (dlopen: Intel MKL's libmkl_rt.so, or the OpenBLAS that ships inside numpy).
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (header of oracle/README.md applies): built into oracle/_build/libjcdf_cpu_baseline.so
 * and called by bench.py's cpu_baseline leg and by tests/ as a checker.  Never linked into libjcdf_hip.so.
 *
 *   jcbl_fock_dense      df_rhf_fock_build_BLAS!   DensityFitting.jl:111-125,185-224: five BLAS calls on the dense
 *                        (Q, N, N) tensor, all BLAS threads: density gemm, V gemv, J gemv^T, W gemm, K gemm.
 *   jcbl_fock_screened   df_rhf_fock_build_screened!  ScreenedDF.jl:80-132 — the reference's DEFAULT CPU mode:
 *                        W per p with a single-threaded gemm, p distributed over threads (:242-289); K from the lower
 *                        triangle of n_blocks x n_blocks blocks with all BLAS threads, mirrored, ragged remainder strip
 *                        (:548-641, n_blocks = 10 :392-395); Coulomb by per-p gemv over the lower-triangle runs of the
 *                        packed (Q, P) tensor (:318-365); J scattered into F (:367-378).
 * Layouts are the reference's (Julia column-major, first index fastest).  Times of the five steps are returned so the
 * bench line can show where the CPU spends its time. */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <omp.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

typedef void (*gemm32_t)(const char *, const char *, const int *, const int *, const int *, const double *, const double *,
                         const int *, const double *, const int *, const double *, double *, const int *);
typedef void (*gemv32_t)(const char *, const int *, const int *, const double *, const double *, const int *, const double *,
                         const int *, const double *, double *, const int *);
typedef void (*gemm64_t)(const char *, const char *, const int64_t *, const int64_t *, const int64_t *, const double *,
                         const double *, const int64_t *, const double *, const int64_t *, const double *, double *,
                         const int64_t *);
typedef void (*gemv64_t)(const char *, const int64_t *, const int64_t *, const double *, const double *, const int64_t *,
                         const double *, const int64_t *, const double *, double *, const int64_t *);
typedef void (*setthr_t)(int);
typedef int (*setthr_local_t)(int);

static void *g_lib;
static int g_ilp64;
static gemm32_t g_gemm32;
static gemv32_t g_gemv32;
static gemm64_t g_gemm64;
static gemv64_t g_gemv64;
static setthr_t g_set_threads;
static setthr_local_t g_set_threads_local;
static char g_name[256];

static double now(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

/* kind 0: MKL (LP64 dgemm_/dgemv_, MKL_Set_Num_Threads[_Local]); kind 1: numpy's OpenBLAS (ILP64 scipy_dgemm_64_ ...) */
int jcbl_load_blas(const char *path, int kind)
{
    /* never dlclose: a BLAS with a live thread pool (MKL, OpenBLAS) does not survive being unloaded and mapped again */
    g_lib = NULL;
    g_gemm32 = NULL; g_gemv32 = NULL; g_gemm64 = NULL; g_gemv64 = NULL; g_set_threads = NULL; g_set_threads_local = NULL;
    g_lib = dlopen(path, RTLD_NOW | RTLD_LOCAL | RTLD_NODELETE);
    if (!g_lib) return 1;
    if (kind == 0) {
        g_ilp64 = 0;
        /* MKL on the GNU OpenMP runtime this file's own parallel regions use (libmkl_gnu_thread): with the default Intel
         * layer two OpenMP runtimes (libiomp5 + libgomp) share the process — oversubscribed, and occasionally wrong */
        typedef int (*layer_t)(int);
        layer_t set_layer = (layer_t)dlsym(g_lib, "MKL_Set_Threading_Layer");
        if (set_layer) set_layer(1 /* MKL_THREADING_GNU */);
        g_gemm32 = (gemm32_t)dlsym(g_lib, "dgemm_");
        g_gemv32 = (gemv32_t)dlsym(g_lib, "dgemv_");
        g_set_threads = (setthr_t)dlsym(g_lib, "MKL_Set_Num_Threads");
        g_set_threads_local = (setthr_local_t)dlsym(g_lib, "MKL_Set_Num_Threads_Local");
        if (!g_gemm32 || !g_gemv32) return 2;
        snprintf(g_name, sizeof g_name, "Intel MKL (libmkl_rt, LP64)");
    } else {
        g_ilp64 = 1;
        g_gemm64 = (gemm64_t)dlsym(g_lib, "scipy_dgemm_64_");
        g_gemv64 = (gemv64_t)dlsym(g_lib, "scipy_dgemv_64_");
        g_set_threads = (setthr_t)dlsym(g_lib, "scipy_openblas_set_num_threads64_");
        g_set_threads_local = (setthr_local_t)dlsym(g_lib, "scipy_openblas_set_num_threads_local64_");
        if (!g_gemm64 || !g_gemv64) return 2;
        snprintf(g_name, sizeof g_name, "OpenBLAS (numpy's scipy_openblas64, ILP64)");
    }
    return 0;
}

const char *jcbl_blas_name(void) { return g_lib ? g_name : "none"; }

static void gemm(char ta, char tb, int64_t m, int64_t n, int64_t k, double alpha, const double *A, int64_t lda, const double *B,
                 int64_t ldb, double beta, double *C, int64_t ldc)
{
    if (g_ilp64) {
        g_gemm64(&ta, &tb, &m, &n, &k, &alpha, A, &lda, B, &ldb, &beta, C, &ldc);
    } else {
        int m_ = (int)m, n_ = (int)n, k_ = (int)k, lda_ = (int)lda, ldb_ = (int)ldb, ldc_ = (int)ldc;
        g_gemm32(&ta, &tb, &m_, &n_, &k_, &alpha, A, &lda_, B, &ldb_, &beta, C, &ldc_);
    }
}

static void gemv(char t, int64_t m, int64_t n, double alpha, const double *A, int64_t lda, const double *x, double beta, double *y)
{
    if (g_ilp64) {
        int64_t one = 1;
        g_gemv64(&t, &m, &n, &alpha, A, &lda, x, &one, &beta, y, &one);
    } else {
        int m_ = (int)m, n_ = (int)n, lda_ = (int)lda, one = 1;
        g_gemv32(&t, &m_, &n_, &alpha, A, &lda_, x, &one, &beta, y, &one);
    }
}

static void blas_threads(int n)
{
    if (g_set_threads_local) g_set_threads_local(0);      /* the calling thread follows the global setting again */
    if (g_set_threads) g_set_threads(n);
}

/* GFLOP/s of an n x n x n dgemm with `threads` BLAS threads (best of `reps`) */
double jcbl_dgemm_calibration(int64_t n, int reps, int threads)
{
    if (!g_lib) return 0.0;
    double *A = (double *)malloc((size_t)(n * n) * 8), *B = (double *)malloc((size_t)(n * n) * 8),
           *C = (double *)malloc((size_t)(n * n) * 8);
    if (!A || !B || !C) { free(A); free(B); free(C); return 0.0; }
    for (int64_t i = 0; i < n * n; ++i) { A[i] = (double)((i * 7919) % 1013) * 1e-3 - 0.5; B[i] = (double)((i * 104729) % 997) * 1e-3 - 0.5; }
    blas_threads(threads);
    double best = 1e300;
    for (int r = 0; r < reps + 1; ++r) {
        double t0 = now();
        gemm('N', 'N', n, n, n, 1.0, A, n, B, n, 0.0, C, n);
        double t = now() - t0;
        if (r > 0 && t < best) best = t;                  /* first call warms the thread pool */
    }
    free(A); free(B); free(C);
    return 2.0 * (double)n * (double)n * (double)n / best * 1e-9;
}

/* times[5] = {density, V, J, W, K} seconds.  B (Q, N, N), C_occ (N, o), H/F (N, N). */
int jcbl_fock_dense(int64_t N, int64_t Q, int64_t o, const double *B, const double *C_occ, const double *H, double *F,
                    double *times, int threads)
{
    if (!g_lib) return 1;
    double *density = (double *)malloc((size_t)(N * N) * 8), *V = (double *)malloc((size_t)Q * 8),
           *W = (double *)malloc((size_t)(o * Q * N) * 8);
    if (!density || !V || !W) { free(density); free(V); free(W); return 2; }
    blas_threads(threads);
    double t0 = now();
    gemm('N', 'T', N, N, o, 1.0, C_occ, N, C_occ, N, 0.0, density, N);                 /* :193 */
    double t1 = now();
    gemv('N', Q, N * N, 1.0, B, Q, density, 0.0, V);                                   /* :195 */
    double t2 = now();
    gemv('T', Q, N * N, 2.0, B, Q, V, 0.0, F);                                         /* :198, beta = 0 */
    double t3 = now();
    gemm('T', 'T', o, Q * N, N, 1.0, C_occ, N, B, Q * N, 0.0, W, o);                   /* :216  W (o, Q N) */
    double t4 = now();
    gemm('T', 'N', N, N, o * Q, -1.0, W, o * Q, W, o * Q, 1.0, F, N);                  /* :219  F -= W^T W */
    double t5 = now();
    if (H) for (int64_t i = 0; i < N * N; ++i) F[i] += H[i];
    times[0] = t1 - t0; times[1] = t2 - t1; times[2] = t3 - t2; times[3] = t4 - t3; times[4] = t5 - t4;
    free(density); free(V); free(W);
    return 0;
}

/* Bp (Q, P) packed; start[p], count[p] = first packed index and K_p of column p; qlist[c] = q of packed index c (ascending
 * inside a p); diag[p] = packed index of (p,p) (the lower-triangle run of p is [diag[p], start[p] + count[p]) ).
 * times[5] = {density, V, J, W, K}. */
int jcbl_fock_screened(int64_t N, int64_t Q, int64_t o, int64_t P, const double *Bp, const int64_t *start, const int64_t *count,
                       const int64_t *qlist, const int64_t *diag, const double *C_occ, const double *H, double *F, double *times,
                       int threads, int n_blocks)
{
    if (!g_lib) return 1;
    double *W = (double *)malloc((size_t)(Q * o * N) * 8);          /* (Q, o, N) :110 */
    double *CT = (double *)malloc((size_t)(o * N) * 8);             /* (o, N) :84 */
    double *density = (double *)malloc((size_t)(N * N) * 8), *d = (double *)calloc((size_t)P, 8), *J = (double *)calloc((size_t)P, 8);
    double *V = (double *)calloc((size_t)Q, 8);
    if (!W || !CT || !density || !d || !J || !V) { free(W); free(CT); free(density); free(d); free(J); free(V); return 2; }
    for (int64_t i = 0; i < o; ++i)
        for (int64_t q = 0; q < N; ++q) CT[i + o * q] = C_occ[q + N * i];
    int64_t kmax = 0;
    for (int64_t p = 0; p < N; ++p) if (count[p] > kmax) kmax = count[p];

    /* ---- W: p over threads, single-threaded gemm each (ScreenedDF.jl:242-289) ---- */
    double t0 = now();
    blas_threads(1);
#pragma omp parallel num_threads(threads)
    {
        if (g_set_threads_local) g_set_threads_local(1);
        double *nz = (double *)malloc((size_t)(o * (kmax > 0 ? kmax : 1)) * 8);
#pragma omp for schedule(dynamic, 1)
        for (int64_t p = 0; p < N; ++p) {
            const int64_t kp = count[p], s = start[p];
            double *Wp = W + p * Q * o;
            if (kp == 0) { memset(Wp, 0, (size_t)(Q * o) * 8); continue; }
            for (int64_t j = 0; j < kp; ++j) memcpy(nz + j * o, CT + qlist[s + j] * o, (size_t)o * 8);        /* o x K_p */
            gemm('N', 'T', Q, o, kp, 1.0, Bp + s * Q, Q, nz, o, 0.0, Wp, Q);
        }
        free(nz);
    }
    double t1 = now();

    /* ---- K: lower-triangle blocks, all BLAS threads, beta = 0 (ScreenedDF.jl:548-641) ---- */
    blas_threads(threads);
    if (N < 100) n_blocks = 1;
    const int64_t bw = N / n_blocks, KK = Q * o;
    const int64_t blk_n = (bw * bw > N * (N % n_blocks) ? bw * bw : N * (N % n_blocks)) + 1;
    double *blk = (double *)malloc((size_t)blk_n * 8);
    if (!blk) { free(W); free(CT); free(density); free(d); free(J); free(V); return 2; }
    for (int bi = 0; bi < n_blocks; ++bi)
        for (int bj = 0; bj <= bi; ++bj) {
            gemm('T', 'N', bw, bw, KK, -1.0, W + (int64_t)bi * bw * KK, KK, W + (int64_t)bj * bw * KK, KK, 0.0, blk, bw);
            for (int64_t c = 0; c < bw; ++c)
                for (int64_t r = 0; r < bw; ++r) {
                    const double v = blk[r + bw * c];
                    F[(bi * bw + r) + N * (bj * bw + c)] = v;
                    F[(bj * bw + c) + N * (bi * bw + r)] = v;
                }
        }
    const int64_t rem = N % n_blocks;
    if (rem) {
        gemm('T', 'N', N, rem, KK, -1.0, W, KK, W + (N - rem) * KK, KK, 0.0, blk, N);
        for (int64_t c = 0; c < rem; ++c)
            for (int64_t r = 0; r < N; ++r) {
                F[r + N * (N - rem + c)] = blk[r + N * c];
                F[(N - rem + c) + N * r] = blk[r + N * c];
            }
    }
    free(blk);
    double t2 = now();

    /* ---- Coulomb: density, packed density, V and J over the lower-triangle runs (ScreenedDF.jl:305-365) ---- */
    gemm('T', 'N', N, N, o, 1.0, CT, o, CT, o, 0.0, density, N);
    for (int64_t p = 0; p < N; ++p)
        for (int64_t c = diag[p]; c < start[p] + count[p]; ++c) {
            const int64_t q = qlist[c];
            d[c] = (q != p ? 2.0 : 1.0) * density[q + N * p];
        }
    double t3 = now();
    blas_threads(1);
#pragma omp parallel num_threads(threads)
    {
        if (g_set_threads_local) g_set_threads_local(1);
        double *Vt = (double *)calloc((size_t)Q, 8);
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < N; ++p) {
            const int64_t lo = diag[p], n = start[p] + count[p] - lo;
            if (n > 0) gemv('N', Q, n, 1.0, Bp + lo * Q, Q, d + lo, 1.0, Vt);
        }
#pragma omp critical
        for (int64_t a = 0; a < Q; ++a) V[a] += Vt[a];
        free(Vt);
    }
    double t4 = now();
#pragma omp parallel num_threads(threads)
    {
        if (g_set_threads_local) g_set_threads_local(1);
#pragma omp for schedule(dynamic, 4)
        for (int64_t p = 0; p < N; ++p) {
            const int64_t lo = diag[p], n = start[p] + count[p] - lo;
            if (n > 0) gemv('T', Q, n, 2.0, Bp + lo * Q, Q, V, 0.0, J + lo);
        }
    }
    for (int64_t p = 0; p < N; ++p)                                     /* :367-378 */
        for (int64_t c = diag[p]; c < start[p] + count[p]; ++c) {
            const int64_t q = qlist[c];
            F[q + N * p] += J[c];
            F[p + N * q] = F[q + N * p];
        }
    double t5 = now();
    blas_threads(threads);
    if (H) for (int64_t i = 0; i < N * N; ++i) F[i] += H[i];
    times[0] = t3 - t2; times[1] = t4 - t3; times[2] = t5 - t4; times[3] = t1 - t0; times[4] = t2 - t1;
    free(W); free(CT); free(density); free(d); free(J); free(V);
    return 0;
}
