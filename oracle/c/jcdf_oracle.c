/* CPU oracle, plain-C twin of oracle/df_fock.py.
 *
 * TEST INFRASTRUCTURE ONLY: built into oracle/_build/libjcdf_oracle.so and loaded
 * by tests/ and __graft_entry__.smoke() as the checker; never linked into or
 * called from the product library (libjcdf_hip.so).
 *
 * Restates the dense DF Fock build of the reference in the reference's own
 * memory layout (Julia column-major, first index fastest):
 *   B     (Q, N, N)   index Q + Qn*(mu + N*nu)        DensityFitting.jl:148-152
 *   C_occ (N, o)      index mu + N*i                  DensityFitting.jl:49
 *   H, F  (N, N)      index mu + N*nu
 * Steps follow DensityFitting.jl:185-224 one BLAS call at a time, written as
 * explicit loops so the result does not depend on any BLAS library:
 *   density = C_o C_o^T                 (:193, gemm 'N','T')
 *   V[Q]    = sum_pq B[Q,pq] density[pq] (:195, gemv 'N')
 *   F[pq]   = 2 sum_Q B[Q,pq] V[Q]       (:198, gemv 'T', beta = 0)
 *   W[i,Q,mu] = sum_nu C_o[nu,i] B[Q,mu,nu]   (:216, gemm 'T','T')
 *   F[mu,nu] -= sum_{i,Q} W[i,Q,mu] W[i,Q,nu] (:219, gemm 'T','N', beta = 1)
 *   F += H  (rank 0)                     (DensityFitting.jl:62-65)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

int jcdf_oracle_fock_dense(int64_t N, int64_t Q, int64_t o,
                           const double *B, const double *C_occ,
                           const double *H, int add_H, double *F)
{
    double *density = (double *)calloc((size_t)(N * N), sizeof(double));
    double *V = (double *)calloc((size_t)Q, sizeof(double));
    double *W = (double *)calloc((size_t)(o * Q * N), sizeof(double));
    if (!density || !V || !W) { free(density); free(V); free(W); return 1; }

    for (int64_t nu = 0; nu < N; ++nu)
        for (int64_t mu = 0; mu < N; ++mu) {
            double s = 0.0;
            for (int64_t i = 0; i < o; ++i) s += C_occ[mu + N * i] * C_occ[nu + N * i];
            density[mu + N * nu] = s;
        }

    for (int64_t pq = 0; pq < N * N; ++pq) {
        const double d = density[pq];
        const double *col = B + Q * pq;
        for (int64_t q = 0; q < Q; ++q) V[q] += col[q] * d;
    }

    for (int64_t pq = 0; pq < N * N; ++pq) {
        const double *col = B + Q * pq;
        double s = 0.0;
        for (int64_t q = 0; q < Q; ++q) s += col[q] * V[q];
        F[pq] = 2.0 * s;
    }

    /* W (o, Q, N): index i + o*(Q + Qn*mu) */
    for (int64_t nu = 0; nu < N; ++nu)
        for (int64_t mu = 0; mu < N; ++mu) {
            const double *col = B + Q * (mu + N * nu);
            for (int64_t q = 0; q < Q; ++q) {
                const double b = col[q];
                double *w = W + o * (q + Q * mu);
                for (int64_t i = 0; i < o; ++i) w[i] += C_occ[nu + N * i] * b;
            }
        }

    for (int64_t nu = 0; nu < N; ++nu)
        for (int64_t mu = 0; mu < N; ++mu) {
            const double *a = W + o * Q * mu;
            const double *b = W + o * Q * nu;
            double s = 0.0;
            for (int64_t k = 0; k < o * Q; ++k) s += a[k] * b[k];
            F[mu + N * nu] -= s;
        }

    if (add_H)
        for (int64_t pq = 0; pq < N * N; ++pq) F[pq] += H[pq];

    free(density); free(V); free(W);
    return 0;
}

/* B = L^-1 T with L = chol(J2c) lower (DensityFitting.jl:137-152), explicit
 * loops: Cholesky-Banachiewicz, forward substitution per column of T.
 * J2c (Q,Q) column-major, lower triangle referenced; T (Q,P) column-major,
 * overwritten with B. */
int jcdf_oracle_form_B(int64_t Q, int64_t P, const double *J2c, double *T)
{
    double *L = (double *)calloc((size_t)(Q * Q), sizeof(double));
    if (!L) return 1;
    for (int64_t j = 0; j < Q; ++j) {
        double d = J2c[j + Q * j];
        for (int64_t k = 0; k < j; ++k) d -= L[j + Q * k] * L[j + Q * k];
        if (d <= 0.0) { free(L); return 2; }
        double ljj = __builtin_sqrt(d);
        L[j + Q * j] = ljj;
        for (int64_t i = j + 1; i < Q; ++i) {
            double s = J2c[i + Q * j];
            for (int64_t k = 0; k < j; ++k) s -= L[i + Q * k] * L[j + Q * k];
            L[i + Q * j] = s / ljj;
        }
    }
    /* L^-1 T == solve L X = T column by column */
    for (int64_t c = 0; c < P; ++c) {
        double *x = T + Q * c;
        for (int64_t i = 0; i < Q; ++i) {
            double s = x[i];
            for (int64_t k = 0; k < i; ++k) s -= L[i + Q * k] * x[k];
            x[i] = s / L[i + Q * i];
        }
    }
    free(L);
    return 0;
}
