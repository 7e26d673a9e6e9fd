// jcdf_host_lapack.hpp — dependency-free host Cholesky + triangular inverse for
// the DF metric: what LAPACK.potrf!('L') + LAPACK.trtri!('L','N') do at
// /root/reference/src/rhf/energy/DensityFitting/GPUDF.jl:890-891 and
// DensityFitting.jl:137-140.  Setup-time only (once per SCF), O(Q^3).
// Blocked, std::thread parallel; written from the textbook algorithms.
#pragma once
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <thread>
#include <vector>

namespace hostlapack {

constexpr int64_t NB = 64;

template <class F>
inline void parallel_for(int64_t n_items, F &&body)
{
    unsigned hw = std::thread::hardware_concurrency();
    int64_t nthr = std::max<int64_t>(1, std::min<int64_t>(hw ? hw : 4, n_items));
    if (nthr > 32) nthr = 32;
    if (nthr == 1) {
        for (int64_t i = 0; i < n_items; ++i) body(i);
        return;
    }
    std::atomic<int64_t> next(0);
    std::vector<std::thread> pool;
    pool.reserve((size_t)nthr);
    for (int64_t t = 0; t < nthr; ++t)
        pool.emplace_back([&]() {
            for (;;) {
                const int64_t i = next.fetch_add(1);
                if (i >= n_items) break;
                body(i);
            }
        });
    for (auto &th : pool) th.join();
}

// A (n x n, column-major, lower triangle referenced).  On success returns 0 and
// A holds L^-1 in its lower triangle with an exactly-zero upper triangle.
// On a non-positive pivot returns (1-based pivot index).
inline int potrf_trtri_lower(double *A, int64_t n)
{
    auto a = [&](int64_t i, int64_t j) -> double & { return A[i + n * j]; };

    // ---- blocked left-looking Cholesky -------------------------------------
    for (int64_t j0 = 0; j0 < n; j0 += NB) {
        const int64_t jb = std::min(NB, n - j0);
        const int64_t rows = n - j0;
        // panel update: A[j0:n, j0:j0+jb] -= A[j0:n, 0:j0] * A[j0:j0+jb, 0:j0]^T
        if (j0 > 0) {
            const int64_t RC = 128;
            parallel_for((rows + RC - 1) / RC, [&](int64_t chunk) {
                const int64_t r0 = j0 + chunk * RC, r1 = std::min(n, r0 + RC);
                for (int64_t k = 0; k < j0; ++k) {
                    const double *ak = &a(0, k);
                    for (int64_t c = 0; c < jb; ++c) {
                        const double s = ak[j0 + c];
                        double *col = &a(0, j0 + c);
                        for (int64_t r = r0; r < r1; ++r) col[r] -= ak[r] * s;
                    }
                }
            });
        }
        // unblocked factor of the diagonal block
        for (int64_t j = j0; j < j0 + jb; ++j) {
            double d = a(j, j);
            for (int64_t k = j0; k < j; ++k) d -= a(j, k) * a(j, k);
            if (!(d > 0.0)) return (int)(j + 1);
            const double ljj = std::sqrt(d);
            a(j, j) = ljj;
            for (int64_t i = j + 1; i < j0 + jb; ++i) {
                double s = a(i, j);
                for (int64_t k = j0; k < j; ++k) s -= a(i, k) * a(j, k);
                a(i, j) = s / ljj;
            }
        }
        // rows below the block: X * L_jj^T = A_panel  (forward substitution per row chunk)
        const int64_t below = n - (j0 + jb);
        if (below > 0) {
            const int64_t RC = 256;
            parallel_for((below + RC - 1) / RC, [&](int64_t chunk) {
                const int64_t r0 = j0 + jb + chunk * RC, r1 = std::min(n, r0 + RC);
                for (int64_t j = j0; j < j0 + jb; ++j) {
                    double *cj = &a(0, j);
                    for (int64_t k = j0; k < j; ++k) {
                        const double s = a(j, k);
                        const double *ck = &a(0, k);
                        for (int64_t r = r0; r < r1; ++r) cj[r] -= ck[r] * s;
                    }
                    const double inv = 1.0 / a(j, j);
                    for (int64_t r = r0; r < r1; ++r) cj[r] *= inv;
                }
            });
        }
    }

    // ---- X = L^-1, one column block per task (independent) ------------------
    std::vector<double> X((size_t)(n * n), 0.0);
    const int64_t nblk = (n + NB - 1) / NB;
    parallel_for(nblk, [&](int64_t bj) {
        const int64_t j0 = bj * NB, jb = std::min(NB, n - j0);
        std::vector<double> acc((size_t)(NB * NB));
        for (int64_t i0 = j0; i0 < n; i0 += NB) {
            const int64_t ib = std::min(NB, n - i0);
            // acc = E_block - L[i0:i0+ib, j0:i0] * X[j0:i0, j0:j0+jb]
            std::fill(acc.begin(), acc.end(), 0.0);
            if (i0 == j0)
                for (int64_t c = 0; c < jb; ++c) acc[(size_t)(c + NB * c)] = 1.0;
            for (int64_t k = j0; k < i0; ++k) {
                const double *lk = &a(i0, k);
                for (int64_t c = 0; c < jb; ++c) {
                    const double s = X[(size_t)(k + n * (j0 + c))];
                    if (s == 0.0) continue;
                    double *ac = &acc[(size_t)(NB * c)];
                    for (int64_t r = 0; r < ib; ++r) ac[r] -= lk[r] * s;
                }
            }
            // solve L[i0:i0+ib, i0:i0+ib] * Xblk = acc   (forward substitution)
            for (int64_t c = 0; c < jb; ++c) {
                double *ac = &acc[(size_t)(NB * c)];
                for (int64_t r = 0; r < ib; ++r) {
                    double s = ac[r];
                    for (int64_t k = 0; k < r; ++k) s -= a(i0 + r, i0 + k) * ac[k];
                    ac[r] = s / a(i0 + r, i0 + r);
                }
                for (int64_t r = 0; r < ib; ++r) X[(size_t)(i0 + r + n * (j0 + c))] = ac[r];
            }
        }
    });
    // upper triangle of X is exactly zero by construction (X starts at 0; entries
    // above the diagonal inside diagonal blocks are 0 - 0 / l)
    std::memcpy(A, X.data(), (size_t)(n * n) * sizeof(double));
    for (int64_t j = 1; j < n; ++j)
        for (int64_t i = 0; i < j; ++i) A[i + n * j] = 0.0;
    return 0;
}

}  // namespace hostlapack
