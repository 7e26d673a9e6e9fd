// jcdf_dc.hpp — divide & conquer eigensolver for a symmetric tridiagonal matrix on the device
// (caller side of the hot path, SURVEY 8 row f1: second stage of the replicated eigensolve of the
// SCF iteration, /root/reference/src/rhf/energy/SCF.jl:1080-1083, after k_sytrd_lower).
//
// Why: rocSOLVER's stedc needs 2.6 ms at N = 510, 1.7 ms of it in five launches of a merge kernel that
// solves the secular equations with little parallelism (profiles/r01_kernel_stats_bench.txt).  The
// algorithm here is the classical one (Cuppen; Gu & Eisenstat; LAPACK dstedc/dlaed1-4), organised for
// the GPU: the matrix is torn down to 1 x 1 leaves, and every level of the tree is a handful of launches
// over ALL merges of that level:
//   prepare : z = [last row of Q1, +-first row of Q2]/sqrt(2), rho = 2|beta|, merge-sort of the poles,
//             deflation (negligible z_i; close poles rotated together), one workgroup per merge
//   secular : every root of 1 + rho sum z_i^2/(d_i - x) by the "middle way" rational iteration inside a
//             shrinking bracket, relative to the nearer pole (so d_i - lambda_j is accurate), 8 lanes/root
//   zhat    : Loewner formula for the z that makes the computed roots exact (orthogonality of the vectors)
//   vectors : X[i][j] = zhat_i/(d_i - lambda_j), normalised; gather of the non-deflated columns of Q
//   update  : Q_new = Q_gathered X  (the O(n^3) part; fp64 MFMA for the large merges)
//   finish  : deflated columns copied, everything sorted ascending
// Eigenvectors are column-major (column j = eigenvector j), eigenvalues ascending, like LAPACK.
#pragma once
#include "jcdf_gemm.hpp"

namespace jcdf {

struct DcMerge {          // one merge of the tree: rows/columns [s, s + n1 + n2) of the matrix
    int s, n1, n2;
    int xoff;             // first row of this merge in the packed work matrices (multiple of 16)
};

constexpr double DC_EPS = 1.1102230246251565e-16;      // unit roundoff (LAPACK dlamch('E'))

// Wave-uniform reads of words that an earlier PHASE OF THE SAME KERNEL may have written (k_dc_merge_small runs all
// phases of a merge in one workgroup): forced onto the vector path — a scalar load could be served from a scalar-
// cache line that another workgroup pulled in before the word was written.
__device__ __forceinline__ int dc_ld(const int *p)
{
    return __hip_atomic_load(const_cast<int *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ double dc_ld(const double *p)
{
    return __hip_atomic_load(const_cast<double *>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// ---- scaling (LAPACK dstedc scales to unit max-norm: the tolerances and products below assume O(1) data) --
// sc[0] = max(|d|, |e|) (1 for the zero matrix), sc[1] = 1/sc[0].  One workgroup.
__global__ __launch_bounds__(256) void k_dc_norm(const double *__restrict__ d, const double *__restrict__ e, int n,
                                                 double *__restrict__ sc, int *__restrict__ info)
{
    __shared__ double red[256];
    __shared__ int bad;
    if (threadIdx.x == 0) bad = 0;
    __syncthreads();
    double mx = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        const double di = d[i], ei = i < n - 1 ? e[i] : 0.0;
        if (!(fabs(di) <= 1.79e308) || !(fabs(ei) <= 1.79e308)) bad = 1;        // NaN / Inf (fmax below would skip a NaN)
        mx = fmax(mx, fmax(fabs(di), fabs(ei)));
    }
    red[threadIdx.x] = mx;
    __syncthreads();
    // the call's info word starts here (no separate memset launch): 2 = non-finite input, every later kernel of this call returns
    // at once (k_dc_leaf included); 0 otherwise, until a leaf reports its iteration cap (1)
    if (threadIdx.x == 0) *info = bad ? 2 : 0;
    for (int off = 128; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) red[threadIdx.x] = fmax(red[threadIdx.x], red[threadIdx.x + off]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double nrm = (red[0] > 0.0 && red[0] < 1.0e300) ? red[0] : 1.0;
        sc[0] = nrm;
        sc[1] = 1.0 / nrm;
    }
}

// ---- level 0: tear the matrix into leaves of DC_LEAF rows and solve them directly ---------------------------------
// Both eigenvector buffers are zeroed: every level writes only the diagonal blocks of its merges and the next level
// relies on the blocks between two children being zero.
__global__ void k_dc_init(int n, double *__restrict__ Z, int64_t ldz, double *__restrict__ Z2, int64_t ldz2)
{
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx < (int64_t)n * n) {
        Z[(idx / n) * ldz + idx % n] = 0.0;
        Z2[(idx / n) * ldz2 + idx % n] = 0.0;
    }
}

// Leaf s .. s + m - 1 (m <= DC_LEAF) of the torn matrix: diagonal d - |e| at a torn end (every block boundary is a tear,
// rho = 2 |e| in the merge that mends it), off-diagonals as they are, all scaled by sc[1].  Implicit QL with Wilkinson
// shifts (EISPACK tql2 / LAPACK dsteqr's algorithm) — three levels of 2-, 4- and 8-row merges were three launches of
// ~25 us of pure latency; this one takes 31 us at n = 510 (64 leaves).  One leaf per workgroup of eight lanes (leaves sharing a wave would wait for each other's
// data-dependent loops): every lane runs the same scalar recurrence on its own copy of d and e (LDS used as indexable
// private memory, element-major so that the lanes sit on different banks) and lane r rotates row r of the eigenvector
// matrix, so nothing is exchanged between lanes and no barrier is needed.  The rotations take 1 / sqrt(f^2 + g^2) from the
// hardware estimate + two Newton steps (full precision) instead of a square root and two divisions: the recurrence is one
// dependent chain, and the IEEE expansions and the LDS round trips inside it were most of its time.
// Out: w[s + k] ascending, row s + k of Z = eigenvector k (components in columns s ..).
constexpr int DC_LEAF = 8;
__global__ __launch_bounds__(64) void k_dc_leaf(const double *__restrict__ d, const double *__restrict__ e, int n, double *__restrict__ w,
                                                double *__restrict__ Z, int64_t ldz, const double *__restrict__ sc, int last, int *__restrict__ info)
{
    __shared__ double sd[DC_LEAF][DC_LEAF], se[DC_LEAF][DC_LEAF], sz[DC_LEAF][DC_LEAF];
    const int lane = threadIdx.x, r = lane;
    const int s = blockIdx.x * DC_LEAF;
    const int m = n - s < DC_LEAF ? n - s : DC_LEAF;
    if (m <= 0 || lane >= DC_LEAF || *info == 2) return;
    const double scale = sc[1];
    for (int i = 0; i < DC_LEAF; ++i) {
        double di = 0.0, ei = 0.0;
        if (i < m) {
            di = d[s + i];
            if (i == 0 && s > 0) di -= fabs(e[s - 1]);
            if (i == m - 1 && s + m < n) di -= fabs(e[s + m - 1]);
            di *= scale;
            if (i < m - 1) ei = e[s + i] * scale;
        }
        sd[i][lane] = di;
        se[i][lane] = ei;
        sz[i][lane] = (i == r) ? 1.0 : 0.0;               // row r of the unit matrix
    }
    const double eps = 2.0 * DC_EPS;
    for (int l = 0; l < m; ++l) {
        int iter = 0, mm;
        do {
            // first negligible off-diagonal at or after l: lane j tests position j of its own (identical) copy, one ballot
            {
                const bool small = lane >= l && lane < m - 1 &&
                                   fabs(se[lane][lane]) <= eps * (fabs(sd[lane][lane]) + fabs(sd[lane < DC_LEAF - 1 ? lane + 1 : lane][lane]));
                const unsigned long long hit = __ballot(small);
                mm = hit ? (int)__builtin_ctzll(hit) : m - 1;
            }
            if (mm != l) {
                if (iter++ == 60) {                       // never seen on finite input; NaN input ends here (`small` is never true):
                    if (lane == 0) *info = 1;             // reported — the caller must not use this decomposition (jcdf.h: stedc info word)
                    break;
                }
                const double el = se[l][lane], dl = sd[l][lane];
                // Wilkinson shift (reciprocals and the root by estimate + two Newton steps, as in the rotations below)
                auto recip = [](double x) { double y = __builtin_amdgcn_rcp(x); y = y * (2.0 - x * y); return y * (2.0 - x * y); };
                double g = (sd[l + 1][lane] - dl) * recip(2.0 * el);
                const double g21 = g * g + 1.0;
                double rq = __builtin_amdgcn_rsq(g21);
                rq = rq * (1.5 - 0.5 * g21 * rq * rq);
                rq = rq * (1.5 - 0.5 * g21 * rq * rq);
                double rr = g21 * rq;
                g = sd[mm][lane] - dl + el * recip(g + copysign(rr, g));
                double sn = 1.0, cs = 1.0, p = 0.0;
                // the chase, i = mm-1 .. l.  What a rotation reads was either produced by the one before it (carried in
                // registers: d_c = d[i+1], z_c = z[i+1]) or is untouched so far and fetched one rotation ahead
                // (e_i, d_i, z_i): the dependent chain is arithmetic only.
                double e_i = se[mm - 1][lane], d_i = sd[mm - 1][lane], z_i = sz[mm - 1][lane];
                double d_c = sd[mm][lane], z_c = sz[mm][lane];
                int i;
                for (i = mm - 1; i >= l; --i) {
                    const int ip = i > 0 ? i - 1 : 0;
                    const double e_n = se[ip][lane], d_n = sd[ip][lane], z_n = sz[ip][lane];
                    const double f = sn * e_i, b = cs * e_i;
                    const double h2 = f * f + g * g;
                    double ri = __builtin_amdgcn_rsq(h2);                     // (h2 == 0: inf, not used)
                    ri = ri * (1.5 - 0.5 * h2 * ri * ri);
                    ri = ri * (1.5 - 0.5 * h2 * ri * ri);
                    rr = (h2 == 0.0) ? 0.0 : h2 * ri;
                    se[i + 1][lane] = rr;
                    if (rr == 0.0) {
                        sd[i + 1][lane] = d_c - p;
                        se[mm][lane] = 0.0;
                        break;
                    }
                    sn = f * ri;
                    cs = g * ri;
                    g = d_c - p;
                    rr = (d_i - g) * sn + 2.0 * cs * b;
                    p = sn * rr;
                    sd[i + 1][lane] = g + p;
                    g = cs * rr - b;
                    sz[i + 1][lane] = sn * z_i + cs * z_c;
                    z_c = cs * z_i - sn * z_c;
                    sz[i][lane] = z_c;
                    d_c = d_i;
                    e_i = e_n;
                    d_i = d_n;
                    z_i = z_n;
                }
                if (rr == 0.0 && i >= l) continue;
                sd[l][lane] = d_c - p;
                se[l][lane] = g;
                se[mm][lane] = 0.0;
            }
        } while (mm != l);
    }
    // ascending order (ties: lower index first), eigenvector k = column k of the rotated unit matrix: lane r holds its component r
    const double unscale = last ? sc[0] : 1.0;
    for (int k = 0; k < m; ++k) {
        const double dk = sd[k][lane];
        int rank = 0;
        for (int j = 0; j < m; ++j) {
            const double dj = sd[j][lane];
            rank += (dj < dk || (dj == dk && j < k)) ? 1 : 0;
        }
        if (r == 0) w[s + rank] = dk * unscale;
        if (r < m) Z[(int64_t)(s + rank) * ldz + s + r] = sz[k][lane];
    }
}


// ---- prepare: one workgroup per merge ---------------------------------------------------------------------
// In : w (eigenvalues of the two children, each ascending), Z (their eigenvectors, block diagonal), e.
// Out (packed at xoff): K (non-deflated count), dl[K] ascending poles, zl[K], col[K] (column of Z, local
//      index, of pole k), defcol[m-K]/defval[m-K] (deflated eigenpairs), rho.  Givens rotations of the
//      close-pole deflation are applied to the columns of Z in place.
// LDS: (2*m + max(m, 256)) doubles + 3*m ints.
__device__ __forceinline__ void dc_prepare_body(const DcMerge *__restrict__ merges, const double *__restrict__ w,
                                                    const double *__restrict__ e, double *__restrict__ Z, int64_t ldz,
                                                    int *__restrict__ Kout, double *__restrict__ rho_out,
                                                    double *__restrict__ dl, double *__restrict__ zl, int *__restrict__ col,
                                                    int *__restrict__ defcol, double *__restrict__ defval,
                                                    const double *__restrict__ sc, int mi, int vbx, int vgx, double *dyn_smem)
{
    double *sm = dyn_smem;
    const DcMerge mg = merges[mi];
    const int s = mg.s, n1 = mg.n1, m = mg.n1 + mg.n2, tid = threadIdx.x, nthr = blockDim.x;
    double *dd = sm, *zz = sm + m, *red = sm + 2 * m;                        // by local column index
    int *perm = reinterpret_cast<int *>(sm + 2 * m + (m > 256 ? m : 256));   // sorted position -> local index
    int *klist = perm + m;
    const double beta = e[s + n1 - 1] * sc[1];
    const double sgn = beta < 0.0 ? -1.0 : 1.0;
    const double rho = 2.0 * fabs(beta);
    double lmax = 0.0;
    for (int i = tid; i < m; i += nthr) {
        const double di = w[s + i];
        const double zi = (i < n1 ? Z[(int64_t)(s + i) * ldz + (s + n1 - 1)] : sgn * Z[(int64_t)(s + i) * ldz + (s + n1)]) * 0.70710678118654752440;
        dd[i] = di;
        zz[i] = zi;
        lmax = fmax(lmax, fmax(fabs(di), fabs(zi)));
    }
    red[tid] = lmax;
    __syncthreads();
    for (int off = nthr / 2; off > 0; off >>= 1) {
        if (tid < off) red[tid] = fmax(red[tid], red[tid + off]);
        __syncthreads();
    }
    const double tol = 8.0 * DC_EPS * red[0];
    __syncthreads();
    // merge of the two ascending runs: rank by binary search in the other run (ties: first run first)
    for (int i = tid; i < m; i += nthr) {
        const double x = dd[i];
        int lo, hi;
        if (i < n1) { lo = n1; hi = m; while (lo < hi) { const int mid = (lo + hi) >> 1; if (dd[mid] < x) lo = mid + 1; else hi = mid; } perm[i + (lo - n1)] = i; }
        else        { lo = 0; hi = n1; while (lo < hi) { const int mid = (lo + hi) >> 1; if (dd[mid] <= x) lo = mid + 1; else hi = mid; } perm[(i - n1) + lo] = i; }
    }
    __syncthreads();
    // Deflation (LAPACK dlaed2).  Common case first, in parallel: poles with negligible z are deflated, and if no
    // two neighbouring survivors are close enough to be rotated together (that test needs only the ORIGINAL
    // values as long as no rotation has happened), the survivors are simply compacted in ascending order.
    int K = 0, ndef = 0;
    const int xo = mg.xoff;
    int *flag = klist;                                                        // 1 = survivor (by sorted position), then scanned
    int *scan = reinterpret_cast<int *>(red);                                 // inclusive prefix sums (m ints <= m doubles)
    for (int t = tid; t < m; t += nthr) flag[t] = (rho * fabs(zz[perm[t]]) <= tol) ? 0 : 1;
    __syncthreads();
    for (int t = tid; t < m; t += nthr) scan[t] = flag[t];
    __syncthreads();
    for (int off = 1; off < m; off <<= 1) {                                   // Hillis-Steele inclusive scan, m <= 8 * nthr
        int v[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = tid + q * nthr;
            v[q] = (t < m && t >= off) ? scan[t - off] : 0;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int t = tid + q * nthr;
            if (t < m) scan[t] += v[q];
        }
        __syncthreads();
    }
    // survivors in order: position scan[t]-1; a survivor's predecessor is the survivor with index scan[t]-2
    int *surv = perm + 2 * m;                                                 // compacted survivors (local column index)
    for (int t = tid; t < m; t += nthr)
        if (flag[t]) surv[scan[t] - 1] = perm[t];
    __syncthreads();
    K = scan[m - 1];
    int close = 0;
    for (int k = 1 + tid; k < K; k += nthr) {
        const int pj = surv[k - 1], j = surv[k];
        const double zj = zz[j], zp = zz[pj];
        // |(d_j - d_pj) c s| <= tol with c s = -z_j z_pj / (z_j^2 + z_pj^2)
        if (fabs((dd[j] - dd[pj]) * zj * zp) <= tol * (zj * zj + zp * zp)) close = 1;
    }
    close = __syncthreads_or(close);
    if (!close) {
        ndef = m - K;
        for (int t = tid; t < m; t += nthr)
            if (!flag[t]) {
                const int q = t - scan[t];                                    // deflated so far, in ascending order
                defcol[xo + q] = perm[t];
                defval[xo + q] = dd[perm[t]];
            }
        __syncthreads();                                                      // flag (== klist) has been read
        for (int k = tid; k < K; k += nthr) klist[k] = surv[k];
    } else {
    // Rare case (close poles: degenerate or clustered spectra): the sequential scan in ascending order, every thread
    // runs the scalar logic on the same LDS values; the rotations are applied to the two columns of Z by the whole
    // workgroup.
    K = 0;
    int pj = -1;
    for (int t = 0; t < m; ++t) {
        const int j = perm[t];
        const double zj = zz[j];
        if (rho * fabs(zj) <= tol) {                                         // eigenpair (d_j, e_j) unchanged
            if (tid == 0) { defcol[xo + ndef] = j; defval[xo + ndef] = dd[j]; }
            ++ndef;
            continue;
        }
        if (pj < 0) { pj = j; continue; }
        const double zp = zz[pj], dp = dd[pj], dj = dd[j];
        if (fabs((dj - dp) * zj * zp) <= tol * (zj * zj + zp * zp)) {         // poles too close: rotate z_pj into z_j
            const double tau = hypot(zj, zp);
            const double c = zj / tau, sn = -zp / tau;
            __syncthreads();
            if (tid == 0) {
                zz[j] = tau;
                zz[pj] = 0.0;
                const double t1 = dp * c * c + dj * sn * sn;
                dd[j] = dp * sn * sn + dj * c * c;
                dd[pj] = t1;
                defcol[xo + ndef] = pj;
                defval[xo + ndef] = t1;
            }
            double *cp = Z + (int64_t)(s + pj) * ldz + s, *cj = Z + (int64_t)(s + j) * ldz + s;
            for (int r = tid; r < m; r += nthr) {                             // drot(Q(:,pj), Q(:,j), c, s)
                const double a = cp[r], b = cj[r];
                cp[r] = c * a + sn * b;
                cj[r] = c * b - sn * a;
            }
            __syncthreads();
            ++ndef;
            pj = j;
        } else {
            if (tid == 0) klist[K] = pj;
            ++K;
            pj = j;
        }
    }
    if (pj >= 0) {
        if (tid == 0) klist[K] = pj;
        ++K;
    }
    }
    __syncthreads();
    for (int k = tid; k < K; k += nthr) {
        const int c = klist[k];
        dl[xo + k] = dd[c];
        zl[xo + k] = zz[c];
        col[xo + k] = c;
    }
    if (tid == 0) {
        Kout[mi] = K;
        rho_out[mi] = rho;
    }
}

__global__ __launch_bounds__(256) void k_dc_prepare(const DcMerge *__restrict__ merges, const double *__restrict__ w,
                                                    const double *__restrict__ e, double *__restrict__ Z, int64_t ldz,
                                                    int *__restrict__ Kout, double *__restrict__ rho_out,
                                                    double *__restrict__ dl, double *__restrict__ zl, int *__restrict__ col,
                                                    int *__restrict__ defcol, double *__restrict__ defval,
                                                    const double *__restrict__ sc, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_prepare_body(merges, w, e, Z, ldz, Kout, rho_out, dl, zl, col, defcol, defval, sc, (int)blockIdx.x, 0, 1, dyn_smem);
}

// ---- secular equation ------------------------------------------------------------------------------------
// Root j of f(x) = 1 + rho sum_i z_i^2/(d_i - x) in (d_j, d_{j+1}) (last: (d_K, d_K + rho |z|^2]),
// returned as (origin o, mu) with lambda = d_o + mu, o the nearer pole.  LANES lanes share the sums.
// (on the VALU by DPP where the group is a quad, a row or the wave — jcdf_eig.hpp; a double __shfl_xor is two trips through
//  the LDS crossbar, and an evaluation of the secular function makes 24 of them: with the division below, most of its time)
template <int LANES>
__device__ __forceinline__ double lanes_sum(double x)
{
    if constexpr (LANES == 64) return wave_sum(x);
    else if constexpr (LANES == 16) return row16_sum(x);
    else if constexpr (LANES == 4) {
        x += dpp_f64<0xB1, 0xf>(x);      // quad_perm [1,0,3,2]
        x += dpp_f64<0x4E, 0xf>(x);      // quad_perm [2,3,0,1]
        return x;
    } else {
#pragma unroll
        for (int off = LANES / 2; off > 0; off >>= 1) x += __shfl_xor(x, off, LANES);
        return x;
    }
}

// 1 / x by the hardware estimate + two Newton steps (full double precision, not correctly rounded): the IEEE division is a
// chain of ~25 dependent instructions, and the sums below are one division per pole
__device__ __forceinline__ double dc_recip(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    return y * (2.0 - x * y);
}

template <int LANES>
__device__ void dc_secular_root(int j, int K, const double *__restrict__ d, const double *__restrict__ z, double rho,
                                int lane, int *org_out, double *mu_out)
{
    // sums over the poles split at j: psi (poles <= j), phi (poles > j), and their derivatives
    auto eval = [&](int o, double mu, double &psi, double &dpsi, double &phi, double &dphi) {
        double a = 0.0, da = 0.0, b = 0.0, db = 0.0;
        const double dorg = d[o];
        for (int i0 = lane; i0 < K; i0 += 4 * LANES) {                       // four independent terms at a time (each is a chain of ~10)
            double t[4], zi[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int i = i0 + u * LANES;
                const bool in = i < K;
                zi[u] = in ? z[i] : 0.0;
                t[u] = zi[u] * dc_recip(in ? (d[i] - dorg) - mu : 1.0);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const double zt = zi[u] * t[u], tt = t[u] * t[u];
                if (i0 + u * LANES <= j) { a += zt; da += tt; } else { b += zt; db += tt; }
            }
        }
        psi = rho * lanes_sum<LANES>(a);
        dpsi = rho * lanes_sum<LANES>(da);
        phi = rho * lanes_sum<LANES>(b);
        dphi = rho * lanes_sum<LANES>(db);
    };
    int o;
    double lo, hi, mu;                                                        // bracket for mu: f(lo) < 0 < f(hi) (or a pole)
    double psi, dpsi, phi, dphi;
    if (j < K - 1) {                                                          // the sign of f at mid-gap picks the nearer pole;
        const double gap = d[j + 1] - d[j];                                   // the same evaluation is the first iterate
        eval(j, 0.5 * gap, psi, dpsi, phi, dphi);
        if (1.0 + psi + phi >= 0.0) { o = j; lo = 0.0; hi = 0.5 * gap; mu = hi; }
        else { o = j + 1; lo = -0.5 * gap; hi = 0.0; mu = lo; }
    } else {
        double zz = 0.0;
        for (int i = lane; i < K; i += LANES) zz += z[i] * z[i];
        zz = lanes_sum<LANES>(zz);
        o = j; lo = 0.0; hi = rho * zz;
        if (!(hi > 0.0)) hi = DC_EPS * fabs(d[j]) + 1e-300;
        mu = 0.5 * hi;
        eval(o, mu, psi, dpsi, phi, dphi);
    }
    const double dj = d[j] - d[o], dj1 = (j < K - 1) ? d[j + 1] - d[o] : 0.0;
    for (int it = 0; it < 100; ++it) {
        if (it > 0) eval(o, mu, psi, dpsi, phi, dphi);
        const double f = 1.0 + psi + phi;
        const double err = 8.0 * (fabs(phi) + fabs(psi)) + 1.0 + fabs(mu) * (dpsi + dphi);   // dlaed4's erretm
        if (fabs(f) <= DC_EPS * err) break;
        if (f < 0.0) lo = mu; else hi = mu;
        if (hi - lo <= 2.0 * DC_EPS * fmax(fabs(lo), fabs(hi))) { mu = 0.5 * (lo + hi); break; }
        // rational model: psi ~ s + a/(D1 - eta) (pole j), phi ~ t + b/(D2 - eta) (pole j+1), eta = step in mu
        const double D1 = dj - mu, D2 = dj1 - mu;
        double next;
        if (j < K - 1) {
            const double a = dpsi * D1 * D1, b = dphi * D2 * D2;
            const double c0 = 1.0 + (psi - dpsi * D1) + (phi - dphi * D2);
            // c0 (D1-eta)(D2-eta) + a (D2-eta) + b (D1-eta) = 0
            const double qa = c0, qb = -(c0 * (D1 + D2) + a + b), qc = c0 * D1 * D2 + a * D2 + b * D1;
            double eta;
            if (qa == 0.0) eta = -qc / qb;
            else {
                const double disc = qb * qb - 4.0 * qa * qc;
                const double sq = sqrt(fmax(disc, 0.0));
                // the root between the two poles (D1 < eta < D2 in the shifted variable): stable form
                const double q = -0.5 * (qb + (qb >= 0.0 ? sq : -sq));
                const double r1 = q / qa, r2 = (q != 0.0) ? qc / q : r1;
                eta = (r1 > D1 && r1 < D2) ? r1 : r2;
            }
            next = mu + eta;
        } else {
            const double a = dpsi * D1 * D1, c0 = 1.0 + (psi - dpsi * D1);   // single pole below
            next = mu + (D1 + a / c0);
        }
        if (!(next > lo && next < hi)) next = 0.5 * (lo + hi);               // also catches NaN
        mu = next;
    }
    if (lane == 0) {
        *org_out = o;
        *mu_out = mu;
    }
}

// grid: (blocks, merges); LANES lanes per root
template <int LANES>
__device__ __forceinline__ void dc_secular_body(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                    const double *__restrict__ rho_in, const double *__restrict__ dl,
                                                    const double *__restrict__ zl, int *__restrict__ org,
                                                    double *__restrict__ mu, int mi, int vbx, int vgx, double *dyn_smem)
{
    const DcMerge mg = merges[mi];
    const int K = dc_ld(Kin + mi);
    const int lane = threadIdx.x % LANES;
    const int rpb = blockDim.x / LANES;                                       // roots per block
    // the poles and weights of the merge go to LDS once: a root needs ~10 evaluations of the secular function, each a pass over
    // all of them, and from global memory every pass paid the L2 latency per term (19 us per launch, ~2 us per evaluation)
    double *sd = dyn_smem, *sz = dyn_smem + K;                                // (2 K doubles <= 2 m: see the launches)
    for (int i = threadIdx.x; i < K; i += blockDim.x) {
        sd[i] = dl[mg.xoff + i];
        sz[i] = zl[mg.xoff + i];
    }
    __syncthreads();
    for (int j = vbx * rpb + threadIdx.x / LANES; j < ((K + rpb - 1) / rpb) * rpb; j += vgx * rpb) {
        // (whole groups stay in the loop together: the reductions need every lane of the group)
        if (j < K) dc_secular_root<LANES>(j, K, sd, sz, dc_ld(rho_in + mi), lane, org + mg.xoff + j, mu + mg.xoff + j);
    }
}

template <int LANES>
__global__ __launch_bounds__(256) void k_dc_secular(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                    const double *__restrict__ rho_in, const double *__restrict__ dl,
                                                    const double *__restrict__ zl, int *__restrict__ org,
                                                    double *__restrict__ mu, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_secular_body<LANES>(merges, Kin, rho_in, dl, zl, org, mu, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, dyn_smem);
}

// ---- zhat (Loewner / Gu-Eisenstat): the z for which the computed roots are the exact eigenvalues ----------
// zhat_i = sign(z_i) sqrt( (lambda_i - d_i) prod_{j != i} (lambda_j - d_i)/(d_j - d_i) ),
// with d_i - lambda_j = (d_i - d_{o_j}) - mu_j.  LANES lanes per i.
template <int LANES>
__device__ __forceinline__ void dc_zhat_body(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                 const double *__restrict__ dl, const double *__restrict__ zl,
                                                 const int *__restrict__ org, const double *__restrict__ mu,
                                                 double *__restrict__ zhat, int mi, int vbx, int vgx, double *dyn_smem)
{
    const DcMerge mg = merges[mi];
    const int K = dc_ld(Kin + mi);
    const double *d = dl + mg.xoff, *m_ = mu + mg.xoff;
    const int *o = org + mg.xoff;
    const int lane = threadIdx.x % LANES, rpb = blockDim.x / LANES;
    for (int i = vbx * rpb + threadIdx.x / LANES; i < ((K + rpb - 1) / rpb) * rpb; i += vgx * rpb) {
        double p = 1.0;
        if (i < K) {
            const double di = d[i];
            for (int j = lane; j < K; j += LANES) {
                const double num = m_[j] - (di - d[o[j]]);                    // lambda_j - d_i
                p *= (j == i) ? num : num / (d[j] - di);
            }
        }
#pragma unroll
        for (int off = LANES / 2; off > 0; off >>= 1) p *= __shfl_xor(p, off, LANES);
        if (i < K && lane == 0) zhat[mg.xoff + i] = copysign(sqrt(fabs(p)), zl[mg.xoff + i]);
    }
}

template <int LANES>
__global__ __launch_bounds__(256) void k_dc_zhat(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                 const double *__restrict__ dl, const double *__restrict__ zl,
                                                 const int *__restrict__ org, const double *__restrict__ mu,
                                                 double *__restrict__ zhat, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_zhat_body<LANES>(merges, Kin, dl, zl, org, mu, zhat, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, dyn_smem);
}

// ---- vectors: X[i][j] = zhat_i/(d_i - lambda_j) (row-major K x K, leading dimension ldx; rows and columns
//      K .. roundup(K,16)-1 zeroed when the MFMA update follows (pad16); the columns are normalised later, on the rows of the
//      product, in k_dc_finish) and the gather Zp[k][r] = Z[r][col_k] (k-major copy of the non-deflated
//      columns, rows K.. zeroed).  Purely elementwise: grid (blocks, merges), grid-stride ----------------------
__device__ __forceinline__ void dc_vectors_body(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                    const double *__restrict__ dl, const int *__restrict__ org,
                                                    const double *__restrict__ mu, const double *__restrict__ zhat,
                                                    const int *__restrict__ col, const double *__restrict__ Z, int64_t ldz,
                                                    double *__restrict__ X, double *__restrict__ Zp, int64_t ldx, int pad16, int mi, int vbx, int vgx, double *dyn_smem)
{
    const DcMerge mg = merges[mi];
    const int K = dc_ld(Kin + mi), m = mg.n1 + mg.n2, s = mg.s;
    const int Kp = pad16 ? (K + 15) / 16 * 16 : K;       // the MFMA update wants whole 16-row k stages
    const double *d = dl + mg.xoff, *zh = zhat + mg.xoff, *m_ = mu + mg.xoff;
    const int *o = org + mg.xoff;
    double *Xm = X + (int64_t)mg.xoff * ldx, *Zm = Zp + (int64_t)mg.xoff * ldx;
    const int64_t nx = (int64_t)Kp * Kp, total = nx + (int64_t)Kp * m;
    for (int64_t idx = (int64_t)vbx * blockDim.x + threadIdx.x; idx < total; idx += (int64_t)vgx * blockDim.x) {
        if (idx < nx) {
            const int i = (int)(idx / Kp), j = (int)(idx % Kp);
            Xm[(int64_t)i * ldx + j] = (i < K && j < K) ? zh[i] / ((d[i] - d[o[j]]) - m_[j]) : 0.0;
        } else {
            const int k = (int)((idx - nx) / m), r = (int)((idx - nx) % m);
            Zm[(int64_t)k * ldx + r] = (k < K) ? Z[(int64_t)(s + col[mg.xoff + k]) * ldz + s + r] : 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void k_dc_vectors(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                    const double *__restrict__ dl, const int *__restrict__ org,
                                                    const double *__restrict__ mu, const double *__restrict__ zhat,
                                                    const int *__restrict__ col, const double *__restrict__ Z, int64_t ldz,
                                                    double *__restrict__ X, double *__restrict__ Zp, int64_t ldx, int pad16, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_vectors_body(merges, Kin, dl, org, mu, zhat, col, Z, ldz, X, Zp, ldx, pad16, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, dyn_smem);
}

// ---- update: G[j][r] = sum_k X[k][j] * Zp[k][r]  (k-major operands; G row j = new eigenvector j over rows r) -
// small/unaligned version: one thread per (j, r), 16 x 16 tiles through LDS.  grid: (tiles, merges)
__device__ __forceinline__ void dc_update_simple_body(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                          const double *__restrict__ X, const double *__restrict__ Zp,
                                                          int64_t ldx, double *__restrict__ Gm, int mi, int vbx, int vgx, double *dyn_smem)
{
    __shared__ double xs[16][17], zs[16][17];
    const DcMerge mg = merges[mi];
    const int K = dc_ld(Kin + mi), m = mg.n1 + mg.n2;
    const int tj = (K + 15) / 16, tr = (m + 15) / 16;
    const double *Xm = X + (int64_t)mg.xoff * ldx, *Zm = Zp + (int64_t)mg.xoff * ldx;
    double *G = Gm + (int64_t)mg.xoff * ldx;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    for (int tile = vbx; tile < tj * tr; tile += vgx) {
        const int j0 = (tile / tr) * 16, r0 = (tile % tr) * 16;
        double acc = 0.0;
        for (int k0 = 0; k0 < K; k0 += 16) {
            xs[ty][tx] = (k0 + ty < K && j0 + tx < K) ? Xm[(int64_t)(k0 + ty) * ldx + j0 + tx] : 0.0;
            zs[ty][tx] = (k0 + ty < K && r0 + tx < m) ? Zm[(int64_t)(k0 + ty) * ldx + r0 + tx] : 0.0;
            __syncthreads();
#pragma unroll
            for (int k = 0; k < 16; ++k) acc += xs[k][ty] * zs[k][tx];
            __syncthreads();
        }
        if (j0 + ty < K && r0 + tx < m) G[(int64_t)(j0 + ty) * ldx + r0 + tx] = acc;
    }
}

__global__ __launch_bounds__(256) void k_dc_update_simple(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                          const double *__restrict__ X, const double *__restrict__ Zp,
                                                          int64_t ldx, double *__restrict__ Gm, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_update_simple_body(merges, Kin, X, Zp, ldx, Gm, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, dyn_smem);
}

// MFMA version for the large merges: T x T output tile per workgroup, T = 64 or 32.  grid: (tiles, merges).  The levels
// near the top have one or two merges: with 64 x 64 tiles a 510-row merge is 64 workgroups on a 256-CU chip, each walking
// the whole contraction alone (41-49 us); 32 x 32 tiles spread the same product over 256 (k_stedc picks the tile).
using DcCfg = GemmCfg<2, 2, 2, 2, 16>;
using DcCfg32 = GemmCfg<1, 1, 2, 2, 16>;
template <class Cfg>
__global__ __launch_bounds__(256) void k_dc_update_mfma(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                        const double *__restrict__ X, const double *__restrict__ Zp,
                                                        int64_t ldx, double *__restrict__ Gm, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    constexpr int T = Cfg::TM;
    static_assert(Cfg::TM == Cfg::TN, "square tiles");
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const DcMerge mg = merges[blockIdx.y];
    const int K = Kin[blockIdx.y], m = mg.n1 + mg.n2;
    const int tj = (K + T - 1) / T, tr = (m + T - 1) / T;
    const double *Xm = X + (int64_t)mg.xoff * ldx, *Zm = Zp + (int64_t)mg.xoff * ldx;
    double *G = Gm + (int64_t)mg.xoff * ldx;
    for (int tile = blockIdx.x; tile < tj * tr; tile += gridDim.x) {
        const int j0 = (tile / tr) * T, r0 = (tile % tr) * T;
        double4_t acc[Cfg::WM][Cfg::WN];
#pragma unroll
        for (int a = 0; a < Cfg::WM; ++a)
#pragma unroll
            for (int b = 0; b < Cfg::WN; ++b) acc[a][b] = double4_t{0.0, 0.0, 0.0, 0.0};
        gemm_tn_core<Cfg, false>(Xm + j0, ldx, Zm + r0, ldx, (K + 15) / 16, acc, smem);
#pragma unroll
        for (int a = 0; a < Cfg::WM; ++a)
#pragma unroll
            for (int b = 0; b < Cfg::WN; ++b)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int jj = j0 + tile_row<Cfg>(a, q), rr = r0 + tile_col<Cfg>(b);
                    if (jj < K && rr < m) G[(int64_t)jj * ldx + rr] = acc[a][b][q];
                }
        __syncthreads();
    }
}

// ---- finish: the m eigenpairs of the merged block (K new ones from G, m-K deflated ones from Z) sorted
//      ascending into Znew / wnew.  grid: (blocks, merges); each block ranks and copies a slice of the pairs --
__device__ __forceinline__ void dc_finish_body(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                   const double *__restrict__ dl, const int *__restrict__ org,
                                                   const double *__restrict__ mu, const int *__restrict__ defcol,
                                                   const double *__restrict__ defval, const double *__restrict__ Gm,
                                                   int64_t ldx, const double *__restrict__ Z, int64_t ldz_in,
                                                   double *__restrict__ Znew, int64_t ldz, double *__restrict__ wnew,
                                                   const double *__restrict__ sc, int last, int mi, int vbx, int vgx, double *dyn_smem)
{
    double *vals = dyn_smem;                                                  // m values: new roots then deflated
    const DcMerge mg = merges[mi];
    const int K = dc_ld(Kin + mi), m = mg.n1 + mg.n2, s = mg.s, xo = mg.xoff;
    for (int a = threadIdx.x; a < m; a += blockDim.x)
        vals[a] = (a < K) ? dl[xo + org[xo + a]] + mu[xo + a] : defval[xo + a - K];
    __syncthreads();
    const int per = (m + vgx - 1) / vgx;
    const int a0 = vbx * per, a1 = min(m, a0 + per);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int a = a0 + wave; a < a1; a += nw) {                                // one wave per eigenpair
        const double x = vals[a];
        int rank = 0;
        for (int b = lane; b < m; b += 64) rank += (vals[b] < x || (vals[b] == x && b < a)) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) rank += __shfl_xor(rank, off, 64);
        const double *src = (a < K) ? Gm + (int64_t)(xo + a) * ldx : Z + (int64_t)(s + dc_ld(defcol + xo + a - K)) * ldz_in + s;
        double *dst = Znew + (int64_t)(s + rank) * ldz + s;
        double scale = 1.0;
        if (a < K) {                                                          // new vector: normalise (X was left unscaled)
            double nrm = 0.0;
            for (int r = lane; r < m; r += 64) nrm += src[r] * src[r];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) nrm += __shfl_xor(nrm, off, 64);
            scale = 1.0 / sqrt(nrm);
        }
        for (int r = lane; r < m; r += 64) dst[r] = src[r] * scale;
        if (lane == 0) wnew[s + rank] = last ? x * sc[0] : x;                 // the last level undoes the scaling
    }
}

__global__ __launch_bounds__(256) void k_dc_finish(const DcMerge *__restrict__ merges, const int *__restrict__ Kin,
                                                   const double *__restrict__ dl, const int *__restrict__ org,
                                                   const double *__restrict__ mu, const int *__restrict__ defcol,
                                                   const double *__restrict__ defval, const double *__restrict__ Gm,
                                                   int64_t ldx, const double *__restrict__ Z, int64_t ldz_in,
                                                   double *__restrict__ Znew, int64_t ldz, double *__restrict__ wnew,
                                                   const double *__restrict__ sc, int last, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    dc_finish_body(merges, Kin, dl, org, mu, defcol, defval, Gm, ldx, Z, ldz_in, Znew, ldz, wnew, sc, last, (int)blockIdx.y, (int)blockIdx.x, (int)gridDim.x, dyn_smem);
}

// ---- small merges (m <= 64): all phases of one merge by ONE workgroup in ONE launch (six launches per level
//      are pure latency at these sizes).  Pointers deliberately without const/__restrict__: phases read what
//      earlier phases wrote ---------------------------------------------------------------------------------
template <int LANES>
__global__ __launch_bounds__(256) void k_dc_merge_small(const DcMerge *merges, double *w, const double *e, double *Z, int64_t ldz,
                                                        int *Kbuf, double *rho, double *dl, double *zl, int *col, int *defcol,
                                                        double *defval, const double *sc, int *org, double *mu, double *zhat,
                                                        double *X, double *Zp, int64_t ldx, double *G, double *Znew, int64_t ldzn,
                                                        double *wnew, int last, const int *__restrict__ dc_info)
{
    if (*dc_info) return;      // non-finite input / a leaf that did not converge (jcdf.h: stedc info word): nothing downstream may index with it
    extern __shared__ __attribute__((aligned(16))) double dyn_smem[];
    const int mi = blockIdx.x;
    dc_prepare_body(merges, w, e, Z, ldz, Kbuf, rho, dl, zl, col, defcol, defval, sc, mi, 0, 1, dyn_smem);
    __syncthreads();
    dc_secular_body<LANES>(merges, Kbuf, rho, dl, zl, org, mu, mi, 0, 1, dyn_smem);
    __syncthreads();
    dc_zhat_body<LANES>(merges, Kbuf, dl, zl, org, mu, zhat, mi, 0, 1, dyn_smem);
    __syncthreads();
    dc_vectors_body(merges, Kbuf, dl, org, mu, zhat, col, Z, ldz, X, Zp, ldx, 0, mi, 0, 1, dyn_smem);
    __syncthreads();
    dc_update_simple_body(merges, Kbuf, X, Zp, ldx, G, mi, 0, 1, dyn_smem);
    __syncthreads();
    dc_finish_body(merges, Kbuf, dl, org, mu, defcol, defval, G, ldx, Z, ldz, Znew, ldzn, wnew, sc, last, mi, 0, 1, dyn_smem);
}

// rows/columns of blocks that are not merged at this level (odd leaf carried up) are copied unchanged
__global__ void k_dc_carry(const DcMerge *__restrict__ carry, const double *__restrict__ Z, int64_t ldz_in,
                           double *__restrict__ Znew, int64_t ldz, const double *__restrict__ w, double *__restrict__ wnew)
{
    const DcMerge c = carry[blockIdx.y];
    const int m = c.n1;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < (int64_t)m * m; idx += (int64_t)gridDim.x * blockDim.x) {
        const int jc = (int)(idx / m), r = (int)(idx % m);
        Znew[(int64_t)(c.s + jc) * ldz + c.s + r] = Z[(int64_t)(c.s + jc) * ldz_in + c.s + r];
        if (r == 0) wnew[c.s + jc] = w[c.s + jc];
    }
}

}  // namespace jcdf
