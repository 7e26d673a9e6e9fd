// jcint_host.cpp — host Gaussian-integral engine behind include/jcint.h.
// McMurchie-Davidson scheme (Helgaker, Jorgensen, Olsen, "Molecular Electronic-Structure Theory", ch. 9):
//   a product of two Cartesian Gaussians is expanded in Hermite Gaussians (coefficients E^{ij}_t per direction),
//   Coulomb integrals between Hermite charge distributions are the Hermite integrals R_{tuv} of the Boys function.
// It stands where the reference calls Libint 2.7.0 (deps/src/jeri-df-tei.hpp:51-95, jeri-oei.hpp:61,106,155,
// jeri-tei.hpp:67-70); conventions in include/jcint.h.  Pure host C++17 + std::thread; no device code.
#include "../../include/jcint.h"
#include "../../include/jcdf.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <functional>
#include <new>
#include <thread>
#include <vector>

namespace {

constexpr int LMAX = 6;
int g_threads = 0;

struct Cart {
    int x, y, z;
};

std::vector<Cart> cart_list(int l)                       // Libint order: lx descending, then ly descending
{
    std::vector<Cart> out;
    for (int lx = l; lx >= 0; --lx)
        for (int ly = l - lx; ly >= 0; --ly) out.push_back({lx, ly, l - lx - ly});
    return out;
}

double dfact(int n)                                      // n!! with (-1)!! = 0!! = 1
{
    double r = 1.0;
    for (int k = n; k > 1; k -= 2) r *= k;
    return r;
}

struct Shell {
    int l = 0, nbas = 0, off = 0;                        // off: first basis function
    double R[3] = {0, 0, 0};
    std::vector<double> a;                               // exponents
    std::vector<double> c;                               // [cart][prim]: coefficients of UNnormalised primitives
    std::vector<Cart> carts;
};

}  // namespace

struct jcint_basis {
    std::vector<Shell> shells;
    int64_t nbf = 0;
};

namespace {

// every contracted Cartesian function gets unit self-overlap (see jcint.h)
void normalise(Shell &sh, const double *coefs)
{
    const int np = (int)sh.a.size(), l = sh.l;
    sh.carts = cart_list(l);
    sh.nbas = (int)sh.carts.size();
    sh.c.assign((size_t)sh.nbas * np, 0.0);
    for (int k = 0; k < sh.nbas; ++k) {
        const Cart ct = sh.carts[k];
        const double dd = dfact(2 * ct.x - 1) * dfact(2 * ct.y - 1) * dfact(2 * ct.z - 1);
        std::vector<double> c(np);
        for (int i = 0; i < np; ++i)
            c[i] = coefs[i] * std::pow(2.0 * sh.a[i] / M_PI, 0.75) * std::pow(4.0 * sh.a[i], 0.5 * l) / std::sqrt(dd);
        double s = 0.0;
        for (int i = 0; i < np; ++i)
            for (int j = 0; j < np; ++j) {
                const double p = sh.a[i] + sh.a[j];
                s += c[i] * c[j] * std::pow(M_PI / p, 1.5) * dd / std::pow(2.0 * p, l);
            }
        const double inv = 1.0 / std::sqrt(s);
        for (int i = 0; i < np; ++i) sh.c[(size_t)k * np + i] = c[i] * inv;
    }
}

// ---- Hermite expansion coefficients of one Cartesian direction: E[(i*(lb+1)+j)*(la+lb+1) + t] ----------------
void hermite_E(int la, int lb, double a, double b, double XAB, std::vector<double> &E)
{
    const int nt = la + lb + 2;                          // one spare t for the recursion
    const double p = a + b, mu = a * b / p, XPA = -b / p * XAB, XPB = a / p * XAB, h = 0.5 / p;
    std::vector<double> W((size_t)(la + 1) * (lb + 1) * nt, 0.0);
    auto w = [&](int i, int j, int t) -> double & { return W[((size_t)i * (lb + 1) + j) * nt + t]; };
    w(0, 0, 0) = std::exp(-mu * XAB * XAB);
    for (int i = 0; i < la; ++i)
        for (int t = 0; t <= i + 1; ++t)
            w(i + 1, 0, t) = XPA * w(i, 0, t) + (t + 1) * w(i, 0, t + 1) + (t > 0 ? h * w(i, 0, t - 1) : 0.0);
    for (int i = 0; i <= la; ++i)
        for (int j = 0; j < lb; ++j)
            for (int t = 0; t <= i + j + 1; ++t)
                w(i, j + 1, t) = XPB * w(i, j, t) + (t + 1) * w(i, j, t + 1) + (t > 0 ? h * w(i, j, t - 1) : 0.0);
    const int ne = la + lb + 1;
    E.assign((size_t)(la + 1) * (lb + 1) * ne, 0.0);
    for (int i = 0; i <= la; ++i)
        for (int j = 0; j <= lb; ++j)
            for (int t = 0; t < ne; ++t) E[((size_t)i * (lb + 1) + j) * ne + t] = w(i, j, t);
}

// ---- Boys function F_n(x), n = 0..nmax ---------------------------------------------------------------------------
void boys(int nmax, double x, double *F)
{
    if (x < 35.0) {                                      // convergent series for F_nmax, then downward recursion
        const double ex = std::exp(-x);
        double term = 1.0 / (2 * nmax + 1), sum = term;
        for (int k = 1; k < 400; ++k) {
            term *= 2.0 * x / (2 * nmax + 2 * k + 1);
            sum += term;
            if (term < 1e-17 * sum) break;
        }
        F[nmax] = ex * sum;
        for (int n = nmax; n > 0; --n) F[n - 1] = (2.0 * x * F[n] + ex) / (2 * n - 1);
    } else {                                             // asymptotic F_0, upward recursion (stable for large x)
        const double ex = std::exp(-x);
        F[0] = 0.5 * std::sqrt(M_PI / x);
        for (int n = 0; n < nmax; ++n) F[n + 1] = ((2 * n + 1) * F[n] - ex) / (2.0 * x);
    }
}

// ---- Hermite Coulomb integrals R_{tuv}(alpha, PQ), t+u+v <= L, stored R[(t*(L+1)+u)*(L+1)+v] -----------------------
struct RWork {
    std::vector<double> cur;                              // [m][t][u][v]
};
void hermite_R(int L, double alpha, const double *PQ, double *R, RWork &wk)
{
    const int n1 = L + 1;
    double F[4 * LMAX + 2];
    boys(L, alpha * (PQ[0] * PQ[0] + PQ[1] * PQ[1] + PQ[2] * PQ[2]), F);
    wk.cur.assign((size_t)n1 * n1 * n1 * n1, 0.0);
    auto c = [&](int m, int t, int u, int v) -> double & { return wk.cur[(((size_t)m * n1 + t) * n1 + u) * n1 + v]; };
    double pw = 1.0;
    for (int m = 0; m <= L; ++m) {
        c(m, 0, 0, 0) = pw * F[m];
        pw *= -2.0 * alpha;
    }
    for (int t = 0; t <= L; ++t)
        for (int u = 0; u <= L - t; ++u)
            for (int v = 0; v <= L - t - u; ++v) {
                if (t + u + v == 0) continue;
                const int mmax = L - (t + u + v);
                for (int m = 0; m <= mmax; ++m) {
                    double val;
                    if (t > 0) val = PQ[0] * c(m + 1, t - 1, u, v) + (t > 1 ? (t - 1) * c(m + 1, t - 2, u, v) : 0.0);
                    else if (u > 0) val = PQ[1] * c(m + 1, t, u - 1, v) + (u > 1 ? (u - 1) * c(m + 1, t, u - 2, v) : 0.0);
                    else val = PQ[2] * c(m + 1, t, u, v - 1) + (v > 1 ? (v - 1) * c(m + 1, t, u, v - 2) : 0.0);
                    c(m, t, u, v) = val;
                }
            }
    for (int t = 0; t <= L; ++t)
        for (int u = 0; u <= L; ++u)
            for (int v = 0; v <= L; ++v) R[((size_t)t * n1 + u) * n1 + v] = (t + u + v <= L) ? c(0, t, u, v) : 0.0;
}

// ---- Hermite density of a shell pair (or a single shell): per primitive pair the list of non-zero
//      (function, t, u, v, weight) entries --------------------------------------------------------------------------
struct PairPrim {
    double p, P[3];
    std::vector<int> func, t, u, v;
    std::vector<double> w;
};
struct PairData {
    int L = 0, nfunc = 0;
    std::vector<PairPrim> prims;
};

PairData pair_data(const Shell &sa, const Shell *sb)
{
    PairData pd;
    const int la = sa.l, lb = sb ? sb->l : 0, nb = sb ? sb->nbas : 1;
    pd.L = la + lb;
    pd.nfunc = sa.nbas * nb;
    const int npa = (int)sa.a.size(), npb = sb ? (int)sb->a.size() : 1;
    const double *A = sa.R, *B = sb ? sb->R : sa.R;
    std::vector<double> E[3];
    const int ne = la + lb + 1;
    for (int ia = 0; ia < npa; ++ia)
        for (int ib = 0; ib < npb; ++ib) {
            const double a = sa.a[ia], b = sb ? sb->a[ib] : 0.0;
            PairPrim pp;
            pp.p = a + b;
            for (int d = 0; d < 3; ++d) {
                pp.P[d] = (a * A[d] + b * B[d]) / pp.p;
                hermite_E(la, lb, a, b, A[d] - B[d], E[d]);
            }
            for (int ka = 0; ka < sa.nbas; ++ka)
                for (int kb = 0; kb < nb; ++kb) {
                    const Cart ca = sa.carts[ka];
                    const Cart cb = sb ? sb->carts[kb] : Cart{0, 0, 0};
                    const double wgt = sa.c[(size_t)ka * npa + ia] * (sb ? sb->c[(size_t)kb * npb + ib] : 1.0);
                    const double *ex = &E[0][((size_t)ca.x * (lb + 1) + cb.x) * ne];
                    const double *ey = &E[1][((size_t)ca.y * (lb + 1) + cb.y) * ne];
                    const double *ez = &E[2][((size_t)ca.z * (lb + 1) + cb.z) * ne];
                    for (int t = 0; t <= ca.x + cb.x; ++t)
                        for (int u = 0; u <= ca.y + cb.y; ++u)
                            for (int v = 0; v <= ca.z + cb.z; ++v) {
                                const double val = wgt * ex[t] * ey[u] * ez[v];
                                if (val == 0.0) continue;
                                pp.func.push_back(ka * nb + kb);
                                pp.t.push_back(t);
                                pp.u.push_back(u);
                                pp.v.push_back(v);
                                pp.w.push_back(val);
                            }
                }
            pd.prims.push_back(std::move(pp));
        }
    return pd;
}

// out[fb * ket.nfunc + fk] = (bra function fb | ket function fk), contracted
void eri_block(const PairData &bra, const PairData &ket, double *out, RWork &wk, std::vector<double> &Rbuf)
{
    const int L = bra.L + ket.L, n1 = L + 1;
    std::fill(out, out + (size_t)bra.nfunc * ket.nfunc, 0.0);
    Rbuf.resize((size_t)n1 * n1 * n1);
    for (const PairPrim &pb : bra.prims)
        for (const PairPrim &pk : ket.prims) {
            const double p = pb.p, q = pk.p, alpha = p * q / (p + q);
            const double PQ[3] = {pb.P[0] - pk.P[0], pb.P[1] - pk.P[1], pb.P[2] - pk.P[2]};
            hermite_R(L, alpha, PQ, Rbuf.data(), wk);
            const double pref = 2.0 * std::pow(M_PI, 2.5) / (p * q * std::sqrt(p + q));
            const size_t nk = pk.w.size();
            for (size_t ib = 0; ib < pb.w.size(); ++ib) {
                const double wb = pb.w[ib] * pref;
                const int t = pb.t[ib], u = pb.u[ib], v = pb.v[ib];
                double *row = out + (size_t)pb.func[ib] * ket.nfunc;
                for (size_t ik = 0; ik < nk; ++ik) {
                    const int x = pk.t[ik], y = pk.u[ik], z = pk.v[ik];
                    const double sgn = ((x + y + z) & 1) ? -1.0 : 1.0;
                    row[pk.func[ik]] += wb * sgn * pk.w[ik] * Rbuf[((size_t)(t + x) * n1 + (u + y)) * n1 + (v + z)];
                }
            }
        }
}

void parallel_for(int64_t n, const std::function<void(int64_t, int)> &body)
{
    int nt = g_threads > 0 ? g_threads : (int)std::thread::hardware_concurrency();
    nt = std::max(1, std::min(nt, 64));
    nt = (int)std::min<int64_t>(nt, std::max<int64_t>(1, n));
    std::atomic<int64_t> next{0};
    auto worker = [&](int tid) {
        for (;;) {
            const int64_t i = next.fetch_add(1);
            if (i >= n) break;
            body(i, tid);
        }
    };
    if (nt == 1) {
        worker(0);
        return;
    }
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(worker, t);
    worker(0);
    for (auto &t : th) t.join();
}

}  // namespace

// ============================================================================================================
extern "C" {

void jcint_set_threads(int32_t n) { g_threads = n; }

int32_t jcint_basis_create(jcint_basis **out, int64_t nshell, const int32_t *l, const int32_t *nprim, const double *exps,
                           const double *coefs, const double *centers)
{
    if (!out || nshell <= 0 || !l || !nprim || !exps || !coefs || !centers) return JCDF_ERR_INVALID;
    jcint_basis *b = new (std::nothrow) jcint_basis;
    if (!b) return JCDF_ERR_ALLOC;
    int64_t pos = 0, off = 0;
    try {
        for (int64_t s = 0; s < nshell; ++s) {
            if (l[s] < 0 || l[s] > LMAX || nprim[s] <= 0) {
                delete b;
                return JCDF_ERR_INVALID;
            }
            Shell sh;
            sh.l = l[s];
            sh.a.assign(exps + pos, exps + pos + nprim[s]);
            for (int d = 0; d < 3; ++d) sh.R[d] = centers[3 * s + d];
            normalise(sh, coefs + pos);
            sh.off = (int)off;
            off += sh.nbas;
            pos += nprim[s];
            b->shells.push_back(std::move(sh));
        }
    } catch (...) {
        delete b;
        return JCDF_ERR_ALLOC;
    }
    b->nbf = off;
    *out = b;
    return JCDF_OK;
}

void jcint_basis_destroy(jcint_basis *b) { delete b; }
int64_t jcint_nbf(const jcint_basis *b) { return b ? b->nbf : 0; }
int64_t jcint_nshell(const jcint_basis *b) { return b ? (int64_t)b->shells.size() : 0; }

int32_t jcint_shell_sizes(const jcint_basis *b, int64_t *nbas_out)
{
    if (!b || !nbas_out) return JCDF_ERR_INVALID;
    for (size_t s = 0; s < b->shells.size(); ++s) nbas_out[s] = b->shells[s].nbas;
    return JCDF_OK;
}

double jcint_nuclear_repulsion(int64_t natoms, const double *Z, const double *R)
{
    double e = 0.0;
    for (int64_t i = 0; i < natoms; ++i)
        for (int64_t j = 0; j < i; ++j) {
            const double dx = R[3 * i] - R[3 * j], dy = R[3 * i + 1] - R[3 * j + 1], dz = R[3 * i + 2] - R[3 * j + 2];
            e += Z[i] * Z[j] / std::sqrt(dx * dx + dy * dy + dz * dz);
        }
    return e;
}

int32_t jcint_one_electron(const jcint_basis *b, int64_t natoms, const double *Z, const double *R, double *S, double *T,
                           double *V)
{
    if (!b || (V && natoms > 0 && (!Z || !R))) return JCDF_ERR_INVALID;
    const int64_t N = b->nbf, ns = (int64_t)b->shells.size();
    try {
        parallel_for(ns * (ns + 1) / 2, [&](int64_t pair, int) {
            int64_t ia = (int64_t)((std::sqrt(8.0 * pair + 1.0) - 1.0) / 2.0);
            while (ia * (ia + 1) / 2 > pair) --ia;
            while ((ia + 1) * (ia + 2) / 2 <= pair) ++ia;
            const int64_t ib = pair - ia * (ia + 1) / 2;
            const Shell &sa = b->shells[ia], &sb = b->shells[ib];
            const int la = sa.l, lb = sb.l, npa = (int)sa.a.size(), npb = (int)sb.a.size();
            std::vector<double> E[3], Rn;
            RWork wk;
            const int ne = la + lb + 3, Lh = la + lb, n1 = Lh + 1;     // E computed with lb + 2 for the kinetic energy
            std::vector<double> sv((size_t)sa.nbas * sb.nbas, 0.0), tv(sv), vv(sv);
            Rn.resize((size_t)n1 * n1 * n1);
            for (int pa = 0; pa < npa; ++pa)
                for (int pb = 0; pb < npb; ++pb) {
                    const double a = sa.a[pa], bb = sb.a[pb], p = a + bb;
                    double P[3];
                    for (int d = 0; d < 3; ++d) {
                        hermite_E(la, lb + 2, a, bb, sa.R[d] - sb.R[d], E[d]);
                        P[d] = (a * sa.R[d] + bb * sb.R[d]) / p;
                    }
                    const double sq = std::sqrt(M_PI / p);
                    auto e0 = [&](int d, int i, int j) { return E[d][((size_t)i * (lb + 3) + j) * ne] * sq; };   // 1-D overlap
                    auto kin = [&](int d, int i, int j) {
                        double t = -2.0 * bb * bb * e0(d, i, j + 2) + bb * (2 * j + 1) * e0(d, i, j);
                        if (j >= 2) t -= 0.5 * j * (j - 1) * e0(d, i, j - 2);
                        return t;
                    };
                    for (int ka = 0; ka < sa.nbas; ++ka)
                        for (int kb = 0; kb < sb.nbas; ++kb) {
                            const Cart ca = sa.carts[ka], cb = sb.carts[kb];
                            const double w = sa.c[(size_t)ka * npa + pa] * sb.c[(size_t)kb * npb + pb];
                            const double sx = e0(0, ca.x, cb.x), sy = e0(1, ca.y, cb.y), sz = e0(2, ca.z, cb.z);
                            sv[(size_t)ka * sb.nbas + kb] += w * sx * sy * sz;
                            tv[(size_t)ka * sb.nbas + kb] += w * (kin(0, ca.x, cb.x) * sy * sz + sx * kin(1, ca.y, cb.y) * sz +
                                                                 sx * sy * kin(2, ca.z, cb.z));
                        }
                    if (V)
                        for (int64_t at = 0; at < natoms; ++at) {
                            const double PC[3] = {P[0] - R[3 * at], P[1] - R[3 * at + 1], P[2] - R[3 * at + 2]};
                            hermite_R(Lh, p, PC, Rn.data(), wk);
                            for (int ka = 0; ka < sa.nbas; ++ka)
                                for (int kb = 0; kb < sb.nbas; ++kb) {
                                    const Cart ca = sa.carts[ka], cb = sb.carts[kb];
                                    const double w = sa.c[(size_t)ka * npa + pa] * sb.c[(size_t)kb * npb + pb];
                                    const double *ex = &E[0][((size_t)ca.x * (lb + 3) + cb.x) * ne];
                                    const double *ey = &E[1][((size_t)ca.y * (lb + 3) + cb.y) * ne];
                                    const double *ez = &E[2][((size_t)ca.z * (lb + 3) + cb.z) * ne];
                                    double acc = 0.0;
                                    for (int t = 0; t <= ca.x + cb.x; ++t)
                                        for (int u = 0; u <= ca.y + cb.y; ++u)
                                            for (int v = 0; v <= ca.z + cb.z; ++v)
                                                acc += ex[t] * ey[u] * ez[v] * Rn[((size_t)t * n1 + u) * n1 + v];
                                    vv[(size_t)ka * sb.nbas + kb] -= Z[at] * w * 2.0 * M_PI / p * acc;
                                }
                        }
                }
            for (int ka = 0; ka < sa.nbas; ++ka)
                for (int kb = 0; kb < sb.nbas; ++kb) {
                    const int64_t i = sa.off + ka, j = sb.off + kb;
                    const size_t k = (size_t)ka * sb.nbas + kb;
                    if (S) S[i + N * j] = S[j + N * i] = sv[k];
                    if (T) T[i + N * j] = T[j + N * i] = tv[k];
                    if (V) V[i + N * j] = V[j + N * i] = vv[k];
                }
        });
    } catch (...) {
        return JCDF_ERR_ALLOC;
    }
    return JCDF_OK;
}

int32_t jcint_two_center(const jcint_basis *aux, double *J)
{
    if (!aux || !J) return JCDF_ERR_INVALID;
    const int64_t Q = aux->nbf, ns = (int64_t)aux->shells.size();
    try {
        std::vector<PairData> pd(ns);
        parallel_for(ns, [&](int64_t s, int) { pd[s] = pair_data(aux->shells[s], nullptr); });
        parallel_for(ns * (ns + 1) / 2, [&](int64_t pair, int) {
            int64_t i = (int64_t)((std::sqrt(8.0 * pair + 1.0) - 1.0) / 2.0);
            while (i * (i + 1) / 2 > pair) --i;
            while ((i + 1) * (i + 2) / 2 <= pair) ++i;
            const int64_t j = pair - i * (i + 1) / 2;
            RWork wk;
            std::vector<double> Rb, blk((size_t)pd[i].nfunc * pd[j].nfunc);
            eri_block(pd[i], pd[j], blk.data(), wk, Rb);
            const Shell &si = aux->shells[i], &sj = aux->shells[j];
            for (int a = 0; a < si.nbas; ++a)
                for (int c = 0; c < sj.nbas; ++c) {
                    const double v = blk[(size_t)a * sj.nbas + c];
                    J[(si.off + a) + Q * (sj.off + c)] = v;
                    J[(sj.off + c) + Q * (si.off + a)] = v;
                }
        });
    } catch (...) {
        return JCDF_ERR_ALLOC;
    }
    return JCDF_OK;
}

int32_t jcint_three_center(const jcint_basis *aux, const jcint_basis *prim, int64_t q0, int64_t q1, int64_t P,
                           const int64_t *pq_p, const int64_t *pq_q, double *T)
{
    if (!aux || !prim || !T || q0 < 0 || q1 <= q0 || q1 > aux->nbf) return JCDF_ERR_INVALID;
    const int64_t N = prim->nbf, R = q1 - q0;
    if ((pq_p == nullptr) != (pq_q == nullptr)) return JCDF_ERR_INVALID;
    if (!pq_p && P != N * N) return JCDF_ERR_INVALID;
    // aux shells of the range (must be shell aligned)
    int64_t s0 = -1, s1 = -1;
    for (size_t s = 0; s < aux->shells.size(); ++s) {
        if (aux->shells[s].off == q0) s0 = (int64_t)s;
        if (aux->shells[s].off + aux->shells[s].nbas == q1) s1 = (int64_t)s + 1;
    }
    if (s0 < 0 || s1 <= s0) return JCDF_ERR_INVALID;
    try {
        // packed index of every kept (q, p) pair (0-based; -1 = screened)
        std::vector<int64_t> map;
        if (pq_p) {
            map.assign((size_t)(N * N), -1);
            for (int64_t c = 0; c < P; ++c) {
                if (pq_p[c] < 0 || pq_p[c] >= N || pq_q[c] < 0 || pq_q[c] >= N) return JCDF_ERR_INVALID;
                map[(size_t)(pq_q[c] + N * pq_p[c])] = c;
            }
        }
        std::vector<PairData> pda(s1 - s0);
        parallel_for(s1 - s0, [&](int64_t s, int) { pda[s] = pair_data(aux->shells[s0 + s], nullptr); });
        const int64_t ns = (int64_t)prim->shells.size();
        parallel_for(ns * (ns + 1) / 2, [&](int64_t pair, int) {
            int64_t m = (int64_t)((std::sqrt(8.0 * pair + 1.0) - 1.0) / 2.0);
            while (m * (m + 1) / 2 > pair) --m;
            while ((m + 1) * (m + 2) / 2 <= pair) ++m;
            const int64_t n = pair - m * (m + 1) / 2;
            const Shell &sm = prim->shells[m], &sn = prim->shells[n];
            if (pq_p) {                                     // a shell pair without any kept function pair is skipped
                bool any = false;                           // (ThreeCenterIntegralsScreened.jl:8-85 skips screened shell pairs)
                for (int a = 0; a < sm.nbas && !any; ++a)
                    for (int c = 0; c < sn.nbas && !any; ++c)
                        any = map[(size_t)((sm.off + a) + N * (sn.off + c))] >= 0 || map[(size_t)((sn.off + c) + N * (sm.off + a))] >= 0;
                if (!any) return;
            }
            const PairData ket = pair_data(sm, &sn);
            RWork wk;
            std::vector<double> Rb, blk;
            for (int64_t s = 0; s < s1 - s0; ++s) {
                const Shell &sq = aux->shells[s0 + s];
                blk.resize((size_t)pda[s].nfunc * ket.nfunc);
                eri_block(pda[s], ket, blk.data(), wk, Rb);
                for (int qa = 0; qa < sq.nbas; ++qa) {
                    const int64_t row = sq.off + qa - q0;
                    for (int a = 0; a < sm.nbas; ++a)
                        for (int c = 0; c < sn.nbas; ++c) {
                            const double v = blk[(size_t)qa * ket.nfunc + (size_t)a * sn.nbas + c];
                            const int64_t mu = sm.off + a, nu = sn.off + c;
                            // both orders: packed index c(q, p) with p outer, q inner
                            const int64_t c1 = pq_p ? map[(size_t)(mu + N * nu)] : mu + N * nu;
                            const int64_t c2 = pq_p ? map[(size_t)(nu + N * mu)] : nu + N * mu;
                            if (c1 >= 0) T[row + R * c1] = v;
                            if (c2 >= 0) T[row + R * c2] = v;
                        }
                }
            }
        });
    } catch (...) {
        return JCDF_ERR_ALLOC;
    }
    return JCDF_OK;
}

int32_t jcint_schwarz(const jcint_basis *prim, double *M, double *shell_sum)
{
    if (!prim || !M) return JCDF_ERR_INVALID;
    const int64_t N = prim->nbf, ns = (int64_t)prim->shells.size();
    try {
        parallel_for(ns * (ns + 1) / 2, [&](int64_t pair, int) {
            int64_t m = (int64_t)((std::sqrt(8.0 * pair + 1.0) - 1.0) / 2.0);
            while (m * (m + 1) / 2 > pair) --m;
            while ((m + 1) * (m + 2) / 2 <= pair) ++m;
            const int64_t n = pair - m * (m + 1) / 2;
            const Shell &sm = prim->shells[m], &sn = prim->shells[n];
            const PairData pd = pair_data(sm, &sn);
            RWork wk;
            std::vector<double> Rb, blk((size_t)pd.nfunc * pd.nfunc);
            eri_block(pd, pd, blk.data(), wk, Rb);
            if (shell_sum) {                                // the reference's shell-pair test sums the whole (mn|mn) block
                double sum = 0.0;
                for (double x : blk) sum += x;
                shell_sum[m + ns * n] = shell_sum[n + ns * m] = sum;
            }
            for (int a = 0; a < sm.nbas; ++a)
                for (int c = 0; c < sn.nbas; ++c) {
                    const size_t f = (size_t)a * sn.nbas + c;
                    const double v = blk[f * pd.nfunc + f];
                    M[(sm.off + a) + N * (sn.off + c)] = v;
                    M[(sn.off + c) + N * (sm.off + a)] = v;
                }
        });
    } catch (...) {
        return JCDF_ERR_ALLOC;
    }
    return JCDF_OK;
}

}  // extern "C"
