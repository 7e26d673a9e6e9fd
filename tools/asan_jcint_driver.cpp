#include "jcint.h"
#include <cstdio>
#include <vector>
#include <cmath>
int main() {
    // water-like: O (s,s,p,d,f), H (s,p) x2
    std::vector<int32_t> l = {0, 0, 1, 2, 3, 0, 1, 0, 1};
    std::vector<int32_t> np = {3, 1, 2, 1, 1, 2, 1, 2, 1};
    std::vector<double> ex, co, cen;
    double pos[3][3] = {{0, -0.14, 0}, {1.64, 1.14, 0}, {-1.64, 1.14, 0}};
    int atom_of[9] = {0, 0, 0, 0, 0, 1, 1, 2, 2};
    for (size_t s = 0; s < l.size(); ++s) {
        for (int p = 0; p < np[s]; ++p) { ex.push_back(0.3 + 1.7 * p + 0.2 * s); co.push_back(0.4 + 0.1 * p); }
        for (int d = 0; d < 3; ++d) cen.push_back(pos[atom_of[s]][d]);
    }
    jcint_basis *b = nullptr, *a = nullptr;
    if (jcint_basis_create(&b, (int64_t)l.size(), l.data(), np.data(), ex.data(), co.data(), cen.data())) return 1;
    std::vector<int32_t> la = {0, 1, 2, 3, 4, 0, 1, 0, 1};
    if (jcint_basis_create(&a, (int64_t)la.size(), la.data(), np.data(), ex.data(), co.data(), cen.data())) return 1;
    const int64_t N = jcint_nbf(b), Q = jcint_nbf(a);
    std::vector<double> S(N * N), T(N * N), V(N * N), J(Q * Q), T3(Q * N * N), M(N * N), sh(81);
    double Z[3] = {8, 1, 1}, R[9];
    for (int i = 0; i < 3; ++i) for (int d = 0; d < 3; ++d) R[3 * i + d] = pos[i][d];
    jcint_set_threads(3);
    int rc = jcint_one_electron(b, 3, Z, R, S.data(), T.data(), V.data());
    rc |= jcint_two_center(a, J.data());
    rc |= jcint_three_center(a, b, 0, Q, N * N, nullptr, nullptr, T3.data());
    rc |= jcint_schwarz(b, M.data(), sh.data());
    double dmax = 0; for (int64_t i = 0; i < N; ++i) dmax = std::fmax(dmax, std::fabs(S[i + N * i] - 1.0));
    printf("rc=%d N=%lld Q=%lld |diag(S)-1|max=%.2e J00=%.6f Enuc=%.6f\n", rc, (long long)N, (long long)Q, dmax, J[0], jcint_nuclear_repulsion(3, Z, R));
    jcint_basis_destroy(a); jcint_basis_destroy(b);
    return rc;
}
