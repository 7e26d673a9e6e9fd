"""CPU prototype (numpy) of the two-stage tridiagonalisation the library runs on the device
(csrc/jcdf_sbr.hpp): stage 1 dense -> band (blocked Householder QR panels, two-sided block-reflector updates),
stage 2 band -> tridiagonal (bulge chasing on the lower band storage AB[j][d] = A[j+d][j], d < 2b), with the
orthogonal factor accumulated row-wise.  Index conventions are the kernels'; used to check them and by nothing else.
"""
import numpy as np


def house(x):
    """LAPACK dlarfg: returns (v with v[0] = 1, tau, beta) such that (I - tau v v^T) x = beta e_1."""
    alpha = x[0]
    sigma = float(np.dot(x[1:], x[1:]))
    v = x.copy()
    v[0] = 1.0
    if sigma == 0.0:
        v[1:] = 0.0
        return v, 0.0, alpha
    beta = -np.copysign(np.sqrt(alpha * alpha + sigma), alpha)
    tau = (beta - alpha) / beta
    v[1:] = x[1:] / (alpha - beta)
    return v, tau, beta


def stage1(A, b):
    """dense -> band of half-width b.  Returns (AB [n][2b] lower band storage, Q1 with A = Q1 B Q1^T)."""
    A = A.copy()
    n = A.shape[0]
    Q = np.eye(n)
    k = 0
    while n - (k + 1) * b >= 2:
        j0, r0 = k * b, (k + 1) * b
        m = n - r0
        P = A[r0:, j0:j0 + b].copy()                       # m x b
        nref = min(b, m - 1)
        V = np.zeros((m, b))
        T = np.zeros((b, b))
        for c in range(nref):
            v, tau, beta = house(P[c:, c])
            w = tau * (v @ P[c:, c + 1:])
            P[c:, c + 1:] -= np.outer(v, w)
            P[c, c] = beta
            P[c + 1:, c] = 0.0
            V[c:, c] = v
            T[c, c] = tau
            if c > 0:
                T[:c, c] = -tau * (T[:c, :c] @ (V[:, :c].T @ V[:, c]))
        A[r0:, j0:j0 + b] = P
        A[j0:j0 + b, r0:] = P.T
        A22 = A[r0:, r0:]
        Y = A22 @ V
        M1 = V.T @ Y
        W = (Y - 0.5 * V @ (T.T @ M1)) @ T
        A22 -= V @ W.T + W @ V.T
        Q[:, r0:] -= ((Q[:, r0:] @ V) @ T) @ V.T
        k += 1
    AB = np.zeros((n, 2 * b))
    for j in range(n):
        for d in range(min(b + 1, n - j)):
            AB[j, d] = A[j + d, j]
    return AB, Q


def band_to_dense(AB):
    n, w = AB.shape
    A = np.zeros((n, n))
    for j in range(n):
        for d in range(min(w, n - j)):
            A[j + d, j] = AB[j, d]
            A[j, j + d] = AB[j, d]
    return A


def stage2(AB, b, Q=None):
    """band -> tridiagonal by bulge chasing, one column per sweep.  AB is overwritten.  Returns (D, E, log) with
    log[(s, t)] = (first row, v, tau); if Q is given it is updated to Q H_(0,0) H_(0,1) ... (row-wise)."""
    n = AB.shape[0]
    log = []

    def getD(r0, L):                                          # symmetric diagonal block rows/cols r0 .. r0+L-1
        Dm = np.zeros((L, L))
        for j in range(L):
            for i in range(j, L):
                Dm[i, j] = Dm[j, i] = AB[r0 + j, i - j]
        return Dm

    def putD(r0, L, Dm):
        for j in range(L):
            for i in range(j, L):
                AB[r0 + j, i - j] = Dm[i, j]

    def getB(r0, L, L2):                                      # block rows r0+L .. r0+L+L2-1, cols r0 .. r0+L-1
        Bm = np.zeros((L2, L))
        for j in range(L):
            for i in range(L2):
                Bm[i, j] = AB[r0 + j, L + i - j]
        return Bm

    def putB(r0, L, L2, Bm):
        for j in range(L):
            for i in range(L2):
                AB[r0 + j, L + i - j] = Bm[i, j]

    for s in range(n - 2):
        r0 = s + 1
        L = min(b, n - r0)
        if L < 2:
            break
        x = AB[s, 1:1 + L].copy()
        v, tau, beta = house(x)
        AB[s, 1] = beta
        AB[s, 2:1 + L] = 0.0
        t = 0
        while True:
            log.append((s, t, r0, v.copy(), tau))
            if Q is not None:
                Q[:, r0:r0 + L] -= tau * np.outer(Q[:, r0:r0 + L] @ v, v)
            Dm = getD(r0, L)
            u = Dm @ v
            g = float(v @ u)
            w = tau * u - 0.5 * tau * tau * g * v
            Dm -= np.outer(v, w) + np.outer(w, v)
            putD(r0, L, Dm)
            L2 = min(b, n - (r0 + L))
            if L2 <= 0:
                break
            assert L == b
            Bm = getB(r0, L, L2)
            Bm -= tau * np.outer(Bm @ v, v)
            if L2 >= 2:
                v2, tau2, beta2 = house(Bm[:, 0].copy())
                Bm[0, 0] = beta2
                Bm[1:, 0] = 0.0
                Bm[:, 1:] -= tau2 * np.outer(v2, v2 @ Bm[:, 1:])
            putB(r0, L, L2, Bm)
            if L2 < 2:
                break
            r0, L, v, tau = r0 + L, L2, v2, tau2
            t += 1
    D = AB[:, 0].copy()
    E = AB[:-1, 1].copy()
    return D, E, log


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for n, b in [(5, 2), (17, 4), (40, 8), (67, 16), (130, 16), (33, 16), (18, 16), (16, 16), (3, 16)]:
        A = rng.standard_normal((n, n))
        A = A + A.T
        AB, Q = stage1(A, b)
        Bd = band_to_dense(AB)
        e1 = np.abs(Q.T @ A @ Q - Bd).max()
        D, E, log = stage2(AB, b, Q)
        Tm = np.diag(D) + np.diag(E, -1) + np.diag(E, 1)
        e2 = np.abs(Q.T @ A @ Q - Tm).max()
        e3 = np.abs(Q.T @ Q - np.eye(n)).max()
        ev = np.abs(np.linalg.eigvalsh(Tm) - np.linalg.eigvalsh(A)).max()
        print("n=%4d b=%2d  |Q1^T A Q1 - B| %.1e  |Q^T A Q - T| %.1e  |Q^T Q - I| %.1e  eig %.1e  steps %d"
              % (n, b, e1, e2, e3, ev, len(log)))
