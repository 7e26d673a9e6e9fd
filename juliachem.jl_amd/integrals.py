"""Host Gaussian-integral engine of the library, the producer that stands where the
reference's JERI/Libint engines stand (`jeri_engine_thread_df`, `jeri_engine_thread`): two- and three-centre
Coulomb integrals in the layouts the DF path consumes, the one-electron matrices of the SCF wrapper and the
Schwarz data of the screening builder.  All arithmetic is in libjcdf_hip.so (include/jcint.h,
csrc/jcint_host.cpp: McMurchie-Davidson, threads over shell pairs); integrals stay on the host, as in the
reference (north_star).  Basis data is input — a dict symbol -> list of {"l", "exps", "coefs"} shells in table
order, the format of tests/golden/*.json — since the reference's basis blobs (records/bsed.h5) are not part of the
snapshot."""
from __future__ import annotations

import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _lib
from ._lib import JCDFError
from .df import DFIntegralEngine, ScreeningData, packed_pq_lists

_P = C.c_void_p
_I64 = C.c_int64
_PROTOS = {
    "jcint_basis_create": (C.c_int32, [C.POINTER(_P), _I64, _P, _P, _P, _P, _P]),
    "jcint_basis_destroy": (None, [_P]),
    "jcint_nbf": (_I64, [_P]),
    "jcint_nshell": (_I64, [_P]),
    "jcint_shell_sizes": (C.c_int32, [_P, _P]),
    "jcint_one_electron": (C.c_int32, [_P, _I64, _P, _P, _P, _P, _P]),
    "jcint_nuclear_repulsion": (C.c_double, [_I64, _P, _P]),
    "jcint_two_center": (C.c_int32, [_P, _P]),
    "jcint_three_center": (C.c_int32, [_P, _P, _I64, _I64, _I64, _P, _P, _P]),
    "jcint_schwarz": (C.c_int32, [_P, _P, _P]),
    "jcint_set_threads": (None, [C.c_int32]),
}
_bound = False


def _load():
    global _bound
    lib = _lib.load()
    if not _bound:
        for name, (res, args) in _PROTOS.items():
            f = getattr(lib, name)
            f.restype, f.argtypes = res, args
        _bound = True
    return lib


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise JCDFError(rc, "%s failed (status %d)" % (what, rc))


class HostBasis:
    """Contracted Cartesian shells of all atoms in input order (the reference's `Basis`, BasisStructs.jl)."""

    def __init__(self, atoms: Sequence[Dict], basis: Dict[str, List[Dict]]):
        ls, nps, exps, coefs, cen = [], [], [], [], []
        for at in atoms:
            for sh in basis[at["symbol"]]:
                ls.append(int(sh["l"]))
                nps.append(len(sh["exps"]))
                exps += list(map(float, sh["exps"]))
                coefs += list(map(float, sh["coefs"]))
                cen += list(map(float, at["center"]))
        self._lib = _load()
        self._h = _P()
        l = np.asarray(ls, dtype=np.int32); n = np.asarray(nps, dtype=np.int32)
        e = np.asarray(exps, dtype=np.float64); c = np.asarray(coefs, dtype=np.float64)
        r = np.asarray(cen, dtype=np.float64)
        _check(self._lib.jcint_basis_create(C.byref(self._h), len(ls), l.ctypes.data, n.ctypes.data, e.ctypes.data,
                                            c.ctypes.data, r.ctypes.data), "jcint_basis_create")
        self.nbf = int(self._lib.jcint_nbf(self._h))
        self.nshell = int(self._lib.jcint_nshell(self._h))
        sizes = np.zeros(self.nshell, dtype=np.int64)
        _check(self._lib.jcint_shell_sizes(self._h, sizes.ctypes.data), "jcint_shell_sizes")
        self.shell_nbas = [int(x) for x in sizes]
        self.shell_pos = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)

    def close(self) -> None:
        if self._h:
            self._lib.jcint_basis_destroy(self._h)
            self._h = _P()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostIntegralEngine(DFIntegralEngine):
    """Integrals of one molecule from basis-set data, behind the `DFIntegralEngine` interface of the operator
    (`df_rhf_fock_build(scf_data, engine, ...)`) plus the one-electron part the SCF wrapper needs."""

    def __init__(self, atoms: Sequence[Dict], basis: Dict[str, List[Dict]], aux_basis: Dict[str, List[Dict]],
                 charges: Dict[str, float]):
        self.atoms = list(atoms)
        self.prim = HostBasis(atoms, basis)
        self.aux = HostBasis(atoms, aux_basis)
        self.Z = np.asarray([float(charges[a["symbol"]]) for a in atoms], dtype=np.float64)
        self.R = np.ascontiguousarray([a["center"] for a in atoms], dtype=np.float64)
        self._lib = _load()
        self._schwarz = None

    # ---- one-electron part (jeri-oei.hpp:61,106,155; EnergyHelpers.jl:5-23) ----------------------------------
    def one_electron(self):
        N = self.prim.nbf
        S, T, V = (np.zeros((N, N), order="F") for _ in range(3))
        _check(self._lib.jcint_one_electron(self.prim._h, len(self.Z), self.Z.ctypes.data, self.R.ctypes.data,
                                            S.ctypes.data, T.ctypes.data, V.ctypes.data), "jcint_one_electron")
        return S, T, V

    def nuclear_repulsion(self) -> float:
        return float(self._lib.jcint_nuclear_repulsion(len(self.Z), self.Z.ctypes.data, self.R.ctypes.data))

    # ---- DFIntegralEngine ------------------------------------------------------------------------------------
    def calculate_two_center_intgrals(self) -> np.ndarray:
        Q = self.aux.nbf
        J = np.zeros((Q, Q), order="F")
        _check(self._lib.jcint_two_center(self.aux._h, J.ctypes.data), "jcint_two_center")
        return np.asfortranarray(np.tril(J))                      # lower triangle valid, upper zero (TwoCenterIntegrals.jl:150-162)

    def calculate_three_center_integrals(self, aux_range: range, sd: Optional[ScreeningData]) -> np.ndarray:
        N = self.prim.nbf
        q0, q1 = aux_range.start, aux_range.stop
        if sd is None:
            P, pp, pq = N * N, None, None
        else:
            pq_p, pq_q = packed_pq_lists(sd)
            pp = np.ascontiguousarray(pq_p, dtype=np.int64); pq = np.ascontiguousarray(pq_q, dtype=np.int64)
            P = len(pp)
        T = np.zeros((q1 - q0, P), order="F")
        _check(self._lib.jcint_three_center(self.aux._h, self.prim._h, q0, q1, P, pp.ctypes.data if pp is not None else None,
                                            pq.ctypes.data if pq is not None else None, T.ctypes.data), "jcint_three_center")
        return T

    def schwarz_data(self):
        if self._schwarz is None:
            N, ns = self.prim.nbf, self.prim.nshell
            M = np.zeros((N, N), order="F"); sh = np.zeros((ns, ns), order="F")
            _check(self._lib.jcint_schwarz(self.prim._h, M.ctypes.data, sh.ctypes.data), "jcint_schwarz")
            self._schwarz = (M, sh)
        return self._schwarz

    def schwarz_mask(self, sigma: float, max_P_P: float) -> Optional[np.ndarray]:
        """basis_function_screen_matrix of schwarz_screen_itegrals_df (SchwarzScreening.jl:9-71): a shell pair is kept
        iff |sum of its (mn|mn) block| >= sigma^2 / max_P_P, a function pair of a kept shell pair iff |(pq|pq)| >= the same."""
        M, sh = self.schwarz_data()
        thr = sigma * sigma / max_P_P
        shell_keep = ~(np.abs(sh) < thr)
        idx = np.repeat(np.arange(self.prim.nshell), self.prim.shell_nbas)
        return shell_keep[np.ix_(idx, idx)] & ~(np.abs(M) < thr)

    def close(self) -> None:
        self.prim.close()
        self.aux.close()
