"""Replicated symmetric eigensolver of the SCF iteration (caller side of the hot path,
SURVEY 8 row f1; reference: `eigen!(Hermitian(F'))`, SCF.jl:1083).

rocSOLVER's syevd needs ~4000 tiny launches for the tridiagonalisation (9 of its 12 ms at
N = 510).  Here every stage is the library's own (libjcdf_hip.so; no vendor kernel on the path):
  * tridiagonalisation: ONE persistent kernel (`jcdf_sytrd_q_device`, csrc/jcdf_eig.hpp) + a one-workgroup finish for the
    last 128 columns; up to n = 1536 it also accumulates the orthogonal factor Q while its reflectors travel between
    workgroups (A = Q T Q^T), so the back-transformation is one GEMM Q Z;
  * tridiagonal eigenproblem: divide & conquer (`jcdf_stedc_device`, csrc/jcdf_dc.hpp: 1.0 ms at N = 510 where
    rocSOLVER's stedc takes 2.6 ms);
  * above n = 1536 (rows of Q no longer fit a workgroup's registers): the first n - 1536 columns through the two-exchange
    kernel, the trailing block through the one-exchange kernel, and the eigenvectors by blocked compact-WY on the MFMA cores
    (`jcdf_ormtr_device`, csrc/jcdf_wy.hpp) — round 3 called rocsolver_dormtr there.
JCDF_EIGH_WY=1 forces the compact-WY back-transformation at every size (tests).  When an in-kernel hand-off reports a
timeout or a stage fails, the step is redone with torch.linalg.eigh (counted in `fallbacks`, never silent).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import torch

from . import _lib

TWO_STAGE_DEFAULT = False     # the two-stage reduction measured at parity (profiles/r02_two_stage_eigh.txt): diagnostic builds only


class DeviceEigh:
    def __init__(self, n: int, device: torch.device):
        self.n, self.device = n, device
        self.U_padded = None
        self.ok = False
        self.calls = self.fallbacks = 0
        self.scratch = False           # True: __call__ may destroy its argument (no copy of the matrix)
        self.timing = False            # True: device events around the tridiagonalisation and the tridiagonal solver of every call
        self.stage_events = []         # [(e0, e1, e2)] per call while `timing`; read with stage_ms()
        try:
            self.lib = _lib.load()
            f64 = dict(dtype=torch.float64, device=device)
            self.A = torch.empty((n, n), **f64)
            self.D = torch.empty(n, **f64)
            self.E = torch.empty(n, **f64)
            self.TAU = torch.zeros(n, **f64)
            self.info = torch.zeros(1, dtype=torch.int32, device=device)     # (kept for callers that fold it into their status word)
            wb = int(self.lib.jcdf_sytrd_workspace_bytes(n))
            self.work = torch.zeros(wb // 8 + 1, **f64)
            self.wb = wb
            if n > int(self.lib.jcdf_sytrd_max_n(0)):
                raise OSError("n = %d is above the tridiagonalisation kernels' limit of %d" % (n, int(self.lib.jcdf_sytrd_max_n(0))))
            self.with_q = n <= int(self.lib.jcdf_sytrd_max_n(1)) and not os.environ.get("JCDF_EIGH_WY")
            # operands of the library's GEMM cores: leading dimension a multiple of 32, zero padding; Zt[j][k] = Z[k][j] is what
            # stedc writes with ldz = npad
            self.npad = (n + 31) // 32 * 32
            self.Qp = torch.zeros((self.npad, self.npad), **f64) if self.with_q else None
            self.Zt = torch.zeros((self.npad, self.npad), **f64)
            self.Up = torch.zeros((self.npad, self.npad), **f64)
            if not self.with_q:
                self.wy_wb = int(self.lib.jcdf_ormtr_workspace_bytes(n))
                self.wy_work = torch.zeros(self.wy_wb // 8 + 8, **f64)
            # two-stage reduction (dense -> band -> tridiagonal, csrc/jcdf_sbr.hpp) where the band fits the LDS of one CU;
            # its Q replay runs on a side stream beside the tridiagonal solver.  JCDF_EIGH_TWO_STAGE=0/1 overrides.
            # (DIAGNOSTIC builds of the library only — the product library does not contain these entry points)
            diag = hasattr(self.lib, "jcdf_sytrd2_device")
            ts = os.environ.get("JCDF_EIGH_TWO_STAGE")
            self.two_stage = (diag and self.with_q and 3 <= n <= int(self.lib.jcdf_sytrd2_max_n())
                              and (ts == "1" if ts is not None else TWO_STAGE_DEFAULT))
            if self.two_stage:
                self.wb2 = int(self.lib.jcdf_sytrd2_workspace_bytes(n))
                self.work2 = torch.zeros(self.wb2 // 8 + 8, **f64)
            # JCDF_EIGH_Q_REPLAY=1 (one-stage kernel, n <= 640): Q rebuilt from the stored reflectors on a side stream beside the
            # divide & conquer instead of inside the persistent kernel — measured equal inside the SCF loop (the kernel's hand-off
            # window, not the Q update inside it, sets the time of a column), so the in-kernel accumulation stays the default
            self.q_replay = (diag and self.with_q and not self.two_stage and n <= 640
                             and os.environ.get("JCDF_EIGH_Q_REPLAY") == "1")
            if self.two_stage or self.q_replay:
                self.side = torch.cuda.Stream(device=device)
                self.ev_fork = torch.cuda.Event()
                self.ev_join = torch.cuda.Event()
            self.dc_wb = int(self.lib.jcdf_stedc_workspace_bytes(n))
            if self.dc_wb < 0:
                raise OSError("jcdf_stedc_workspace_bytes(%d) = -1: too large for the library's divide & conquer" % n)
            self.dc_work = torch.zeros(self.dc_wb // 8 + 8, **f64)
            self.ok = True
        except Exception as e:                                    # library not loadable / size outside its kernels: plain torch path
            self.reason = repr(e)

    def __call__(self, Fp: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        """Fp: symmetric (n, n) device tensor.  Returns (eigenvalues ascending, U with
        eigenvectors in columns), like torch.linalg.eigh."""
        self.calls += 1
        self.U_padded = None                                      # set when U was produced zero padded (npad x npad) by the library
        if not self.ok:                                           # counted in `fallbacks` when the switch happened; see `reason`
            return torch.linalg.eigh(Fp)
        n, npad = self.n, self.npad
        st = torch.cuda.current_stream(self.device).cuda_stream
        p = lambda t: C.c_void_p(t.data_ptr())
        if self.two_stage or self.q_replay:
            self.A.copy_(Fp)
            return self._two_stage(Fp, st, p) if self.two_stage else self._one_stage_replay(Fp, st, p)
        # `scratch`: the caller hands over a work matrix it no longer needs (row stride = its leading dimension): tridiagonalised
        # in place, no copy; Q goes straight into the zero padded operand of the back-transformation GEMM
        in_place = self.scratch and Fp.stride(1) == 1
        if in_place:
            A, lda = Fp, Fp.stride(0)
        else:
            self.A.copy_(Fp)
            A, lda = self.A, n
        ev = self._stamp(None)
        rc = self.lib.jcdf_sytrd_q_device(C.c_void_p(st), n, p(A), lda, p(self.D), p(self.E), p(self.TAU),
                                          p(self.Qp) if self.with_q else None, npad, p(self.work), self.wb)
        ev = self._stamp(ev)
        if rc == 0:
            rc = self.lib.jcdf_stedc_device(C.c_void_p(st), n, p(self.D), p(self.E), p(self.Zt), npad, p(self.dc_work), self.dc_wb)
        ev = self._stamp(ev)
        if rc == 0 and self.with_q:
            # U[m][j] = sum_k Q[m][k] Z[k][j] = sum_k Qp[m][k] Zt[j][k]: the NT core; U comes back zero padded (npad x npad)
            rc = self.lib.jcdf_gemm_nt_device(C.c_void_p(st), npad, npad, npad, p(self.Qp), npad, p(self.Zt), npad, p(self.Up), npad)
        elif rc == 0:
            # Z <- Q Z from the stored reflectors (blocked compact-WY on the MFMA cores), U = its transpose-free copy in Up
            rc = self.lib.jcdf_ormtr_device(C.c_void_p(st), n, p(A), lda, p(self.TAU), p(self.Zt), npad, p(self.Up), npad,
                                            p(self.wy_work), self.wy_wb)
        if rc != 0:
            # the kernels of a stage may already have run on the caller's matrix (`scratch`): an eigensolve of what is left of it
            # would be silently wrong — that caller rebuilds its matrix and calls again (DeviceSCF: check() -> _diag)
            self.ok = False
            self.reason = "library call failed rc=%d" % rc
            self.fallbacks += 1
            if in_place:
                raise RuntimeError("DeviceEigh: %s after the caller's matrix was overwritten (scratch): rebuild it and call again" % self.reason)
            return torch.linalg.eigh(Fp)
        self.U_padded = self.Up
        return self.D, self.Up[:n, :n]

    def _stamp(self, ev):
        """timing only: one more device event on the current stream (ev = None starts a call's record)"""
        if not self.timing:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record(torch.cuda.current_stream(self.device))
        if ev is None:
            ev = []
            self.stage_events.append(ev)
        ev.append(e)
        return ev

    def stage_ms(self):
        """(tridiagonalisation ms, tridiagonal solver ms) averaged over the calls recorded while `timing`; clears the record.
        Call after a device synchronise."""
        recs = [r for r in self.stage_events if len(r) == 3]
        self.stage_events = []
        if not recs:
            return None, None
        return (sum(r[0].elapsed_time(r[1]) for r in recs) / len(recs), sum(r[1].elapsed_time(r[2]) for r in recs) / len(recs))

    def _two_stage(self, Fp, st, p):
        n, npad = self.n, self.npad
        main = torch.cuda.current_stream(self.device)
        # Q (stage 1) straight into the zero padded operand of the back-transformation GEMM
        rc = self.lib.jcdf_sytrd2_device(C.c_void_p(st), n, p(self.A), n, p(self.D), p(self.E), p(self.Qp), npad, p(self.work2),
                                         self.wb2)
        if rc == 0:
            self.ev_fork.record(main)
            self.side.wait_event(self.ev_fork)
            rc = self.lib.jcdf_sytrd2_apply_q_device(C.c_void_p(self.side.cuda_stream), n, p(self.Qp), npad, p(self.work2), self.wb2)
            self.ev_join.record(self.side)
        if rc == 0:
            rc = self.lib.jcdf_stedc_device(C.c_void_p(st), n, p(self.D), p(self.E), p(self.Zt), npad, p(self.dc_work), self.dc_wb)
        main.wait_event(self.ev_join)
        if rc == 0:
            rc = self.lib.jcdf_gemm_nt_device(C.c_void_p(st), npad, npad, npad, p(self.Qp), npad, p(self.Zt), npad, p(self.Up), npad)
        if rc != 0:
            self.ok = False
            self.reason = "library call failed rc=%d (two-stage reduction)" % rc
            self.fallbacks += 1
            return torch.linalg.eigh(Fp)
        self.U_padded = self.Up
        return self.D, self.Up[:n, :n]

    def _one_stage_replay(self, Fp, st, p):
        n, npad = self.n, self.npad
        main = torch.cuda.current_stream(self.device)
        rc = self.lib.jcdf_sytrd_q_device(C.c_void_p(st), n, p(self.A), n, p(self.D), p(self.E), p(self.TAU), None, 0, p(self.work), self.wb)
        if rc == 0:
            self.ev_fork.record(main)
            self.side.wait_event(self.ev_fork)
            rc = self.lib.jcdf_sytrd_replay_q_device(C.c_void_p(self.side.cuda_stream), n, p(self.A), n, p(self.TAU), p(self.Qp), npad)
            self.ev_join.record(self.side)
        if rc == 0:
            rc = self.lib.jcdf_stedc_device(C.c_void_p(st), n, p(self.D), p(self.E), p(self.Zt), npad, p(self.dc_work), self.dc_wb)
        main.wait_event(self.ev_join)
        if rc == 0:
            rc = self.lib.jcdf_gemm_nt_device(C.c_void_p(st), npad, npad, npad, p(self.Qp), npad, p(self.Zt), npad, p(self.Up), npad)
        if rc != 0:
            self.ok = False
            self.reason = "library call failed rc=%d" % rc
            self.fallbacks += 1
            return torch.linalg.eigh(Fp)
        self.U_padded = self.Up
        return self.D, self.Up[:n, :n]

    def _err_word(self) -> torch.Tensor:
        """1-element int32 view of the error word of the reduction in use"""
        w = self.work2 if self.two_stage else self.work
        return w[1:2].view(torch.int32)[0:1]

    def _dc_info(self) -> torch.Tensor:
        """1-element int32 view of the divide & conquer's info word (byte 0 of its workspace: a leaf that did not converge)"""
        return self.dc_work[0:1].view(torch.int32)[0:1]

    def status(self) -> torch.Tensor:
        """0-d float64 device tensor, non-zero iff the last call failed (hand-off timeout in the tridiagonalisation or a
        tridiagonal leaf that did not converge) — for callers that fold it into a device-to-host copy they make anyway."""
        if not self.ok:
            return torch.zeros((), dtype=torch.float64, device=self.device)
        return (self._err_word()[0].abs() + self._dc_info()[0].abs()).to(torch.float64)

    def status_tensors(self):
        """(int32 error word of the tridiagonalisation, int32 info word of the divide & conquer) as 1-element device views, or
        (None, None) when the plain torch path is in use — for `jcdf_scf_tail_device`."""
        if not self.ok:
            return None, None
        return self._err_word(), self._dc_info()

    def check(self) -> bool:
        """True if the persistent kernel's hand-offs never timed out and the divide & conquer converged
        (reads two words from the device: call where the stream is synchronised anyway)."""
        if not self.ok:
            return True
        err = int(self._err_word()[0].item())                       # int at byte offset 8
        info = int(self._dc_info()[0].item())
        bad = err != 0 or info != 0
        if bad:
            self.ok = False
            self.fallbacks += 1
            self.reason = "sytrd hand-off timeout" if err else "stedc info=%d" % info
        return not bad


class DeviceSP2:
    """The projector onto the n_occ lowest eigenvectors of a symmetric matrix without an eigensolve
    (`jcdf_sp2_device`, csrc/jcdf_sp2.hpp: trace-correcting spectral projection, matrix squarings on MFMA).
    Optional replacement of the per-iteration eigensolve in DeviceSCF (scf flag density_solver = "sp2"): the SCF
    energy, density and DIIS error depend on the occupied SPACE only (SCF.jl:1072-1125 forms D = 2 C_o C_o^T).
    No host round trip: `iterations` squarings are enqueued (adapted from the count the previous call needed),
    `info` stays on the device for the caller's one copy per SCF iteration.  P comes back zero padded (npad x npad,
    npad = n rounded up to 32): the operand shape of the library's GEMM cores."""

    def __init__(self, n: int, n_occ: int, device: torch.device):
        self.n, self.n_occ, self.device = n, n_occ, device
        self.lib = _lib.load()
        f64 = dict(dtype=torch.float64, device=device)
        self.wb = int(self.lib.jcdf_sp2_workspace_bytes(n))
        self.work = torch.zeros(self.wb // 8 + 8, **f64)
        self.npad = (n + 31) // 32 * 32
        self.Pp = torch.zeros((self.npad, self.npad), **f64)
        self.P = self.Pp[:n, :n]
        self.info = torch.zeros(8, **f64)
        self.iterations = 72
        self.calls = 0

    def __call__(self, Fp: torch.Tensor, iterations: Optional[int] = None, ref=None) -> torch.Tensor:
        """Fp: symmetric (n, n) device tensor or view (row stride = its leading dimension).  Returns the (n, n) view of Pp.
        ref = (F_ref, eigs): a matrix diagonalised before (same shape, unit column stride) and a device tensor with four of its
        eigenvalues {lowest, HOMO, LUMO, highest} — spectral bounds by Weyl's inequality from ||Fp - F_ref||_F and, while they
        bracket the gap, the accelerated recursion (`jcdf_sp2_ref_device`: about half the squarings, same projector)."""
        self.calls += 1
        st = torch.cuda.current_stream(self.device).cuda_stream
        if Fp.stride(1) != 1:
            Fp = Fp.contiguous()
        Fref, eigs = ref if ref is not None else (None, None)
        rc = self.lib.jcdf_sp2_ref_device(C.c_void_p(st), self.n, self.n_occ, C.c_void_p(Fp.data_ptr()), Fp.stride(0),
                                          C.c_void_p(self.Pp.data_ptr()), self.npad, int(iterations or self.iterations),
                                          C.c_void_p(self.work.data_ptr()), self.wb, C.c_void_p(self.info.data_ptr()),
                                          C.c_void_p(Fref.data_ptr()) if Fref is not None else None, Fref.stride(0) if Fref is not None else 0,
                                          C.c_void_p(eigs.data_ptr()) if eigs is not None else None)
        if rc != 0:
            raise RuntimeError("jcdf_sp2_ref_device failed (status %d)" % rc)
        return self.P

    def adapt(self, used: float, finished: bool) -> None:
        """after the caller has read info: enqueue (needed + 4) squarings next time, at least 16, more after a miss"""
        self.iterations = min(400, max(16, int(used) + 4)) if finished else min(400, 2 * self.iterations)


class DeviceLowdin:
    """Orthonormal basis of the span of o row vectors (`jcdf_lowdin_rows_device`, csrc/jcdf_blas.hpp: Loewdin
    orthonormalisation by Newton-Schulz, products on the MFMA cores) for the SP2 step of DeviceSCF: Y = (P Cp)^T holds the old
    occupied orbitals projected into the new occupied space.  Buffers are (op x npad) zero padded.  `steps` Newton-Schulz steps
    are enqueued (adapted from what the previous call needed); info[1] (device) = steps needed, 0 = not converged."""

    def __init__(self, o: int, n: int, device: torch.device):
        self.o, self.n, self.device = o, n, device
        self.lib = _lib.load()
        f64 = dict(dtype=torch.float64, device=device)
        self.op, self.npad = (o + 31) // 32 * 32, (n + 31) // 32 * 32
        self.wb = int(self.lib.jcdf_lowdin_workspace_bytes(o))
        self.work = torch.zeros(self.wb // 8 + 8, **f64)
        self.info = torch.zeros(4, **f64)
        self.steps = 12

    def __call__(self, Yp: torch.Tensor, out: torch.Tensor) -> None:
        assert Yp.shape == (self.op, self.npad) and out.shape == (self.op, self.npad) and Yp.is_contiguous() and out.is_contiguous()
        st = torch.cuda.current_stream(self.device).cuda_stream
        rc = self.lib.jcdf_lowdin_rows_device(C.c_void_p(st), self.o, self.n, C.c_void_p(Yp.data_ptr()), self.npad, C.c_void_p(out.data_ptr()),
                                              self.npad, int(self.steps), C.c_void_p(self.work.data_ptr()), self.wb, C.c_void_p(self.info.data_ptr()))
        if rc != 0:
            raise RuntimeError("jcdf_lowdin_rows_device failed (status %d)" % rc)

    def adapt(self, needed: float) -> None:
        """needed: info[1] as read by the caller (0: did not converge in the steps enqueued)"""
        self.steps = min(40, max(3, int(needed) + 2)) if needed >= 1 else min(40, 2 * self.steps)
