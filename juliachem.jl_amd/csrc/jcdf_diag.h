/* jcdf_diag.h - entry points that exist ONLY in diagnostic builds of the library (-DJCDF_DIAGNOSTIC, tools/build_diag.sh ->
 * tools/_build/libjcdf_hip_diag.so, selected with JCDF_LIB_PATH).  Not part of the product ABI (include/jcdf.h): experiment
 * kernels, optional paths that were measured at parity and not adopted (the two-stage tridiagonalisation of csrc/jcdf_sbr.hpp,
 * the Q replay), and in-kernel cycle stamps.  Diagnostic builds also read the JCDF_* variant environment variables. */
#ifndef JCDF_DIAG_H
#define JCDF_DIAG_H
#include "../../include/jcdf.h"
#ifdef __cplusplus
extern "C" {
#endif

/* Q of A = Q T Q^T rebuilt from the reflectors jcdf_sytrd_device left in d_A / d_TAU (LAPACK dorgtr's matrix, row-major,
 * leading dimension ldq), row-parallel; n <= 640.  It needs neither D nor E: the caller runs it on a second stream beside
 * jcdf_stedc_device, so that the persistent kernel spends its hand-off window on the rank-2 update instead of on Q. */
int32_t jcdf_sytrd_replay_q_device(void *stream, int64_t n, const double *d_A, int64_t lda, const double *d_TAU,
                                   double *d_Q, int64_t ldq);

/* The same reduction in two stages (csrc/jcdf_sbr.hpp): dense -> band of half-width 16 (one Householder QR per panel of
 * 16 columns inside one workgroup + MFMA block-reflector updates) -> tridiagonal (bulge chasing in the LDS of one
 * workgroup) — 3 n/16 kernel boundaries instead of n chip-wide hand-offs.  jcdf_sytrd2_device leaves D, E (device, n and
 * n-1), the stage-1 orthogonal factor in d_Q (n x n row-major, leading dimension ldq) and the stage-2 reflectors in
 * d_work; jcdf_sytrd2_apply_q_device (any stream ordered behind the first call; it needs neither D nor E, so it may run
 * beside jcdf_stedc_device) completes d_Q to the Q of A = Q T Q^T.  d_A (symmetric, fully stored) is overwritten.
 * The int at byte offset 8 of d_work is non-zero afterwards if a wait inside the chase gave up (result invalid).
 * n <= jcdf_sytrd2_max_n() (the band must fit the LDS of one CU: 590).  The stage-1 factor is accumulated on an internal
 * per-device side stream beside the chase and joined into `stream` before the call returns: one call at a time per device. */
int64_t jcdf_sytrd2_max_n(void);
int64_t jcdf_sytrd2_workspace_bytes(int64_t n);
int32_t jcdf_sytrd2_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E, double *d_Q,
                           int64_t ldq, void *d_work, int64_t work_bytes);
int32_t jcdf_sytrd2_apply_q_device(void *stream, int64_t n, double *d_Q, int64_t ldq, const void *d_work,
                                   int64_t work_bytes);

/* Experiment helper (tools/gap_test3.py; not on any product path): `workgroups` workgroups of `threads` (64 / 128 / 256)
 * threads that stay on the device for `microseconds` — mode 0: sleeping waves only, 1: a dependent fp64 FMA chain per wave,
 * 2: fp64 MFMAs, with `pause` x 64 clocks of s_sleep between two bursts — or until *d_stop != 0 (d_stop may be NULL).
 * d_sink: one double of device memory.  Answers what the clock governor looks at when the shader clock drops during the
 * replicated eigensolve (profiles/r02_clock_gap.txt). */
int32_t jcdf_keepalive_device(void *stream, int32_t workgroups, int32_t threads, double microseconds, int32_t mode,
                              int32_t pause, const int32_t *d_stop, double *d_sink);

/* Diagnostic only (environment JCDF_W_ABLATE=32 with JCDF_W_REM=0 at jcdf_configure, 81..96 occupied orbitals): shader
 * cycles per wave spent in the five segments of the W kernel's phases during the last build — DMA issue, operand reads +
 * MFMA issue, index loads / epilogue, counted vmcnt wait, barrier — and the number of phases; 6 words per wave. */
int64_t jcdf_w_stall_cycles(jcdf_handle *h, unsigned long long *out, int64_t max_waves);

#ifdef __cplusplus
}
#endif
#endif /* JCDF_DIAG_H */
