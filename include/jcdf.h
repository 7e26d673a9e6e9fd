/* jcdf.h — C ABI of libjcdf_hip.so: MI355X (gfx950) density-fitted RHF Fock build.
 *
 * Drop-in boundary for ONE path of JuliaChem.jl: the GPU density-fitted Fock
 * build that `df_rhf_fock_build!` dispatches to
 *   /root/reference/src/rhf/energy/DensityFitting/DensityFitting.jl:51-52,78-90
 *   -> df_rhf_fock_build_GPU!        (GPUDF.jl:11-304)
 *   -> df_rhf_fock_build_dense_GPU!  (DenseGPUDF.jl:9-162)
 * Each entry point below names the reference code it replaces.  INTEGRATION.md
 * shows the Julia `ccall` stubs a maintainer adds (julia/JCDFHip.jl).
 *
 * Conventions
 *   - plain C: pointers + int64 sizes, no C++/torch types.  Every function
 *     returns int32 status (0 = ok, != 0 see jcdf_status); the message is read
 *     with jcdf_last_error().  No exception crosses the boundary.
 *   - host matrices are Julia arrays: column-major fp64, first index fastest.
 *     Index arrays are int64 and 0-BASED on this side (the glue subtracts 1).
 *   - the caller owns every host pointer; the library copies before returning
 *     and never retains a host pointer (Julia GC may move/free it afterwards).
 *   - one handle == one HIP device == one auxiliary-index shard (the reference's
 *     "global device id", GPUDF.jl:1026-1056).  Handles are independent; the
 *     per-shard Fock matrices are summed either by a group of handles — all devices of one
 *     process, summed ON THE DEVICES over RCCL / peer-mapped buffers: the end of this
 *     header — or by the caller (reference: host axpy GPUDF.jl:267-277 and
 *     MPI.Allreduce! DensityFitting.jl:68-71; one process per GPU: RCCL all-reduce on
 *     the device buffer returned by jcdf_fock_build_device).
 *   - calls on one handle must come from one host thread at a time.
 */
#ifndef JCDF_H
#define JCDF_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jcdf_handle jcdf_handle;

enum jcdf_status {
    JCDF_OK = 0,
    JCDF_ERR_INVALID = 1,     /* bad argument / call order                      */
    JCDF_ERR_HIP = 2,         /* a HIP runtime call failed                      */
    JCDF_ERR_NO_DEVICE = 3,   /* no usable gfx950 device (product path has NO CPU fallback) */
    JCDF_ERR_ALLOC = 4,
    JCDF_ERR_NOT_SPD = 5,     /* metric (P|Q) not positive definite             */
    JCDF_ERR_INTERNAL = 6
};

/* Per-call device timings in seconds, measured with HIP events on the handle's
 * stream.  Field names follow the reference's JCTiming keys
 * (shared/JCTiming.jl:52-105: "GPU_-N-_W_time-", "GPU_-N-_K_time-", ...;
 * written at GPUDF.jl:280-301 / DenseGPUDF.jl:139-160). */
typedef struct jcdf_timings {
    double non_zero_coeff_time; /* C_occ staging (replaces build_non_zero_coefficients_kernel, GPUDF.jl:459-480) */
    double W_time;              /* exchange intermediate W (+ fused V)   GPUDF.jl:637-667 / DenseGPUDF.jl:107 */
    double K_time;              /* K = W^T W                             GPUDF.jl:669-826 / DenseGPUDF.jl:111 */
    double V_time;              /* 0: V is produced inside the W pass    GPUDF.jl:539-542 */
    double J_time;              /* J = 2 B^T V                           GPUDF.jl:544-547 */
    double density_time;        /* 0 unless the density is requested     GPUDF.jl:327-345 */
    double H_add_time;          /* folded into assemble                  GPUDF.jl:221-225 */
    double copy_J_time;         /* assemble: J scatter + symmetrise + H  GPUDF.jl:482-536 */
    double fock_time;           /* whole device Fock build               "GPU_-N-_fock_time-" */
    double copy_time;           /* H2D C_occ + D2H F (host entry only)   GPUDF.jl:206,267-277 */
} jcdf_timings;

/* Kernel launch records of the most recent jcdf_fock_build*: one entry per
 * kernel launched (name, average seconds, algorithmic flops and bytes).  Used by
 * bench.py for the roofline object. */
typedef struct jcdf_kernel_stat {
    char name[48];
    double seconds;
    double flops;      /* executed fp64 flops (incl. padding)   */
    double alg_flops;  /* algorithmic flops (SURVEY 8d formula) */
    double alg_bytes;  /* algorithmic HBM bytes                 */
} jcdf_kernel_stat;

/* ---- lifetime -------------------------------------------------------------- */
/* Replaces get_default_gpu_data_cuda() (shared/GPUData_cuda.jl:40-46) + the
 * CUDA.device!() selection (GPUDF.jl:192).  Fails with JCDF_ERR_NO_DEVICE when
 * no HIP device exists — there is no CPU fallback. */
int32_t jcdf_create(jcdf_handle **out, int32_t device_id);
/* Replaces CUDA.unsafe_free!/GC of the device arrays (GPUDF.jl:1001-1006). */
int32_t jcdf_destroy(jcdf_handle *h);
/* Message of the last failing call on `h` (h == NULL: last jcdf_create failure). */
const char *jcdf_last_error(const jcdf_handle *h);
/* Stream every later call on `h` enqueues its copies and kernels on.  use_own != 0:
 * the handle's own non-blocking stream (the default after jcdf_create);
 * otherwise `stream` is a hipStream_t of the handle's device, NULL meaning the HIP
 * legacy default stream.  A host that produces device inputs on its own stream
 * (e.g. PyTorch's current stream) sets that stream here so everything is ordered. */
int32_t jcdf_set_stream(jcdf_handle *h, void *stream, int32_t use_own);
/* ABI version of this header (major*1000 + minor). */
int32_t jcdf_abi_version(void);

/* ---- setup (iteration == 1 branch, GPUDF.jl:37-165 / DenseGPUDF.jl:29-72) ---- */
/* Problem sizes + the packed pq layout + this handle's aux shard.
 *   N        scf_data.mu  (# AO)                    DensityFitting.jl:41
 *   Q_total  scf_data.A   (# aux, all shards)       DensityFitting.jl:42
 *   q0,q1    this shard's aux function range [q0,q1), 0-based
 *            (static_load_rank_indicies, DynamicLoad.jl:174-203)
 *   n_occ    scf_data.occ                            DensityFitting.jl:43
 *   P        screening_data.screened_indices_count  (ScreenedDF.jl:28)
 *   pq_p,pq_q  for packed index c: outer index p and inner index q, i.e. the
 *            inverse of sparse_pq_index_map[q,p] (SchwarzScreening.jl:72-81;
 *            reference builds it on device at GPUDF.jl:422-438).  Must be a
 *            symmetric set: (q,p) kept <=> (p,q) kept.  NULL,NULL with
 *            P == N*N selects the unscreened map c = q + N*p
 *            (SchwarzScreening.jl:97-111) == the dense GPU path. */
int32_t jcdf_configure(jcdf_handle *h, int64_t N, int64_t Q_total, int64_t q0, int64_t q1,
                       int64_t n_occ, int64_t P, const int64_t *pq_p, const int64_t *pq_q);

/* Tuning knobs a caller may set (the library reads NO environment variable).  Call before jcdf_configure; values persist
 * across jcdf_configure; 0 restores the library's own rule.  Keys:
 *   "k_slices_per_xcd"  split-K slices of the exchange-K SYRK per XCD (the reference's df_exchange_n_blocks plays this role,
 *                       GPUDF.jl:61-72; results do not depend on it beyond summation order)
 *   "w_chunk_stages"    contraction stages per workgroup chunk of the exchange-W kernel
 *   "host_cholesky"     1: factor the metric with the library's host potrf/trtri instead of on the device
 *   "j_workgroups"      workgroups of the Coulomb pass while it runs beside the exchange-K pass (library rule: one per CU
 *                       while K is the longer kernel; results do not depend on it)
 *   "k_first"           1: enqueue K in front of J in that phase (library rule: J first)
 * JCDF_ERR_INVALID for an unknown key or a value out of range. */
int32_t jcdf_set_tuning(jcdf_handle *h, const char *key, int64_t value);

/* Exchange screening, the reference's scf flag df_exchange_screen (SCFOptions.jl:92-93; off by default there and here):
 * calculate_exchange_block_screen_matrix + calculate_K_lower_diagonal_block (ScreenedDF.jl:385-457, 459-545).  K is cut
 * into n_blocks x n_blocks blocks of width N / n_blocks (N < 100: one block; the reference's df_exchange_n_blocks, CPU
 * default 10); a block of the lower triangle is computed only if it holds a kept pair of the packed pq map, K = 0 in the
 * others (what the reference's skipped blocks hold on a fresh Fock array); the ragged strip beyond n_blocks * width is
 * always computed.  Here: a 64 x 64 block of the K kernel is launched only if it overlaps a kept reference block, and
 * elements of screened reference blocks are zeroed in the assemble kernel.  n_blocks = 0: off.  Call before
 * jcdf_configure (it persists across jcdf_configure). */
int32_t jcdf_set_exchange_screening(jcdf_handle *h, int64_t n_blocks);

/* Metric.  `J2c` = two_center_integrals (Q_total x Q_total column-major, only the
 * lower triangle is read, TwoCenterIntegrals.jl:7-29).  Performs potrf('L') +
 * trtri('L','N') ON THE DEVICE (blocked fp64-MFMA factorisation, csrc/jcdf_chol.hpp —
 * the placement of CUSOLVER.potrf!/trtri! at DenseGPUDF.jl:185-193; the screened
 * path does it with host LAPACK at GPUDF.jl:890-891) and keeps rows [q0,q1) of
 * L^-1.  JCDF_ERR_NOT_SPD on a non-positive pivot.  jcdf_set_tuning(h, "host_cholesky", 1)
 * selects the library's host potrf/trtri instead (the reference's GPUDF.jl:890-891 placement). */
int32_t jcdf_set_metric(jcdf_handle *h, const double *J2c);
/* Same, when the caller already holds L^-1 (Q_total x Q_total, lower
 * triangular, upper = 0) — what GPUDF.jl:893-902 uploads / broadcasts. */
int32_t jcdf_set_metric_inverse(jcdf_handle *h, const double *Linv);

/* Push the three-centre integrals of aux rows [s0,s1) (global, 0-based):
 * T is (s1-s0) x P, column-major (ThreeCenterIntegralsScreened.jl:8-85 /
 * ThreeCenterIntegrals.jl:9-42 with the dense map).  The library accumulates
 *   B[q0:q1, :] += Linv[q0:q1, s0:s1] * T
 * on the device (trmm/gemm of GPUDF.jl:907,939-943; DenseGPUDF.jl:210,270).
 * Call once per row block, in any order, covering [0,Q_total) (blocks with
 * s0 >= q1 contribute zero and may be skipped: L^-1 is lower triangular).
 * The first push zeroes B. */
int32_t jcdf_push_three_center(jcdf_handle *h, int64_t s0, int64_t s1, const double *T);
/* Same with T already on this handle's device (e.g. received over RCCL). */
int32_t jcdf_push_three_center_device(jcdf_handle *h, int64_t s0, int64_t s1, const double *d_T);
/* Alternative to set_metric + push: the caller already holds this shard of
 * B = L^-1 T ((q1-q0) x P column-major), e.g. from the reference's CPU path
 * (ScreenedDF.jl:103). */
int32_t jcdf_set_B(jcdf_handle *h, const double *B);
/* The same for packed columns [c0,c1) of that matrix, already on this handle's device ((q1-q0) x (c1-c0)
 * column-major): B formed elsewhere on the device, or a tensor too large to stage on the host (a 154 GB B is set
 * in column blocks).  The caller covers [0,P). */
int32_t jcdf_set_B_columns_device(jcdf_handle *h, int64_t c0, int64_t c1, const double *d_B);
/* Read back this shard of B in the reference layout ((q1-q0) x P col-major). */
int32_t jcdf_get_B(jcdf_handle *h, double *B_out);

/* Core Hamiltonian, N x N.  NULL: this shard adds no H (reference adds H on
 * rank 0 / device 1 only: GPUDF.jl:221-225, DenseGPUDF.jl:114-119). */
int32_t jcdf_set_core_hamiltonian(jcdf_handle *h, const double *H);

/* ---- per SCF iteration (GPUDF.jl:188-277 / DenseGPUDF.jl:83-137) ------------- */
/* C_occ: N x n_occ column-major (= coefficients[:,1:occ], DensityFitting.jl:49).
 * F_out: N x N host buffer, fully overwritten with this shard's
 * 2J - K (+ H if set) — the contract of scf_data.two_electron_fock
 * (DensityFitting.jl:62-75).  t may be NULL. */
int32_t jcdf_fock_build(jcdf_handle *h, const double *C_occ, double *F_out, jcdf_timings *t);
/* The same in two halves, so that a host driving several handles (num_devices > 1,
 * one Julia task per device at GPUDF.jl:188-193) overlaps the devices from one thread:
 * begin() copies C_occ and enqueues the kernels and returns; finish() copies F back and
 * synchronises.  jcdf_fock_build == begin + finish. */
int32_t jcdf_fock_build_begin(jcdf_handle *h, const double *C_occ);
int32_t jcdf_fock_build_finish(jcdf_handle *h, double *F_out, jcdf_timings *t);
/* Same with device pointers on this handle's device.  Work is enqueued on
 * `stream` (a hipStream_t; NULL = the handle's stream, see jcdf_set_stream) and is
 * NOT synchronised on return, so the caller can chain an RCCL all-reduce of d_F. */
int32_t jcdf_fock_build_device(jcdf_handle *h, const double *d_C_occ, double *d_F, void *stream);
/* The same with leading dimensions: orbital i of d_C_occ starts at d_C_occ + ldc * i, column p of d_F at d_F + ldf * p
 * (ldc, ldf >= N) — a device-resident caller that keeps its matrices zero padded for its own GEMMs passes them as they are
 * (no repacking kernels between the eigensolver, the Fock build and the DIIS products).  Only the N x N part of d_F is written. */
int32_t jcdf_fock_build_device_ld(jcdf_handle *h, const double *d_C_occ, int64_t ldc, double *d_F, int64_t ldf, void *stream);
/* The Coulomb pass (streams half of B, no MFMA) and the exchange K pass (MFMA, W out of L2) both depend only on
 * the W pass: by default J is enqueued on an internal side stream and runs BESIDE K (forked and joined with events
 * on the build's stream; 3.3 -> 3.1 ms per build on the C20H42 shape).  overlap_jk = 0 runs them one after the
 * other — then J_time / K_time of jcdf_timings are the stand-alone durations the rooflines are quoted on; with the
 * overlap J_time is the (longer) time J takes while it shares the device. */
int32_t jcdf_set_overlap(jcdf_handle *h, int32_t overlap_jk);
/* Blocks until work enqueued by the previous call has finished; fills timings. */
int32_t jcdf_synchronize(jcdf_handle *h, jcdf_timings *t);

/* Intermediates of the last build, for parity tests (reference names):
 * V = device_coulomb_intermediate (length q1-q0); W = device_exchange_intermediate
 * in the reference GPU layout (q1-q0, n_occ, N) column-major (GPUDF.jl:140). */
int32_t jcdf_get_V(jcdf_handle *h, double *V_out);
int32_t jcdf_get_W(jcdf_handle *h, double *W_out);

/* Host utility used by jcdf_set_metric, exported for testing without a GPU:
 * in-place lower Cholesky + triangular inverse of the n x n column-major matrix
 * A (LAPACK.potrf!('L') + trtri!('L','N'), GPUDF.jl:890-891).  Returns 0, or the
 * 1-based index of the first non-positive pivot.  Upper triangle is zeroed. */
int32_t jcdf_host_potrf_trtri(double *A, int64_t n);
/* The device factorisation jcdf_set_metric runs, exported with the same contract for
 * parity tests: A (host, n x n column-major, lower triangle read) is overwritten with
 * L^-1 (lower triangular, upper = 0).  Returns a jcdf_status (JCDF_ERR_NOT_SPD on a
 * non-positive pivot). */
int32_t jcdf_device_potrf_trtri(int32_t device_id, double *A, int64_t n);

/* Caller-side helper for the replicated eigensolve of the SCF iteration (reference: host
 * LAPACK eigen!(Hermitian(.)) at src/rhf/energy/SCF.jl:1083): Householder tridiagonalisation
 * (LAPACK dsytrd 'L' semantics: D, E, TAU and reflectors below the sub-diagonal of A, column-
 * major) of the symmetric n x n device matrix d_A on `stream`: ONE persistent chip-wide kernel for
 * the columns 0 .. n-129 and a one-workgroup kernel for the last 128 (a matrix of 32 .. 128 rows goes
 * to the latter whole).  The upper triangle of d_A is scratch.
 * d_work: jcdf_sytrd_workspace_bytes(n) bytes of device memory; its int at byte offset 8 is
 * non-zero afterwards if the in-kernel grid barrier timed out (result invalid).  n <~ 2200. */
int64_t jcdf_sytrd_workspace_bytes(int64_t n);
int32_t jcdf_sytrd_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E,
                          double *d_TAU, void *d_work, int64_t work_bytes);
/* Same, and additionally d_Q (n x n, row-major == the transpose in column-major) receives the
 * orthogonal matrix Q = H_0 H_1 ... of A = Q T Q^T (row stride ldq >= n, so that it can be written straight into a zero
 * padded GEMM operand), accumulated inside the chip-wide kernel while the
 * reflectors travel between workgroups (the last 128 reflectors by one row-parallel launch), so the
 * eigenvectors of A are ONE GEMM Q*Z away (instead of LAPACK's dormtr back-transformation).  d_Q may be NULL (== jcdf_sytrd_device).
 * jcdf_sytrd_max_n(with_q): largest n whose working set fits the LDS of the device (JCDF_ERR_INVALID above):
 * 1536 with Q (the one-exchange kernel: rows of Q in registers), ~2040 without (there the first n - 1536 columns go through
 * the two-exchange kernel and the trailing 1536 x 1536 block through the one-exchange kernel; the eigenvectors then take
 * jcdf_ormtr_device instead of the GEMM with Q). */
int32_t jcdf_sytrd_q_device(void *stream, int64_t n, double *d_A, int64_t lda, double *d_D, double *d_E,
                            double *d_TAU, double *d_Q, int64_t ldq, void *d_work, int64_t work_bytes);
int64_t jcdf_sytrd_max_n(int32_t with_q);
/* The tridiagonalisation kernels are PERSISTENT: their G <= 256 workgroups hand columns to each other and must all be resident
 * at once.  Before every such launch the library checks hipOccupancyMaxActiveBlocksPerMultiprocessor x CUs >= G for the
 * kernel at its block size and LDS (JCDF_ERR_INVALID otherwise: refused, not discovered by a spin timeout).  The launch
 * itself: mode 2 = always hipLaunchCooperativeKernel; mode 1 (default) = cooperative when G exceeds half the CUs — two such
 * kernels (another stream, another rank on the card) could not be resident together and must queue up rather than each
 * hold part of the chip — and a plain launch for the smaller grids that fit side by side (the cooperative launch costs
 * ~50 us per eigensolve); mode 0 = never cooperative.  The bounded spins inside the kernels (50 ms, error word of
 * jcdf_sytrd_workspace_bytes) stay as the second line.  Process-wide. */
int32_t jcdf_set_persistent_launch_mode(int32_t mode);
/* Back-transformation of the same eigensolve for the sizes above jcdf_sytrd_max_n(1), where Q is not accumulated in the
 * tridiagonalisation (LAPACK dormtr('L','L','N') semantics, the third stage of dsyevd behind SCF.jl:1083): C <- Q C with
 * Q = H_0 H_1 ... from the reflectors and d_TAU that jcdf_sytrd_device left in d_A, by blocked compact-WY on the library's
 * fp64 MFMA cores (csrc/jcdf_wy.hpp; no vendor kernel).  d_Ct holds C TRANSPOSED — row j = column j of C, contiguous, what
 * jcdf_stedc_device writes with ldz = ldc — zero padded to roundup(n, 32) rows and columns (ldc >= that, even); it is
 * updated in place.  d_Out (optional): the result once more as a row-major matrix [component][vector] with leading dimension
 * ldo >= roundup(n, 32), zero padded (the operand shape of jcdf_gemm_tn_device).  d_work: jcdf_ormtr_workspace_bytes(n). */
int64_t jcdf_ormtr_workspace_bytes(int64_t n);
int32_t jcdf_ormtr_device(void *stream, int64_t n, const double *d_A, int64_t lda, const double *d_TAU, double *d_Ct, int64_t ldc,
                          double *d_Out, int64_t ldo, void *d_work, int64_t work_bytes);
/* The Pulay (DIIS) step of the SCF wrapper on the device, so that the iteration needs no round trip to the host
 * between the Fock build and the eigensolve (reference: DIIS, EnergyHelpers.jl:234-258, called at SCF.jl:472-501):
 * d_Bmat nd x nd ring buffer of error-vector dot products (row and column `head` are first overwritten with
 * d_dots[nd]); n vectors in use, newest first slot_k = (head - k) mod nd; solve != 0: the bordered (n+1) system is
 * solved and d_coef[slot] receives the extrapolation coefficients (0 for unused slots); solve == 0 or a singular /
 * non-finite system (then d_flag[0] = 1, the reference's "Faulty DIIS" path): unit vector on the newest entry.  nd <= 15. */
int32_t jcdf_diis_device(void *stream, int32_t nd, int32_t head, int32_t n, int32_t solve, double *d_Bmat,
                         const double *d_dots, double *d_coef, int32_t *d_flag);
/* Second stage of the same eigensolve: all eigenvalues and eigenvectors of the symmetric tridiagonal
 * matrix (d_D diagonal, d_E sub-diagonal, both device, length n and n-1) by divide & conquer
 * (LAPACK dstedc 'I' semantics; csrc/jcdf_dc.hpp).  On return (stream-ordered) d_D holds the eigenvalues
 * ascending and d_Z (n x n column-major, leading dimension ldz) the eigenvectors in its columns; d_E is
 * unchanged.  d_work: jcdf_stedc_workspace_bytes(n) bytes of device memory (-1: n is too large for this solver, ~2700 rows:
 * known before anything is enqueued); its int at byte offset 0 is non-zero afterwards if a leaf's QL iteration did not
 * converge (non-finite input does this): the decomposition must then not be used. */
int64_t jcdf_stedc_workspace_bytes(int64_t n);
int32_t jcdf_stedc_device(void *stream, int64_t n, double *d_D, double *d_E, double *d_Z, int64_t ldz,
                          void *d_work, int64_t work_bytes);

/* Scalar tail of one SCF iteration, caller-side helper of the device SCF loop: E_elec = 1/2 sum D o (F + H)
 * (SCF.jl:1116-1123) and ||D - D_old||_F from the four n x n device matrices, packed with the iteration's status words
 * into d_out (8 doubles, device): {E_elec, ||dD||, *d_diis_flag, |*d_eig_err| + |*d_eig_info|, d_sp2_info[1],
 * d_sp2_info[2], *d_pivot, d_sp2_info[0]} — any of the five status pointers may be NULL (read as 0).  Sums over a fixed
 * partition in a fixed order (bit-reproducible).  d_work: 256 doubles of device memory. */
int32_t jcdf_scf_tail_device(void *stream, int64_t n, const double *d_D, const double *d_D_old, const double *d_F, const double *d_H,
                             const int32_t *d_diis_flag, const int32_t *d_eig_err, const int32_t *d_eig_info,
                             const double *d_sp2_info, const double *d_pivot, double *d_work, double *d_out);

/* Orthonormal basis of the span of o row vectors, caller-side helper of the SP2 step (the old occupied orbitals
 * projected into the new occupied space, Y = (P C_prev)^T), for any number of rows (o <= 4096), without a factorisation: Loewdin (symmetric) orthonormalisation by the
 * coupled Newton-Schulz iteration for G^{-1/2}, G = Y Y^T — o x o products on the fp64 MFMA cores only (csrc/jcdf_blas.hpp).
 * d_Y: o rows of length n, row-major with leading dimension ldy, ZERO PADDED to roundup(o, 32) rows and roundup(n, 32)
 * columns (ldy, ldz >= that, even); d_Z receives Z = G^{-1/2} Y in the same padded shape (rows >= o come out zero).
 * `iterations` (1..40) steps are enqueued; d_info (4 doubles, device) = {||I - G||_F, steps that were needed (0: the
 * iteration had not converged to 2e-7 before its last step: reject the result), ||I - Z Y||_F before the last step,
 * steps run}.  The eigenvalues of G must lie in (0, ~1.6) (orthonormal vectors times a projector: (0, 1]); a row set that
 * has lost rank never converges and is reported through info[1] = 0.  d_work: jcdf_lowdin_workspace_bytes(o) bytes. */
int64_t jcdf_lowdin_workspace_bytes(int64_t o);
int32_t jcdf_lowdin_rows_device(void *stream, int64_t o, int64_t n, const double *d_Y, int64_t ldy, double *d_Z, int64_t ldz,
                                int32_t iterations, void *d_work, int64_t work_bytes, double *d_info);

/* Alternative to the eigensolve inside the SCF step (optional; SCF.jl:1072-1125 takes the density from eigen()):
 * the spectral projector P onto the n_occ lowest eigenvectors of the symmetric matrix d_F (n x n, device,
 * leading dimension ldf) by trace-correcting second-order spectral projection (csrc/jcdf_sp2.hpp) — matrix
 * squarings on MFMA, no host round trip.  `iterations` squarings are enqueued; those after convergence return at
 * once.  On return (stream-ordered) d_P (n x n, leading dimension ldp) holds the current iterate and d_info
 * (8 doubles, device) = {squarings done, finished (1/0), tr P, last tr(X - X^2), spectral bounds lo, hi, accelerated (1/0), delta}:
 * the caller accepts P when finished == 1 and |tr P - n_occ| is small, and otherwise calls again with more
 * iterations or takes the eigensolver.  d_work: jcdf_sp2_workspace_bytes(n) bytes of device memory. */
int64_t jcdf_sp2_workspace_bytes(int64_t n);
int32_t jcdf_sp2_device(void *stream, int64_t n, int64_t n_occ, const double *d_F, int64_t ldf, double *d_P, int64_t ldp,
                        int32_t iterations, void *d_work, int64_t work_bytes, double *d_info);
/* The same with a REFERENCE decomposition: d_Fref (n x n, leading dimension ldr) is a matrix the caller has diagonalised before
 * (an earlier SCF iteration) and d_ref_eigs (4 doubles, device) = {lowest, HOMO (n_occ-th), LUMO, highest} of ITS eigenvalues.
 * With delta = ||d_F - d_Fref||_F (formed on the device) Weyl's inequality bounds the spectrum of d_F by
 * [lowest - delta, highest + delta] — taken instead of the Gershgorin interval where tighter — and brackets its gap by
 * HOMO + delta < LUMO - delta; while that bracket is open the accelerated recursion of Rubensson (JCTC 7, 1233) is used (each
 * step preceded by the affine stretch that folds the far end of the spectrum onto itself; csrc/jcdf_sp2.hpp): about half the
 * squarings.  The bounds are rigorous, so the result is the same projector; info[6] = 1 when the accelerated recursion ran,
 * info[7] = delta.  Both NULL: jcdf_sp2_device. */
int32_t jcdf_sp2_ref_device(void *stream, int64_t n, int64_t n_occ, const double *d_F, int64_t ldf, double *d_P, int64_t ldp,
                            int32_t iterations, void *d_work, int64_t work_bytes, double *d_info, const double *d_Fref, int64_t ldr,
                            const double *d_ref_eigs);


/* Small dense products of the device-resident SCF iteration on the library's own fp64 MFMA cores (csrc/jcdf_blas.hpp), so
 * that an iteration issues no vendor BLAS kernel (reference: BLAS.symm!/gemm! at SCF.jl:473-481, 1080-1108).  Row-major
 * device matrices; M, N multiples of 32 (zero padded), leading dimensions even.
 *   _tn:  C[m][n] = alpha sum_k A[k][m] B[k][n]   (K multiple of 32; for symmetric A: C = alpha A B)
 *   _nt:  C[m][n] = sum_k A[m][k] B[n][k]         (K multiple of 16; the eigensolver's back-transformation U = Q Z) */
int32_t jcdf_gemm_tn_device(void *stream, int64_t M, int64_t N, int64_t K, double alpha, const double *d_A, int64_t lda,
                            const double *d_B, int64_t ldb, double *d_C, int64_t ldc);
int32_t jcdf_gemm_nt_device(void *stream, int64_t M, int64_t N, int64_t K, const double *d_A, int64_t lda, const double *d_B,
                            int64_t ldb, double *d_C, int64_t ldc);
/* DIIS history bookkeeping on the device (EnergyHelpers.jl:234-258, SCF.jl:473-488): e = T^T - T for T = S D F (leading
 * dimension ld) packed n x n into d_e_slot, F into d_f_slot; the new row of Pulay dot products <e_s, e_head> for all nd
 * slots (each of length len); and the extrapolated F = sum_s coef[s] F_s written back with leading dimension ld. */
int32_t jcdf_diis_push_device(void *stream, int64_t n, int64_t ld, const double *d_T, const double *d_F, double *d_e_slot, double *d_f_slot);
int32_t jcdf_diis_dots_device(void *stream, int32_t nd, int32_t head, int64_t len, const double *d_e_hist, double *d_dots,
                              double *d_work /* 64 * nd doubles */);
int32_t jcdf_diis_mix_device(void *stream, int32_t nd, int64_t n, int64_t ld, const double *d_f_hist, const double *d_coef, double *d_F);
/* The whole DIIS + damping part of one SCF iteration (SCF.jl:472-505) as ONE call / four launches: push (e = T^T - T and F into
 * slot `head` of the histories), the new row of Pulay dot products (partial sums in d_work, 64 * nd doubles), the bordered
 * system with n_use vectors (solve != 0; as jcdf_diis_device, consuming the partial sums directly), and
 * d_F = (1 - x) d_F_old + x sum_s coef[s] F_s (x = 1: no damping, d_F_old may be NULL; without solve and damping d_F is left as it is). */
int32_t jcdf_diis_step_device(void *stream, int32_t nd, int32_t head, int32_t n_use, int32_t solve, int64_t n, int64_t ld, const double *d_T,
                              double *d_F, double *d_e_hist, double *d_f_hist, double *d_Bmat, double *d_coef, int32_t *d_flag, double *d_work,
                              const double *d_F_old, double x);

/* ---- introspection ------------------------------------------------------------ */
/* Device bytes held (reference: get_gpu_data_size_dense_MB, DenseGPUDF.jl:305-319). */
int64_t jcdf_device_bytes(const jcdf_handle *h);
/* Per-kernel stats of the last build; returns the number of records written
 * (<= max_records). */
int32_t jcdf_kernel_stats(jcdf_handle *h, jcdf_kernel_stat *out, int32_t max_records);
/* The same records with `seconds` SUMMED over all builds since the last reset (the event times of a build are folded in
 * when the next build is enqueued, or here for the last one), *n_builds = builds in the sums, *fock_seconds = summed
 * whole-build time: per-launch averages over a timed loop without any host call inside the loop.  A build that was
 * still running when its successor was enqueued is left out of the sums (and of *n_builds). */
int32_t jcdf_kernel_stats_total(jcdf_handle *h, jcdf_kernel_stat *out, int32_t max_records, int64_t *n_builds, double *fock_seconds,
                                int32_t reset);

/* ---- multi-device group: all devices of one process behind ONE call, F reduced ON THE DEVICES ---------------
 * Replaces, for scf flag num_devices > 1: the one-task-per-device loop (GPUDF.jl:188-193, DenseGPUDF.jl:83-86), the
 * D2H of every device's F + the host axpy! over devices (GPUDF.jl:267-277, DenseGPUDF.jl:131-137) and, inside a node,
 * the MPI.Allreduce! of the partial Fock matrices (DensityFitting.jl:68-71).  A group owns n handles (member i = aux
 * shard i on device_ids[i]); ONE host thread drives them all:
 *   - C_occ goes up ONCE (H2D to member 0) and reaches the other devices device-to-device (hipMemcpyPeerAsync over xGMI);
 *   - every member runs the same kernels as jcdf_fock_build on its shard, concurrently, each on its own stream;
 *   - the n partial Fock matrices are summed on the devices by a reduce-scatter over slices of the N*N elements
 *     (jcdf_group_reduce_plan), transport "rccl": ncclReduceScatter of librccl.so.1 (dlopen'ed; one communicator per
 *     member, ncclCommInitAll) — transport "peer": a hand-written kernel on every member that sums its slice from all
 *     members' peer-mapped buffers in FIXED member order (bit-reproducible; needs hipDeviceEnablePeerAccess);
 *   - the reduced slices leave the devices as ONE pass of D2H copies into F_out (slice i from device i: N*N doubles in
 *     total, each device over its own PCIe link), or, for a device-resident caller, are gathered on member 0.
 * "auto" (default) = "rccl" when the members are n > 1 distinct devices and librccl.so.1 can be loaded, else "peer".
 * Several members MAY share a device (several aux shards per GPU; also how the group logic is tested on one GPU) —
 * only the "peer" transport serves such a group (RCCL refuses duplicate devices).  A transport that cannot be set up
 * or fails returns JCDF_ERR_HIP / JCDF_ERR_INVALID with the reason in jcdf_group_last_error: there is NO silent
 * fallback to a host-side reduce.  Calls on one group must come from one host thread at a time. */
typedef struct jcdf_group jcdf_group;
#define JCDF_GROUP_MAX_DEVICES 16

/* Group-level times of the last jcdf_group_fock_build* in seconds (device events on member streams / the host clock). */
typedef struct jcdf_group_timings {
    double bcast_time;     /* C_occ: H2D to member 0 + device-to-device copies to the others   GPUDF.jl:206 (one H2D per device there) */
    double build_time;     /* longest member Fock build (device events)                         "GPU_-N-_fock_time-"   */
    double reduce_time;    /* device-side reduce-scatter of F (longest member)                 replaces axpy! GPUDF.jl:273-276 */
    double d2h_time;       /* the one pass of slice copies into F_out (host entry) / gather on member 0 (device entry) */
    double total_time;     /* host clock around the whole call                                  "total_fock_gpu_time-" */
} jcdf_group_timings;

/* device_ids[i] = HIP device of member i (0 <= n_devices <= JCDF_GROUP_MAX_DEVICES).  JCDF_ERR_NO_DEVICE as jcdf_create. */
int32_t jcdf_group_create(jcdf_group **out, int32_t n_devices, const int32_t *device_ids);
int32_t jcdf_group_destroy(jcdf_group *g);
/* Message of the last failing call on g (g == NULL: last jcdf_group_create failure). */
const char *jcdf_group_last_error(const jcdf_group *g);
int32_t jcdf_group_size(const jcdf_group *g);
/* Member i's handle, BORROWED (never jcdf_destroy it): for the per-handle calls that have no group form — jcdf_set_tuning,
 * jcdf_get_V/W/B, jcdf_set_B, jcdf_kernel_stats, jcdf_device_bytes ...  NULL when i is out of range. */
jcdf_handle *jcdf_group_handle(jcdf_group *g, int32_t i);
/* "auto" | "rccl" | "peer" (before the first Fock build or between builds).  JCDF_ERR_INVALID for an unknown name or
 * "rccl" on a group with a shared device; JCDF_ERR_HIP when librccl.so.1 / peer access is not available. */
int32_t jcdf_group_set_transport(jcdf_group *g, const char *name);
/* Transport in effect, e.g. "rccl 2.22.3 (librccl.so.1, 8 ranks)" or "peer (8 members, fixed-order slice sums)". */
const char *jcdf_group_transport(const jcdf_group *g);
/* The reduce-scatter partition of `count` = N*N elements over n members, exported for tests (pure host code):
 * offsets[i] .. offsets[i+1] is member i's slice (equal chunks of a multiple of 256 elements, the last ones clipped to
 * count; offsets has n + 1 entries).  Returns the chunk length, or -1 on bad arguments. */
int64_t jcdf_group_reduce_plan(int64_t count, int32_t n, int64_t *offsets);

/* jcdf_configure on every member: shard_q0[i] .. shard_q0[i+1] = member i's aux function range (n + 1 entries,
 * ascending and contiguous: calculate_device_ranges_GPU, GPUDF.jl:1026-1056; dense path static_load_rank_indicies
 * per device, DenseGPUDF.jl:228-236).  One process: shard_q0[0] = 0 and shard_q0[n] = Q_total; with several ranks a
 * group holds its rank's devices, a contiguous part of the auxiliary basis, and its F is that part's partial sum.
 * jcdf_set_tuning on the members and jcdf_group_set_exchange_screening go before it. */
int32_t jcdf_group_configure(jcdf_group *g, int64_t N, int64_t Q_total, const int64_t *shard_q0, int64_t n_occ,
                             int64_t P, const int64_t *pq_p, const int64_t *pq_q);
int32_t jcdf_group_set_exchange_screening(jcdf_group *g, int64_t n_blocks);
/* Metric: potrf + trtri ONCE, on member 0's device ("always do J_AB_INV on the first device", DenseGPUDF.jl:177-193);
 * the other members take their rows of L^-1 device-to-device (reference: D2H + one H2D per device, DenseGPUDF.jl:221-227). */
int32_t jcdf_group_set_metric(jcdf_group *g, const double *J2c);
/* Three-centre rows [s0,s1) (as jcdf_push_three_center): ONE H2D, to the first member that needs the block; the other
 * members read it device-to-device (reference: one H2D of the block per device, DenseGPUDF.jl:258-262, GPUDF.jl:864-866).
 * Members whose rows lie above the block (s0 >= q1) are skipped: L^-1 is lower triangular.  T may also be device memory
 * of any device of the process (the first copy is a hipMemcpyDefault): a block received over RCCL is pushed as it is. */
int32_t jcdf_group_push_three_center(jcdf_group *g, int64_t s0, int64_t s1, const double *T);
/* H on member 0 only (GPUDF.jl:158-161, 221-225); NULL: no member adds H. */
int32_t jcdf_group_set_core_hamiltonian(jcdf_group *g, const double *H);

/* One Fock build over all members: C_occ N x n_occ column-major host, F_out N x N host, fully overwritten with
 * sum_i (2 J_i - K_i) (+ H) — the contract of scf_data.two_electron_fock after the device loop and host reduce of
 * GPUDF.jl:188-277.  t: n_devices entries (member i's device timings) or NULL; gt may be NULL. */
int32_t jcdf_group_fock_build(jcdf_group *g, const double *C_occ, double *F_out, jcdf_timings *t, jcdf_group_timings *gt);
/* The same for a device-resident caller on member 0's device (as jcdf_fock_build_device_ld): orbital i at
 * d_C_occ + ldc * i, column p of the REDUCED F at d_F + ldf * p.  Work is ordered behind `stream` (a hipStream_t of
 * member 0's device; NULL = member 0's stream) and the result is complete on that stream when the call returns
 * (NOT synchronised): the SCF loop of a single process can keep its replicated part on one device and still shard the
 * Fock build over all of them. */
int32_t jcdf_group_fock_build_device_ld(jcdf_group *g, const double *d_C_occ, int64_t ldc, double *d_F, int64_t ldf, void *stream);
/* Blocks until the last group build has finished on every member; fills t (n entries) / gt like jcdf_group_fock_build. */
int32_t jcdf_group_synchronize(jcdf_group *g, jcdf_timings *t, jcdf_group_timings *gt);

#ifdef __cplusplus
}
#endif
#endif /* JCDF_H */
