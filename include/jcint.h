/* jcint.h — host Gaussian-integral engine of libjcdf_hip.so (the producers just off the
 * hot path).  Plain C ABI; everything here runs on the HOST (north_star keeps integral generation there), threads
 * over shell pairs.  It stands where the reference's JERI/Libint engines stand:
 *   - deps/src/jeri-df-tei.hpp:51-95   DFRHFTEIEngine: compute_eri_block_df (3-centre), compute_two_center_eri_block
 *   - deps/src/jeri-oei.hpp:61,106,155 overlap / kinetic / nuclear attraction
 *   - deps/src/jeri-tei.hpp:67-70      4-centre blocks, used here only for the Schwarz diagonal (pq|pq)
 * and produces arrays in exactly the layouts the callers build from them:
 *   - calculate_two_center_intgrals            src/rhf/energy/DensityFitting/TwoCenterIntegrals.jl:7-29
 *   - calculate_three_center_integrals         .../ThreeCenterIntegrals.jl:9-42, ThreeCenterIntegralsScreened.jl:8-85
 *   - schwarz_screen_itegrals_df               .../SchwarzScreening.jl:9-71
 * Conventions (same as the reference): Cartesian functions in Libint order (xx,xy,xz,yy,yz,zz; ...),
 * (l+1)(l+2)/2 per shell (BasisStructs.jl:31-33), EVERY Cartesian function unit-normalised (Libint normalises the
 * axial function, JuliaChem's axial_normalization_factor rescales the others: Globals.jl:6-28,
 * EnergyHelpers.jl:260-411); contraction coefficients refer to normalised primitives (basis-set-exchange style).
 * Algorithm: McMurchie-Davidson (Hermite Gaussians), fp64.  Status codes: jcdf_status of jcdf.h. */
#ifndef JCINT_H
#define JCINT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct jcint_basis jcint_basis;

/* A basis = list of contracted shells in the order of the reference's Basis object (JCBasis.jl: atoms in input order,
 * shells in table order; an "L" (sp) shell is passed as its s shell followed by its p shell).
 *   l[s] angular momentum (0..6), nprim[s], centers[3*s..] in bohr; exps / coefs: the primitives of all shells
 *   concatenated. */
int32_t jcint_basis_create(jcint_basis **out, int64_t nshell, const int32_t *l, const int32_t *nprim, const double *exps,
                           const double *coefs, const double *centers);
void jcint_basis_destroy(jcint_basis *b);
int64_t jcint_nbf(const jcint_basis *b);               /* number of Cartesian basis functions */
int64_t jcint_nshell(const jcint_basis *b);
/* nbas_out[s] = functions of shell s (what DynamicLoad.jl:160-203 shards by) */
int32_t jcint_shell_sizes(const jcint_basis *b, int64_t *nbas_out);

/* Overlap S, kinetic T, nuclear attraction V (N x N, symmetric, fully stored); any of S, T, V may be NULL.
 * Z, R: natoms charges and centres (3 each, bohr). */
int32_t jcint_one_electron(const jcint_basis *b, int64_t natoms, const double *Z, const double *R, double *S, double *T,
                           double *V);
double jcint_nuclear_repulsion(int64_t natoms, const double *Z, const double *R);       /* EnergyHelpers.jl:5-23 */

/* (P|Q), Q x Q column-major, fully stored (the reference keeps the lower triangle, TwoCenterIntegrals.jl:150-162). */
int32_t jcint_two_center(const jcint_basis *aux, double *J);

/* (Q|mu nu) for the auxiliary FUNCTION range [q0, q1) (must start and end on shell boundaries: a shard of
 * static_load_rank_indicies), as the (q1-q0) x P column-major block the path consumes:
 *   pq_p, pq_q (length P): packed pair c -> (p, q) of sparse_pq_index_map (SchwarzScreening.jl:72-81);
 *   NULL, NULL with P == N*N: the dense map c = q + N*p (SchwarzScreening.jl:97-111). */
int32_t jcint_three_center(const jcint_basis *aux, const jcint_basis *prim, int64_t q0, int64_t q1, int64_t P,
                           const int64_t *pq_p, const int64_t *pq_q, double *T);

/* Schwarz data of SchwarzScreening.jl:9-71: M[p][q] = (pq|pq), N x N symmetric — a FUNCTION pair is kept iff
 * |(pq|pq)| >= sigma^2 / max_P (P|P) (:59-60) — and (optional, may be NULL) shell_sum[m][n] = the sum over the whole
 * (mn|mn) shell-quartet block, nshell x nshell — a SHELL pair is kept iff |sum| >= the same threshold (:44-47);
 * functions of a screened shell pair are all screened (:49-55). */
int32_t jcint_schwarz(const jcint_basis *prim, double *M, double *shell_sum);

/* worker threads for the calls above (default: hardware concurrency, at most 64) */
void jcint_set_threads(int32_t n);

#ifdef __cplusplus
}
#endif
#endif
