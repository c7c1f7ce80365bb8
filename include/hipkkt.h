/*
 * hipkkt.h -- C ABI of libhipkkt.so: an MI355X-native (HIP, gfx950) KKT linear-system solver
 * that drops in behind Clarabel.jl's KKT-solver interfaces.  fp64 throughout.
 *
 * Two boundaries are exported, plus the two layers either side of them
 * (section 8f); citations are into /root/reference:
 *
 *   Level A  hipkkt_ldl_*   replaces an AbstractDirectLDLSolver backend
 *            contract  src/kktsolvers/direct-ldl/directldl_defaults.jl:1-72
 *            example   src/kktsolvers/direct-ldl/directldl_qdldl.jl:1-96   (the CPU path)
 *   Level B  hipkkt_kkt_*   replaces the whole DirectLDLKKTSolver <: AbstractKKTSolver
 *            contract  src/kktsolvers/kktsolver_defaults.jl:2-48
 *            example   src/kktsolvers/kktsolver_directldl.jl:5-466
 *   Level C  hipkkt_kkt_system_*   the caller of level B, DefaultKKTSystem, with its vectors in HBM
 *            src/kktsystem.jl:21-215  (kkt_update!, kkt_solve_initial_point!, kkt_solve!)
 *   Data     hipkkt_equilibrate / hipkkt_scale_matrix_values   Ruiz equilibration before level B is built
 *            src/problemdata.jl:133-242, src/data_updating.jl:169-194
 *
 * Conventions
 *   - plain pointers and sizes only; the caller owns every array it passes and may free or
 *     move it as soon as the call returns (the library copies during the call).
 *   - index arrays are int64 with the caller's `index_base` (1 for Julia, 0 for C/Python).
 *   - return value: 0 = success; > 0 = numeric failure (non-finite pivot or residual), which the
 *     reference reports as `false` (directldl_qdldl.jl:79, kktsolver_directldl.jl:411,429);
 *     < 0 = usage or HIP error (hipkkt_last_error() has the text).
 *   - one handle = one HIP stream; calls on one handle must be sequential (the reference calls
 *     its backend from one thread: update_values!* -> refactor! -> solve!*).
 *   - functions ending in _dev take DEVICE pointers (resident in HBM on the handle's device) and
 *     are asynchronous on the handle's stream unless they return a status that needs a read-back.
 */
#ifndef HIPKKT_H
#define HIPKKT_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HIPKKT_OK 0
#define HIPKKT_NUMERIC_FAILURE 1
#define HIPKKT_REFINEMENT_INCOMPLETE 2   /* hipkkt_kkt_deferred_status only */
#define HIPKKT_ERR_ARG (-1)
#define HIPKKT_ERR_HIP (-2)
#define HIPKKT_ERR_INTERNAL (-3)

/* cone kinds (src/cones/cone_api.jl:18-55); dims[] = numel, except PSD: matrix side */
#define HIPKKT_CONE_ZERO 0
#define HIPKKT_CONE_NN 1
#define HIPKKT_CONE_SOC 2
#define HIPKKT_CONE_PSD 3

/* fill-reducing ordering */
#define HIPKKT_ORDER_AMD 0      /* approximate minimum degree (what the reference asks QDLDL for) */
#define HIPKKT_ORDER_ND 1       /* nested dissection over AMD leaves: short, bushy trees (default) */
#define HIPKKT_ORDER_NATURAL 2
#define HIPKKT_ORDER_USER 3     /* settings.user_perm */

typedef struct hipkkt_ldl_s *hipkkt_ldl_t;
typedef struct hipkkt_kkt_s *hipkkt_kkt_t;

/* mirrors the path-relevant fields of Clarabel.Settings (src/settings.jl:110-132) */
typedef struct {
    double static_regularization_constant;      /* 1e-8   :118 */
    double static_regularization_proportional;  /* eps^2  :119 */
    double dynamic_regularization_eps;          /* 1e-13  :123 */
    double dynamic_regularization_delta;        /* 2e-7   :124 */
    double iterative_refinement_reltol;         /* 1e-13  :128 */
    double iterative_refinement_abstol;         /* 1e-12  :129 */
    double iterative_refinement_stop_ratio;     /* 5      :131 */
    int32_t iterative_refinement_max_iter;      /* 10     :130 */
    int32_t static_regularization_enable;       /* true   :117 */
    int32_t iterative_refinement_enable;        /* true   :127 */
    int32_t ordering;                           /* HIPKKT_ORDER_*, default ND */
    int32_t nd_leaf_size;                       /* sub-domain size below which ND hands over to AMD */
    int32_t device;                             /* HIP device ordinal, -1 = current */
    const int64_t *user_perm;                   /* length N, caller's index base; ORDER_USER only */
    double amd_dense_scale;                     /* 1.5 (directldl_qdldl.jl:24) */
} hipkkt_settings;

typedef struct {
    int64_t n, m, p, N;        /* N = n + m + p, p = 2 per sparse second-order cone */
    int64_t nnzK;              /* entries of the triu KKT matrix */
    int64_t nnzL;              /* structural nnz(L) (QDLDL's count; what linear_solver_info reports) */
    int64_t nnzL_stored;       /* incl. explicit zeros of amalgamated supernodes */
    int64_t nsuper, nlevels, max_front, etree_height;
    int64_t nHs;               /* length of Hsblocks */
    int64_t nsparse_soc, sparse_soc_len;
    double factor_flops;
    double front_bytes, update_bytes;
} hipkkt_info;

/* accumulated device time per phase (hipEvents on the handle's stream), for bench.py */
typedef struct {
    double update_ms, factor_ms, trisolve_ms, residual_ms, other_ms;
    int64_t n_update, n_factor, n_trisolve, n_residual;
    int64_t ir_iterations;     /* refinement rounds beyond the first residual check */
    int64_t dynamic_regularizations;
    /* Fallbacks taken over the handle's LIFETIME (not cleared by hipkkt_kkt_profile_reset): a bounded wait of the
     * factorisation's overlap mode / of the persistent top-of-tree sweep kernel expired, the mechanism was switched off
     * for the handle and the operation repeated level by level.  The reference has nothing like it (its only report is the
     * Bool of refactor! / solve!, directldl_qdldl.jl:79): results stay correct, but a non-zero count means a 50 ms stall
     * happened and the handle runs on its slower path from then on.  Expected value: 0. */
    int64_t overlap_fallbacks, top_fallbacks;
    /* Factorisations of this handle (lifetime) that ran level by level because ANOTHER handle's overlapped factorisation
     * was in flight on the device: one overlapped factorisation per device at a time (the mode's forward-progress
     * argument needs every other kernel on the device to end by itself).  Not a fallback: no stall, nothing is
     * switched off, the next factorisation asks again. */
    int64_t overlap_deferrals;
    /* ... and sweeps that took the chained kernels instead of the persistent one for the same reason (one kernel with a
     * residency requirement per device at a time, whichever handle's). */
    int64_t top_deferrals;
} hipkkt_profile;

/* -------------------------------------------------------------------- general */
int hipkkt_available(void);                 /* 1 if a gfx950 device is usable; never throws
                                               (ldlsolver_is_available, directldl_defaults.jl:22-27) */
const char *hipkkt_last_error(void);
void hipkkt_default_settings(hipkkt_settings *s);
const char *hipkkt_version(void);

/* host-only symbolic analysis of a triu CSC pattern (no GPU needed): fill-reducing permutation
 * (perm[k] = 0-based original index eliminated k-th) and the structure statistics.  This is the
 * setup step QDLDL.qdldl(...; logical=true) performs for the reference (directldl_qdldl.jl:18-25). */
int hipkkt_symbolic_analyse(int64_t N, const int64_t *colptr, const int64_t *rowval, int index_base,
                            int ordering, int nd_leaf_size, int64_t *perm_out, hipkkt_info *info_out);

/* ------------------------------------------- Level A: AbstractDirectLDLSolver */
/* constructor (directldl_qdldl.jl:6-28): symbolic analysis of the triu CSC matrix K, keeps a
 * device copy of nzval.  dsigns: +1/-1 expected pivot signs (kktsolver_directldl.jl:112-126). */
int hipkkt_ldl_create(hipkkt_ldl_t *out, int64_t N, const int64_t *colptr, const int64_t *rowval,
                      const double *nzval, const int64_t *dsigns, const hipkkt_settings *settings,
                      int index_base);
void hipkkt_ldl_destroy(hipkkt_ldl_t h);
/* update_values! / scale_values! (directldl_qdldl.jl:46-68): index = positions in K.nzval */
int hipkkt_ldl_update_values(hipkkt_ldl_t h, const int64_t *index, const double *values, int64_t k);
int hipkkt_ldl_scale_values(hipkkt_ldl_t h, const int64_t *index, double scale, int64_t k);
/* refactor! (directldl_qdldl.jl:72-81): numeric LDL^T; 1 if some pivot is not finite */
int hipkkt_ldl_refactor(hipkkt_ldl_t h);
/* solve! (directldl_qdldl.jl:85-96): x = K^{-1} b, host vectors of length N, x != b allowed */
int hipkkt_ldl_solve(hipkkt_ldl_t h, double *x, const double *b);
int hipkkt_ldl_solve_dev(hipkkt_ldl_t h, double *d_x, const double *d_b);
/* solve! for nrhs right-hand sides against the same factors (SURVEY.md 8b "batched", 8e(ii):
 * the reference has no such call -- its solve! at directldl_qdldl.jl:85-96 takes one vector --
 * a caller with several vectors loops over it).  X, B: N x nrhs column-major, host (ld = N) or
 * device (ldx, ldb >= N); X may alias B.  Column j's result equals hipkkt_ldl_solve on column j. */
int hipkkt_ldl_solve_multi(hipkkt_ldl_t h, int64_t nrhs, double *X, const double *B);
int hipkkt_ldl_solve_multi_dev(hipkkt_ldl_t h, int64_t nrhs, double *d_X, int64_t ldx,
                               const double *d_B, int64_t ldb);
/* linear_solver_info (directldl_qdldl.jl:35-42) */
int hipkkt_ldl_info(hipkkt_ldl_t h, hipkkt_info *info);
int hipkkt_ldl_get_perm(hipkkt_ldl_t h, int64_t *perm /* N, 0-based */);
/* out[0] = overlap-mode fallbacks, out[1] = persistent-sweep-kernel fallbacks of this handle so far (see hipkkt_profile) */
int hipkkt_ldl_fallbacks(hipkkt_ldl_t h, int64_t out[2]);

/* ---------------------------------------------- Level B: AbstractKKTSolver */
/* constructor (kktsolver_directldl.jl:46-92): P n x n triu CSC, A m x n CSC, cone list.
 * Does the KKT assembly + data maps (directldl_kkt_assembly.jl:15-175), Dsigns, symbolic LDL. */
int hipkkt_kkt_create(hipkkt_kkt_t *out, int64_t n, int64_t m,
                      const int64_t *Pcolptr, const int64_t *Prowval, const double *Pnzval,
                      const int64_t *Acolptr, const int64_t *Arowval, const double *Anzval,
                      int64_t ncones, const int32_t *cone_kinds, const int64_t *cone_dims,
                      const hipkkt_settings *settings, int index_base);
void hipkkt_kkt_destroy(hipkkt_kkt_t h);
int hipkkt_kkt_info(hipkkt_kkt_t h, hipkkt_info *info);

/* kktsolver_update! (kktsolver_directldl.jl:197-294) with the cone data the reference reads from
 * its cones: Hsblocks = get_Hs! output (positive W'W blocks, |Hs| values), and for each sparse
 * second-order cone its u, v (concatenated) and eta^2.  Scatter, regularise, refactor. */
int hipkkt_kkt_update_cones(hipkkt_kkt_t h, const double *Hsblocks, const double *soc_u,
                            const double *soc_v, const double *soc_eta2);
/* device-native variant: NT scaling + Hs blocks computed on the device from (s, z)
 * (update_scaling! + get_Hs!, src/cones/coneops_*.jl), then as above.  Returns 1 also when a
 * point is not interior (update_scaling! returning false). */
int hipkkt_kkt_update_from_sz(hipkkt_kkt_t h, const double *s, const double *z);
int hipkkt_kkt_update_from_sz_dev(hipkkt_kkt_t h, const double *d_s, const double *d_z);
/* kktsolver_update_P! / kktsolver_update_A! (kktsolver_directldl.jl:374-386) */
int hipkkt_kkt_update_P(hipkkt_kkt_t h, const double *Pnzval);
int hipkkt_kkt_update_A(hipkkt_kkt_t h, const double *Anzval);
/* kktsolver_setrhs! (:313-327) and kktsolver_solve! (:346-371, with iterative refinement
 * :389-449).  lhsx / lhsz may be NULL (Union{Nothing,...}). */
int hipkkt_kkt_setrhs(hipkkt_kkt_t h, const double *rhsx, const double *rhsz);
int hipkkt_kkt_solve(hipkkt_kkt_t h, double *lhsx, double *lhsz);
int hipkkt_kkt_setrhs_dev(hipkkt_kkt_t h, const double *d_rhsx, const double *d_rhsz);
int hipkkt_kkt_solve_dev(hipkkt_kkt_t h, double *d_lhsx, double *d_lhsz);
/* Deferred status (no counterpart in the reference, whose calls are synchronous): with defer = 1 the device-pointer
 * entry points hipkkt_kkt_update_from_sz_dev, hipkkt_kkt_solve_dev and hipkkt_kkt_solve_multi_dev with nrhs <= 8 (the
 * columns that share the single-column sweeps; ir_iterations then receives -1 per column) only ENQUEUE their work and
 * return 0 at once (hipkkt_kkt_solve_multi_dev with more than 8 columns always synchronises and returns its own status);
 * what they would have returned is accumulated on the device.  The refinement loop's accept / stop decisions
 * (kktsolver_directldl.jl:389-449) are taken on the device either way; in this mode a solve runs as many refinement
 * rounds ahead as the previous solves on the handle needed (at least one).  hipkkt_kkt_deferred_status synchronises,
 * returns the worst status since the previous query and clears it: 0, HIPKKT_NUMERIC_FAILURE (some pivot, cone point
 * or residual was not finite), or HIPKKT_REFINEMENT_INCOMPLETE (some solve stopped while the reference's loop would
 * have gone on: its result is the last accepted iterate; later solves run one more round ahead -- repeat the step,
 * or call hipkkt_kkt_solve_dev with defer = 0, for the reference's exact loop).  A caller issues a whole iteration's
 * update + solves and asks once. */
int hipkkt_kkt_set_deferred_status(hipkkt_kkt_t h, int defer);
int hipkkt_kkt_deferred_status(hipkkt_kkt_t h);
/* kktsolver_setrhs! + kktsolver_solve! for nrhs right-hand sides at once (SURVEY.md 8b
 * "hipkkt_kkt_solve_multi"): rhsx n x nrhs, rhsz m x nrhs, column-major, contiguous; lhsx / lhsz
 * likewise, either may be NULL.  Every column goes through the reference's refinement rule on its
 * own (kktsolver_directldl.jl:389-449) and ends where its single solve would; ir_iterations
 * (host, nrhs entries, may be NULL) receives the rounds each column took.  Returns 1 if any
 * column's residual is not finite.  Does not disturb the right-hand side set by hipkkt_kkt_setrhs. */
int hipkkt_kkt_solve_multi(hipkkt_kkt_t h, int64_t nrhs, const double *rhsx, const double *rhsz,
                           double *lhsx, double *lhsz, int64_t *ir_iterations);
int hipkkt_kkt_solve_multi_dev(hipkkt_kkt_t h, int64_t nrhs, const double *d_rhsx,
                               const double *d_rhsz, double *d_lhsx, double *d_lhsz,
                               int64_t *ir_iterations);
/* ------------------------------------------- Level C: DefaultKKTSystem, device-resident
 * The layer that calls the KKT solver three times per interior-point iteration
 * (/root/reference/src/kktsystem.jl:21-215) with all its vectors in HBM: right-hand-side
 * construction (Delta_s_from_Delta_z_offset!, coneops_compositecone.jl:185-202) and the recovery of
 * (dtau, dx, dz, ds, dkappa) (dots, quad_form mathutils.jl:299-337, mul_Hs!) run on the device, so
 * per solve only the scalars cross PCIe.  Vectors named d_* are
 * device pointers: x-like length n, s/z-like length m.  Covers zero, nonnegative, second-order and PSD
 * cones (PSD side <= 48, the limit of hipkkt_kkt_update_from_sz). */
/* DefaultKKTSystem constructor (kktsystem.jl:21-52): q (n), b (m) host vectors, copied */
int hipkkt_kkt_system_init(hipkkt_kkt_t h, const double *q, const double *b);
/* kkt_update! (kktsystem.jl:62-78): cone scaling from (s, z), refactor, constant-RHS solve */
int hipkkt_kkt_system_update(hipkkt_kkt_t h, const double *d_s, const double *d_z);
/* _kkt_solve_constant_rhs! (kktsystem.jl:80-92) alone, after hipkkt_kkt_update_cones */
int hipkkt_kkt_system_solve_constant_rhs(hipkkt_kkt_t h);
/* kkt_solve_initial_point! (kktsystem.jl:95-143): LP / QP branch on nnz(P) */
int hipkkt_kkt_system_solve_initial_point(hipkkt_kkt_t h, double *d_x, double *d_s, double *d_z);
/* kkt_solve! (kktsystem.jl:145-215): lhs <- step for right-hand side rhs at the iterate
 * `variables`; steptype 0 = :affine, 1 = :combined.  lhs_tau_kappa (host, 2) receives (dtau, dkappa). */
int hipkkt_kkt_system_solve(hipkkt_kkt_t h, double *d_lhs_x, double *d_lhs_s, double *d_lhs_z,
                            double *lhs_tau_kappa,
                            const double *d_rhs_x, const double *d_rhs_s, const double *d_rhs_z,
                            double rhs_tau, double rhs_kappa,
                            const double *d_var_x, const double *d_var_s, const double *d_var_z,
                            double var_tau, double var_kappa, int steptype);

/* kkt_update! followed by the affine kkt_solve! in ONE call (kktsystem.jl:62-92 and :145-215 with steptype :affine).
 * The constant right-hand side (-q, b) of kkt_update! and the affine right-hand side (rhs.x, s - rhs.z) do not depend on
 * each other, so their two solves share every triangular sweep (one 2-column solve with per-column refinement).  Same
 * results as hipkkt_kkt_system_update(h, d_var_s, d_var_z) followed by hipkkt_kkt_system_solve(..., steptype 0); the
 * affine step does not read rhs.s (:157-158). */
int hipkkt_kkt_system_update_and_solve_affine(hipkkt_kkt_t h, double *d_lhs_x, double *d_lhs_s, double *d_lhs_z,
                                              double *lhs_tau_kappa, const double *d_rhs_x, const double *d_rhs_z,
                                              double rhs_tau, double rhs_kappa,
                                              const double *d_var_x, const double *d_var_s, const double *d_var_z,
                                              double var_tau, double var_kappa);

/* Lazy constant-RHS solve: the same pairing reached through the reference's own TWO calls, so that solver.jl:278-295
 * stays as it is.  With lazy = 1, kkt_update! (hipkkt_kkt_system_update / _update_cones) scales, scatters and
 * refactors, returns the factorisation's status and only NOTES that (x2, z2) = K \ (-q, b) is due
 * (kktsystem.jl:71-77 would solve it at once); the next hipkkt_kkt_system_solve with steptype :affine sends both
 * right-hand sides through the sweeps as one 2-column solve and returns the AND of the two solves' status -- which is
 * what solver.jl:279-295 computes from the two calls (`is_kkt_solve_success = kkt_update!(...)`, then
 * `is_kkt_solve_success && kkt_solve!(..., :affine)`), so the loop's control flow is unchanged.  Any other consumer of
 * (x2, z2) -- a :combined solve arriving first, a structure that takes one right-hand side per sweep -- makes the
 * pending solve run by itself first; hipkkt_kkt_system_solve_initial_point does not read (x2, z2) and leaves it
 * pending (the next kkt_update! supersedes it: solver.jl:389-393). */
int hipkkt_kkt_system_set_lazy(hipkkt_kkt_t h, int lazy);
/* kkt_update!(kktsystem, data, cones) for a caller that keeps the reference's cone objects (the Julia glue: kkt_update!
 * gets `cones`, not the iterate): the reference's data for kktsolver_update! (as hipkkt_kkt_update_cones) plus the NT
 * scaling the step recovery of kkt_solve! reads from the same cones -- w (m: nonnegative cones sqrt(s/z), second-order
 * cones the normalised w, coneops_nncone.jl:75-86, coneops_socone.jl:75-123), eta (one per cone; second-order cones),
 * lambda (m; a PSD cone of side k keeps its k values in the first k of its slots), and R, Rinv of the PSD cones
 * (k x k column-major, concatenated in cone order; coneops_psdtrianglecone.jl:127-132).  Then the constant-RHS solve,
 * or its note in lazy mode. */
int hipkkt_kkt_system_update_cones(hipkkt_kkt_t h, const double *Hsblocks, const double *soc_u, const double *soc_v,
                                   const double *soc_eta2, const double *w, const double *eta, const double *lambda,
                                   const double *psd_R, const double *psd_Rinv);
/* The same kkt_update! from the NT scaling ALONE: the Hs blocks and the sparse second-order-cone vectors u, v, eta^2 are
 * functions of (w, eta) and of R (get_Hs!: coneops_nncone.jl:91-101, coneops_socone.jl:125-192,
 * coneops_psdtrianglecone.jl:135-161) and are formed on the device -- for the headline workload 3.2 MB cross PCIe per
 * iteration instead of 6.4, for PSD cones of side k the k x k factor R instead of the t(t+1)/2-entry block, t = k(k+1)/2.
 * The values equal get_Hs!'s to round-off (bit for bit when the scaling came from the device: the tests pin that).  In
 * lazy mode the call only enqueues, like hipkkt_kkt_system_update; the arrays may be reused when it returns. */
int hipkkt_kkt_system_update_scaling(hipkkt_kkt_t h, const double *w, const double *eta, const double *lambda,
                                     const double *psd_R, const double *psd_Rinv);
/* The same entry points for a caller whose iterate lives in HOST memory (DefaultVariables are Vector{T},
 * variables.jl:1-30): vectors are staged through buffers the handle owns (n + 2m doubles each way per call).
 * hipkkt_kkt_system_solve_host: var_x = var_s = var_z = NULL means "the variables of the previous call" -- they do not
 * change between the affine and the combined kkt_solve! of an iteration (solver.jl:289-323), so the glue sends them once. */
int hipkkt_kkt_system_update_host(hipkkt_kkt_t h, const double *s, const double *z);
int hipkkt_kkt_system_solve_initial_point_host(hipkkt_kkt_t h, double *x, double *s, double *z);
int hipkkt_kkt_system_solve_host(hipkkt_kkt_t h, double *lhs_x, double *lhs_s, double *lhs_z, double *lhs_tau_kappa,
                                 const double *rhs_x, const double *rhs_s, const double *rhs_z,
                                 double rhs_tau, double rhs_kappa,
                                 const double *var_x, const double *var_s, const double *var_z,
                                 double var_tau, double var_kappa, int steptype);

/* Self-test of the hand-over protocol by which kernels that run side by side pass data (persistent / chained sweep
 * kernels, the factorisation's overlap mode; contract stated in csrc/factor_kernels.hip): `pairs` producer / consumer
 * workgroup pairs on different XCDs hand `words` doubles over `rounds` times.  variant 0 = the contract; 1 = without the
 * producer's s_waitcnt before its signal; 2 = with plain instead of agent-scope payload accesses.  out[0] = payload words
 * read stale, out[1] = expired waits.  Test infrastructure (tests/test_gpu_parity.py::test_handover_litmus): variant 0
 * must give (0, 0). */
int hipkkt_selftest_handover(int variant, int pairs, int words, int rounds, int device, int64_t out[2]);

/* Page-lock a host array the caller keeps for the solver's lifetime (an interior-point method's work vectors:
 * DefaultVariables, the right-hand sides, the cones' w / lambda) so that the copies of the *_host entry points run at
 * the link's rate and without the runtime's per-call pinning (cfg2: the host-vector iteration 4.4 -> 3.9 ms).  Optional;
 * unregister before the array is freed.  Returns HIPKKT_OK, or HIPKKT_ERR_HIP if the range cannot be registered (the
 * entry points work with unregistered memory all the same). */
int hipkkt_host_register(void *ptr, int64_t bytes);
int hipkkt_host_unregister(void *ptr);

/* ------------------------------------------- problem-data scaling (before the KKT solver is built)
 * data_equilibrate! (/root/reference/src/problemdata.jl:133-221): Ruiz equilibration of
 * [P A'; A 0], q, b on the device.  P: n x n upper-triangular CSC,
 * A: m x n CSC.  Pnzval, Anzval, q, b are overwritten with c D P D, E A D, c D q, E b; d (n),
 * e (m) and c (1) receive the scalings (all ones / one when max_iter = 0).  Defaults of the
 * reference: max_iter 10, min_scaling 1e-4, max_scaling 1e4 (settings.jl:98-101).  Cones that do
 * not admit elementwise scaling (second-order, PSD) get one common factor per cone
 * (rectify_equilibration!, coneops_defaults.jl:32-44). */
int hipkkt_equilibrate(int64_t n, int64_t m,
                       const int64_t *Pcolptr, const int64_t *Prowval, double *Pnzval,
                       const int64_t *Acolptr, const int64_t *Arowval, double *Anzval,
                       double *q, double *b,
                       int64_t ncones, const int32_t *cone_kinds, const int64_t *cone_dims,
                       int32_t max_iter, double min_scaling, double max_scaling,
                       double *d, double *e, double *c, int index_base, int device);
/* _update_matrix (data_updating.jl:169-194), the re-scaling update_P! / update_A! apply to new
 * values before kktsolver_update_P!/A!: nzval <- cscale * lscale[row] * rscale[col] * nzval
 * (P: lscale = rscale = d, cscale = c; A: lscale = e, rscale = d, cscale = 1). */
int hipkkt_scale_matrix_values(int64_t nrows, int64_t ncols, const int64_t *colptr,
                               const int64_t *rowval, double *nzval, const double *lscale,
                               const double *rscale, double cscale, int index_base, int device);

/* y = W'W x over all cones with the current scaling (mul_Hs!, coneops_compositecone.jl:138-150);
 * valid after hipkkt_kkt_update_from_sz*.  Host vectors of length m. */
int hipkkt_kkt_mul_Hs(hipkkt_kkt_t h, double *y, const double *x);

/* introspection used by the parity tests: the assembled K (triu CSC, 0-based) and data maps */
int hipkkt_kkt_get_pattern(hipkkt_kkt_t h, int64_t *colptr /* N+1 */, int64_t *rowval /* nnzK */);
int hipkkt_kkt_get_values(hipkkt_kkt_t h, double *nzval /* nnzK, un-regularised */);
int hipkkt_kkt_get_maps(hipkkt_kkt_t h, int64_t *mapP, int64_t *mapA, int64_t *mapHs,
                        int64_t *map_diag_full, int64_t *map_soc_u, int64_t *map_soc_v,
                        int64_t *map_soc_D, int64_t *dsigns);   /* any may be NULL */
int hipkkt_kkt_get_perm(hipkkt_kkt_t h, int64_t *perm /* N, 0-based */);
int hipkkt_kkt_get_Hs(hipkkt_kkt_t h, double *Hsblocks /* |Hs|, positive */);
/* the NT scaling held on the device after hipkkt_kkt_update_from_sz*: the scaled point lambda (length m; a PSD cone
 * of side k keeps its k singular values, descending, in the first k of its slots), and for the PSD cones R and Rinv
 * (coneops_psdtrianglecone.jl:127-132), k x k column-major each, concatenated in cone order.  Any may be NULL. */
int hipkkt_kkt_get_scaling(hipkkt_kkt_t h, double *lambda, double *psd_R, double *psd_Rinv);
/* the rest of the device's NT scaling: w (m) and eta (one per cone), as hipkkt_kkt_system_update_cones takes them */
int hipkkt_kkt_get_scaling_w(hipkkt_kkt_t h, double *w, double *eta);
double hipkkt_kkt_last_regularizer(hipkkt_kkt_t h);
int64_t hipkkt_kkt_last_ir_iterations(hipkkt_kkt_t h);
/* (tests) the refinement rounds a solve enqueues ahead of its first status read-back -- what the previous solves took
 * (kktsolver_directldl.jl:397-449 decides round by round; here the rounds are enqueued speculatively and the reference's
 * accept / stop rule runs on the device).  set >= 0 replaces it first; returns the value in force, < 0 on a null handle. */
int hipkkt_kkt_speculative_rounds(hipkkt_kkt_t h, int set);

/* run on the caller's stream (e.g. torch's current stream) instead of the handle's own */
int hipkkt_kkt_set_stream(hipkkt_kkt_t h, void *hip_stream);
int hipkkt_kkt_synchronize(hipkkt_kkt_t h);
/* per-phase device timing */
int hipkkt_kkt_profile_enable(hipkkt_kkt_t h, int enable);
int hipkkt_kkt_profile_reset(hipkkt_kkt_t h);
int hipkkt_kkt_profile_get(hipkkt_kkt_t h, hipkkt_profile *out);

#ifdef __cplusplus
}
#endif
#endif
