/*
 * oracle/kkt_oracle.h -- TEST INFRASTRUCTURE ONLY.
 *
 * CPU restatement (plain C99, single thread) of the reference's KKT hot path:
 * Clarabel.jl v0.11.0 `DirectLDLKKTSolver` + the QDLDL direct-LDL engine it calls.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * this library.  The product (cuclarabel_amd/, libhipkkt.so) never links, imports
 * or calls it.
 *
 * PARITY STATUS
 *   - factor level (K, L, D): PARITY UNPINNED.  The reference holds no test that pins
 *     the KKT matrix, the factors or a single K x = b solve, and
 *     the LDL arithmetic lives in the third-party QDLDL.jl 0.4.x (Project.toml:14,38),
 *     which is not under /root/reference; its published algorithm is restated here.
 *   - solution level: pinned by (1) scipy.sparse.linalg.splu / dense numpy on the same
 *     K, b (tests/test_oracle_ldl.py), (2) the reference's cone-algebra unit tests
 *     (test/UnitTests/test_coneops_secondordercone.jl:31-66,
 *     test_coneops_psdtrianglecone.jl:213-251) restated in tests/test_oracle_cones.py,
 *     (3) the reference's end-to-end known answers (test/OptTests/basic_qp.jl, basic_lp.jl,
 *     basic_socp.jl, basic_eq_constrained.jl incl. the redundant-row and dual-infeasible cases,
 *     typed in as tests/golden/reference_fixtures.py) through the IPM test driver
 *     cuclarabel_amd/ipm.py with this oracle as its KKT backend (tests/test_ipm_fixtures.py):
 *     all reproduced at the reference's own tolerance (atol 1e-3).
 *
 * All indices are 0-based int64 here (the reference is 1-based Int64).
 */
#ifndef KKT_ORACLE_H
#define KKT_ORACLE_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int64_t orc_int;

/* cone kinds; dims[] holds numel for ZERO/NN/SOC and the matrix side k for PSD
 * (numel = k(k+1)/2), as the reference's PSDTriangleConeT(k) does (cone_types.jl:171-186) */
enum { ORC_ZERO = 0, ORC_NN = 1, ORC_SOC = 2, ORC_PSD = 3 };

typedef struct orc_kkt orc_kkt;

typedef struct {
    double static_reg_constant;      /* settings.jl:118  1e-8  */
    double static_reg_proportional;  /* settings.jl:119  eps^2 */
    double dynamic_reg_eps;          /* settings.jl:123  1e-13 */
    double dynamic_reg_delta;        /* settings.jl:124  2e-7  */
    double ir_reltol;                /* settings.jl:128  1e-13 */
    double ir_abstol;                /* settings.jl:129  1e-12 */
    double ir_stop_ratio;            /* settings.jl:131  5     */
    int    ir_max_iter;              /* settings.jl:130  10    */
    int    static_reg_enable;        /* settings.jl:117  true  */
    int    ir_enable;                /* settings.jl:127  true  */
} orc_settings;

void orc_default_settings(orc_settings *s);

/* P: n x n upper-triangular CSC; A: m x n CSC; perm: fill-reducing ordering to use
 * (perm[k] = original index placed k-th), or NULL for the oracle's own minimum-degree. */
orc_kkt *orc_kkt_new(orc_int n, orc_int m,
                     const orc_int *Pp, const orc_int *Pi, const double *Px,
                     const orc_int *Ap, const orc_int *Ai, const double *Ax,
                     orc_int ncones, const int *kinds, const orc_int *dims,
                     const orc_int *perm, const orc_settings *settings);
void orc_kkt_free(orc_kkt *k);

/* sizes: out[0]=n out[1]=m out[2]=p out[3]=N out[4]=nnzK out[5]=|Hsblocks| out[6]=nnzL
 * out[7]=#sparse SOC out[8]=sum of sparse SOC dims */
void orc_kkt_sizes(const orc_kkt *k, orc_int *out);
/* borrowed pointers to the assembled triu CSC KKT matrix and its data maps */
const orc_int *orc_kkt_colptr(const orc_kkt *k);
const orc_int *orc_kkt_rowval(const orc_kkt *k);
const double  *orc_kkt_nzval(const orc_kkt *k);
const orc_int *orc_kkt_map_P(const orc_kkt *k);
const orc_int *orc_kkt_map_A(const orc_kkt *k);
const orc_int *orc_kkt_map_Hs(const orc_kkt *k);
const orc_int *orc_kkt_map_diag_full(const orc_kkt *k);
const orc_int *orc_kkt_map_soc_u(const orc_kkt *k);   /* concatenated over sparse SOCs */
const orc_int *orc_kkt_map_soc_v(const orc_kkt *k);
const orc_int *orc_kkt_map_soc_D(const orc_kkt *k);   /* 2 per sparse SOC */
const orc_int *orc_kkt_dsigns(const orc_kkt *k);
const orc_int *orc_kkt_perm(const orc_kkt *k);
double orc_kkt_last_regularizer(const orc_kkt *k);
orc_int orc_kkt_last_ir_iters(const orc_kkt *k);
orc_int orc_kkt_num_dyn_regularized(const orc_kkt *k);

/* cones: update_scaling! (coneops_*.jl) -- returns 1 on success, 0 if (s,z) not interior */
int  orc_cones_update_scaling(orc_kkt *k, const double *s, const double *z);
void orc_cones_set_identity_scaling(orc_kkt *k);
/* get_Hs! (coneops_compositecone.jl:123-132): positive W^T W blocks, length |Hsblocks| */
void orc_cones_get_Hs(const orc_kkt *k, double *Hs);
/* mul_Hs! (coneops_compositecone.jl:138-150): y = W^T W x over all cones, length m */
void orc_cones_mul_Hs(const orc_kkt *k, double *y, const double *x);
/* sparse SOC data as the reference keeps it: u, v concatenated; eta2, d per sparse SOC */
void orc_cones_soc_sparse(const orc_kkt *k, double *u, double *v, double *eta2, double *d);
/* lambda (scaled variable) for all cones, length m (PSD cones: zero-padded diag form) */
void orc_cones_lambda(const orc_kkt *k, double *lam);
/* PSD cones' R and Rinv (coneops_psdtrianglecone.jl:127-132), k x k col-major each, concatenated in cone order */
void orc_cones_psd_scaling(const orc_kkt *k, double *R, double *Rinv);

/* kktsolver_update! (kktsolver_directldl.jl:197-294): scatter -Hs and the sparse-cone
 * columns from the current cone scaling, regularise, refactor.  returns 1 on success. */
int orc_kkt_update(orc_kkt *k);
/* same, but with caller-provided Hs data (what boundary B hands over): Hs positive */
int orc_kkt_update_values(orc_kkt *k, const double *Hs, const double *soc_u,
                          const double *soc_v, const double *soc_eta2);
/* kktsolver_update_P!/A! (kktsolver_directldl.jl:374-386) */
void orc_kkt_update_P(orc_kkt *k, const double *Px);
void orc_kkt_update_A(orc_kkt *k, const double *Ax);
/* kktsolver_setrhs!/solve! (kktsolver_directldl.jl:313-371). lhsx/lhsz may be NULL. */
void orc_kkt_setrhs(orc_kkt *k, const double *rhsx, const double *rhsz);
int  orc_kkt_solve(orc_kkt *k, double *lhsx, double *lhsz);
/* the bare LDL solve without IR (directldl_qdldl.jl:85-96): x = K_reg^{-1} b, length N */
void orc_ldl_solve(const orc_kkt *k, double *x, const double *b);
/* refactor only (directldl_qdldl.jl:72-81) with whatever values are loaded */
int  orc_ldl_refactor(orc_kkt *k);
/* e = b - K_sym x on the un-regularised K, returns ||e||_inf (kktsolver_directldl.jl:455-466) */
double orc_kkt_residual(const orc_kkt *k, double *e, const double *b, const double *x);
/* factor access for tests: D^{-1} (length N, permuted order) */
const double *orc_ldl_Dinv(const orc_kkt *k);

/* stand-alone minimum-degree ordering on a triu CSC pattern (oracle's own; perm out) */
void orc_min_degree(orc_int N, const orc_int *colptr, const orc_int *rowval, orc_int *perm);

#ifdef __cplusplus
}
#endif
#endif
