/* The C ABI from plain C (no Python, no torch): what a Julia ccall or any other FFI does.
 *
 * Problem: the basic QP of the reference's tests (test/OptTests/basic_qp.jl:6-19),
 *   P = [4 1; 1 2], A = [-I3x2-ish; ...] as below, cones = Nonnegative(6).
 * Level B (kktsystem.jl:62-92): create -> kktsolver_update! under identity scaling (Hs = 1) -> setrhs!(-q, b) -> solve!,
 * then the residual of the un-regularised KKT system is checked on the host.  Level A (AbstractDirectLDLSolver) on the
 * same K: constructor, update_values! of the regularised diagonal, refactor!, solve!.  Level C (DefaultKKTSystem) in lazy
 * mode: kkt_update!, kkt_solve!(:affine), kkt_solve!(:combined) as three calls with host vectors.
 *
 * Build: gcc -O2 -I include tests/c_abi/kkt_from_c.c -o /tmp/kkt_from_c -L cuclarabel_amd -lhipkkt -Wl,-rpath,$PWD/cuclarabel_amd -lm
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "hipkkt.h"

#define CHECK(call)                                                                      \
    do {                                                                                 \
        int rc_ = (call);                                                                \
        if (rc_ != 0) {                                                                  \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, hipkkt_last_error());          \
            return 1;                                                                    \
        }                                                                                \
    } while (0)

int main(void)
{
    if (!hipkkt_available()) {
        fprintf(stderr, "no gfx950 device\n");
        return 2;
    }
    /* P upper triangle, CSC, 0-based */
    const int64_t n = 2, m = 6;
    const int64_t Pp[] = {0, 1, 3}, Pi[] = {0, 0, 1};
    const double Px[] = {4.0, 1.0, 2.0};
    /* A = [-A0; A0], A0 = [1 1; 1 0; 0 1], CSC */
    const int64_t Ap[] = {0, 4, 8}, Ai[] = {0, 1, 3, 4, 0, 2, 3, 5};
    const double Ax[] = {-1.0, -1.0, 1.0, 1.0, -1.0, -1.0, 1.0, 1.0};
    const double q[] = {1.0, 1.0}, b[] = {-1.0, 0.0, 0.0, 1.0, 0.7, 0.7};
    const int32_t kinds[] = {HIPKKT_CONE_NN};
    const int64_t dims[] = {6};

    hipkkt_settings st;
    hipkkt_default_settings(&st);
    hipkkt_kkt_t h = NULL;
    CHECK(hipkkt_kkt_create(&h, n, m, Pp, Pi, Px, Ap, Ai, Ax, 1, kinds, dims, &st, 0));

    hipkkt_info info;
    CHECK(hipkkt_kkt_info(h, &info));
    if (info.N != 8 || info.nHs != 6) { fprintf(stderr, "unexpected sizes\n"); return 1; }

    double Hs[6] = {1, 1, 1, 1, 1, 1};                    /* identity scaling */
    CHECK(hipkkt_kkt_update_cones(h, Hs, NULL, NULL, NULL));
    double rx[2] = {-q[0], -q[1]}, x[2], z[6];
    CHECK(hipkkt_kkt_setrhs(h, rx, b));
    CHECK(hipkkt_kkt_solve(h, x, z));

    /* residual of [P A'; A -I] [x; z] = [-q; b] */
    const double Pf[2][2] = {{4, 1}, {1, 2}};
    const double A0[3][2] = {{1, 1}, {1, 0}, {0, 1}};
    double Af[6][2], r = 0.0;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 2; ++j) { Af[i][j] = -A0[i][j]; Af[i + 3][j] = A0[i][j]; }
    for (int j = 0; j < 2; ++j) {
        double v = Pf[j][0] * x[0] + Pf[j][1] * x[1] - rx[j];
        for (int i = 0; i < 6; ++i) v += Af[i][j] * z[i];
        if (fabs(v) > r) r = fabs(v);
    }
    for (int i = 0; i < 6; ++i) {
        const double v = Af[i][0] * x[0] + Af[i][1] * x[1] - z[i] - b[i];
        if (fabs(v) > r) r = fabs(v);
    }
    printf("level B: x = (%.12f, %.12f), refinement rounds %lld, residual %.3e\n", x[0], x[1],
           (long long)hipkkt_kkt_last_ir_iterations(h), r);
    if (!(r < 1e-11)) { fprintf(stderr, "residual too large\n"); return 1; }

    /* ---------------------------------------------------------------- level A: AbstractDirectLDLSolver
     * (directldl_defaults.jl:1-72; the sequence of kktsolver_directldl.jl:247-310): the caller owns K.  Take the
     * assembled, un-regularised K of the level-B handle, construct the backend on it, shift the diagonal by +-eps
     * through update_values! (what _kktsolver_regularize_and_refactor! does), refactor!, solve!, and check the
     * residual of the REGULARISED system on the host (level A has no refinement of its own). */
    {
        int64_t Kp[9], Ki[64];
        double Kx[64];
        CHECK(hipkkt_kkt_get_pattern(h, Kp, Ki));
        CHECK(hipkkt_kkt_get_values(h, Kx));
        const int64_t N = 8, nnzK = Kp[N];
        if (nnzK > 64) { fprintf(stderr, "unexpected nnz(K)\n"); return 1; }
        int64_t dsigns[8], diag_idx[8];
        double diag_val[8], Kd[8][8] = {{0}};
        const double eps = 1e-8;
        for (int64_t j = 0; j < N; ++j) {
            dsigns[j] = j < n ? 1 : -1;
            diag_idx[j] = Kp[j + 1] - 1;                   /* the diagonal is the last entry of every column */
            if (Ki[diag_idx[j]] != j) { fprintf(stderr, "diagonal not last in column %lld\n", (long long)j); return 1; }
            diag_val[j] = Kx[diag_idx[j]] + (double)dsigns[j] * eps;
        }
        hipkkt_ldl_t l = NULL;
        CHECK(hipkkt_ldl_create(&l, N, Kp, Ki, Kx, dsigns, &st, 0));
        CHECK(hipkkt_ldl_update_values(l, diag_idx, diag_val, N));
        CHECK(hipkkt_ldl_refactor(l));
        double rhs[8], sol[8];
        for (int j = 0; j < 2; ++j) rhs[j] = rx[j];
        for (int i = 0; i < 6; ++i) rhs[2 + i] = b[i];
        CHECK(hipkkt_ldl_solve(l, sol, rhs));
        for (int64_t j = 0; j < N; ++j)
            for (int64_t e = Kp[j]; e < Kp[j + 1]; ++e) {
                const double v = Ki[e] == j ? diag_val[j] : Kx[e];
                Kd[Ki[e]][j] = v;
                Kd[j][Ki[e]] = v;
            }
        double ra = 0.0, da = 0.0;
        for (int i = 0; i < 8; ++i) {
            double v = -rhs[i];
            for (int j = 0; j < 8; ++j) v += Kd[i][j] * sol[j];
            if (fabs(v) > ra) ra = fabs(v);
        }
        for (int j = 0; j < 2; ++j) if (fabs(sol[j] - x[j]) > da) da = fabs(sol[j] - x[j]);
        for (int i = 0; i < 6; ++i) if (fabs(sol[2 + i] - z[i]) > da) da = fabs(sol[2 + i] - z[i]);
        hipkkt_info li;
        CHECK(hipkkt_ldl_info(l, &li));
        int64_t fb[2];
        CHECK(hipkkt_ldl_fallbacks(l, fb));
        printf("level A: residual of the regularised system %.3e, distance to level B's refined solution %.3e, nnz(L) %lld\n",
               ra, da, (long long)li.nnzL);
        hipkkt_ldl_destroy(l);
        if (!(ra < 1e-11) || !(da < 1e-6) || fb[0] != 0 || fb[1] != 0) { fprintf(stderr, "level A check failed\n"); return 1; }
    }

    /* ---------------------------------------------------------------- level C: DefaultKKTSystem, lazy mode
     * The three calls of an interior-point iteration exactly as solver.jl:278-323 issues them -- kkt_update!,
     * kkt_solve!(:affine), kkt_solve!(:combined) -- with HOST vectors (the *_host entry points the Julia glue binds),
     * the handle in lazy mode (the constant-RHS solve of kkt_update! rides with the affine one), the combined call
     * re-using the affine call's variables.  Checked against the defining equations of kkt_solve!
     * (kktsystem.jl:145-215) on the host: with (x1, z1) = K \ (rhs.x, const - rhs.z), (x2, z2) = K \ (-q, b),
     *   dtau = tau_num / tau_den,  dx = x1 + dtau x2,  dz = z1 + dtau z2,  ds = -(Hs dz + const),  dkappa = -(rhs.kappa + kappa dtau) / tau. */
    {
        CHECK(hipkkt_kkt_system_init(h, q, b));
        CHECK(hipkkt_kkt_system_set_lazy(h, 1));
        double sv[6], zv[6], xv[2] = {0.3, -0.2};
        for (int i = 0; i < 6; ++i) { sv[i] = 1.0 + 0.1 * i; zv[i] = 2.0 - 0.2 * i; }
        const double tau = 1.3, kappa = 0.7, rtau = 0.4, rkappa = -0.2;
        double rhsx[2] = {0.5, -1.0}, rhss[6], rhsz[6];
        for (int i = 0; i < 6; ++i) { rhss[i] = 0.3 - 0.1 * i; rhsz[i] = -0.4 + 0.15 * i; }
        CHECK(hipkkt_kkt_system_update_host(h, sv, zv));
        double dx[2], ds[6], dz[6], tk[2];
        /* Hs = diag(s / z) for the nonnegative cone (coneops_nncone.jl:77-101) */
        double Hd[6];
        for (int i = 0; i < 6; ++i) Hd[i] = sv[i] / zv[i];
        for (int step = 0; step < 2; ++step) {
            const int affine = step == 0;
            CHECK(hipkkt_kkt_system_solve_host(h, dx, ds, dz, tk, rhsx, rhss, rhsz, rtau, rkappa,
                                               affine ? xv : NULL, affine ? sv : NULL, affine ? zv : NULL, tau, kappa, affine ? 0 : 1));
            /* the constant term of Delta_s: s itself (affine), ds_rhs / z for the nonnegative cone (combined; coneops_nncone.jl:140-148) */
            double konst[6], rc = 0.0;
            for (int i = 0; i < 6; ++i) konst[i] = affine ? sv[i] : rhss[i] / zv[i];
            /* the step satisfies the reduced system: P dx + A' dz + q dtau = rhs.x ... checked through its definition:
             * [P A'; A -Hs] [dx; dz] = [rhs.x; const - rhs.z] + dtau [-q; b] */
            for (int j = 0; j < 2; ++j) {
                double v = Pf[j][0] * dx[0] + Pf[j][1] * dx[1] - rhsx[j] + tk[0] * q[j];
                for (int i = 0; i < 6; ++i) v += Af[i][j] * dz[i];
                if (fabs(v) > rc) rc = fabs(v);
            }
            for (int i = 0; i < 6; ++i) {
                const double v = Af[i][0] * dx[0] + Af[i][1] * dx[1] - Hd[i] * dz[i] - (konst[i] - rhsz[i]) - tk[0] * b[i];
                if (fabs(v) > rc) rc = fabs(v);
                const double w = ds[i] + Hd[i] * dz[i] + konst[i];
                if (fabs(w) > rc) rc = fabs(w);
            }
            const double dk = -(rkappa + kappa * tk[0]) / tau;
            if (fabs(dk - tk[1]) > rc) rc = fabs(dk - tk[1]);
            printf("level C (%s, lazy): dtau %.12f dkappa %.12f, defect of the step's defining equations %.3e\n",
                   affine ? "affine" : "combined", tk[0], tk[1], rc);
            if (!(rc < 1e-9)) { fprintf(stderr, "level C check failed\n"); return 1; }
        }
        hipkkt_profile pr;
        CHECK(hipkkt_kkt_profile_get(h, &pr));
        if (pr.overlap_fallbacks != 0 || pr.top_fallbacks != 0) { fprintf(stderr, "a fallback was taken\n"); return 1; }
    }
    hipkkt_kkt_destroy(h);
    printf("C ABI OK\n");
    return 0;
}
