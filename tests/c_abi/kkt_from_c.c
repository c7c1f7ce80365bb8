/* The C ABI from plain C (no Python, no torch): what a Julia ccall or any other FFI does.
 *
 * Problem: the basic QP of the reference's tests (test/OptTests/basic_qp.jl:6-19),
 *   P = [4 1; 1 2], A = [-I3x2-ish; ...] as below, cones = Nonnegative(6).
 * Sequence (kktsystem.jl:62-92): create -> kktsolver_update! under identity scaling (Hs = 1) ->
 * setrhs!(-q, b) -> solve!, then the residual of the un-regularised KKT system is checked on the host.
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
    printf("x = (%.12f, %.12f), refinement rounds %lld, residual %.3e\n", x[0], x[1],
           (long long)hipkkt_kkt_last_ir_iterations(h), r);
    hipkkt_kkt_destroy(h);
    if (!(r < 1e-11)) { fprintf(stderr, "residual too large\n"); return 1; }
    printf("C ABI OK\n");
    return 0;
}
