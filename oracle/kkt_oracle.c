/*
 * oracle/kkt_oracle.c -- TEST INFRASTRUCTURE ONLY (see kkt_oracle.h for the parity status).
 *
 * Plain-C restatement of the reference's KKT hot path (Clarabel.jl v0.11.0):
 *   KKT assembly + data maps      src/kktsolvers/direct-ldl/directldl_kkt_assembly.jl:15-175
 *                                 src/utils/csc_assembly.jl:3-272
 *                                 src/kktsolvers/direct-ldl/directldl_datamaps.jl:8-79,170-214
 *   cone layout                   src/cones/compositecone_type.jl:114-141
 *   NT scaling / Hs blocks        src/cones/coneops_{zero,nn,so,psdtriangle}cone.jl
 *   update / regularise / IR      src/kktsolvers/kktsolver_directldl.jl:112-126,211-466
 *   LDL engine as called          src/kktsolvers/direct-ldl/directldl_qdldl.jl:6-96
 *                                 (QDLDL.jl 0.4.x is NOT in the tree; its published
 *                                 up-looking algorithm is restated)
 * Nothing here is shipped; the product never calls it.
 */
#include "kkt_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <float.h>

typedef orc_int I;

/* ------------------------------------------------------------------ cones */
typedef struct {
    int kind;
    I dim;      /* numel for ZERO/NN/SOC; matrix side for PSD */
    I numel;
    I off;      /* first index in (s,z): rng_cones */
    I boff;     /* first index in Hsblocks: rng_blocks */
    I blen;
    int sparse; /* SOC with dim > SOC_NO_EXPANSION_MAX_SIZE(4), cone_types.jl:101-112 */
    I sidx;     /* index among sparse SOCs; soff: offset in concatenated u/v */
    I soff;
    double eta, d;
    double *w, *lam, *u, *v;        /* NN: w, lam; SOC: w, lam, (u, v) */
    double *R, *Rinv, *Hs, *plam;   /* PSD: k x k col-major R, Rinv; t x t Hs; k lam */
} cone_t;

struct orc_kkt {
    I n, m, p, N, nnzK, nHs, ncones, nsparse, sparse_len;
    orc_settings st;
    cone_t *cones;
    /* KKT triu CSC */
    I *Kp, *Ki; double *Kx;
    /* maps */
    I *mapP, *mapA, *mapHs, *mapDiag, *mapU, *mapV, *mapD;
    I nnzP, nnzA;
    I *dsigns;
    /* LDL engine (QDLDL-equivalent) */
    I *perm, *iperm;
    I *Tp, *Ti; double *Tx;      /* triu(P K P^T) and the value map A->PAPt */
    I *AtoPAPt;
    I *pdsigns;
    I *etree, *Lnz, *Lp, *Li; double *Lx, *D, *Dinv;
    I nnzL;
    I *iwork; unsigned char *bwork; double *fwork;
    I ndyn;
    /* solve state */
    double *x, *b, *work1, *work2, *solvetmp;
    double *Hsvals;
    double last_eps;
    I last_ir;
};

void orc_default_settings(orc_settings *s)
{
    s->static_reg_constant = 1e-8;
    s->static_reg_proportional = DBL_EPSILON * DBL_EPSILON;
    s->dynamic_reg_eps = 1e-13;
    s->dynamic_reg_delta = 2e-7;
    s->ir_reltol = 1e-13;
    s->ir_abstol = 1e-12;
    s->ir_stop_ratio = 5.0;
    s->ir_max_iter = 10;
    s->static_reg_enable = 1;
    s->ir_enable = 1;
}

static I tri(I k) { return (k * (k + 1)) / 2; }

/* -------------------------------------------------- minimum degree (oracle's own)
 * Exact external-degree minimum degree on a quotient graph with element absorption.
 * Only used when no permutation is handed in.  Any permutation gives a valid LDL^T of
 * the quasi-definite K, so this affects fill, not results. */
typedef struct { I *v; I len, cap; } ivec;
static void ipush(ivec *a, I x)
{
    if (a->len == a->cap) {
        a->cap = a->cap ? 2 * a->cap : 4;
        a->v = (I *)realloc(a->v, (size_t)a->cap * sizeof(I));
    }
    a->v[a->len++] = x;
}

void orc_min_degree(I N, const I *colptr, const I *rowval, I *perm)
{
    ivec *adj = (ivec *)calloc((size_t)N, sizeof(ivec));   /* variable neighbours */
    ivec *els = (ivec *)calloc((size_t)N, sizeof(ivec));   /* adjacent elements   */
    ivec *Le = (ivec *)calloc((size_t)N, sizeof(ivec));    /* element members     */
    I *deg = (I *)malloc((size_t)N * sizeof(I));
    I *mark = (I *)malloc((size_t)N * sizeof(I));
    unsigned char *state = (unsigned char *)calloc((size_t)N, 1); /* 0 var, 1 element, 2 dead */
    I *head = (I *)malloc((size_t)(N + 1) * sizeof(I));
    I *next = (I *)malloc((size_t)N * sizeof(I));
    I *prev = (I *)malloc((size_t)N * sizeof(I));
    I tag = 0, mindeg = 0, k, i, j, q;
    for (j = 0; j < N; j++)
        for (q = colptr[j]; q < colptr[j + 1]; q++) {
            i = rowval[q];
            if (i != j) { ipush(&adj[i], j); ipush(&adj[j], i); }
        }
    for (i = 0; i <= N; i++) head[i] = -1;
    for (i = 0; i < N; i++) mark[i] = -1;
#define DL_INSERT(ii) do { I d_ = deg[ii]; next[ii] = head[d_]; prev[ii] = -1; \
        if (head[d_] >= 0) prev[head[d_]] = (ii); head[d_] = (ii); } while (0)
#define DL_REMOVE(ii) do { I d_ = deg[ii]; if (prev[ii] >= 0) next[prev[ii]] = next[ii]; \
        else head[d_] = next[ii]; if (next[ii] >= 0) prev[next[ii]] = prev[ii]; } while (0)
    for (i = 0; i < N; i++) {
        /* dedupe adjacency */
        I w = 0; tag++;
        for (q = 0; q < adj[i].len; q++) {
            j = adj[i].v[q];
            if (mark[j] != tag) { mark[j] = tag; adj[i].v[w++] = j; }
        }
        adj[i].len = w; deg[i] = w;
    }
    for (i = 0; i < N; i++) mark[i] = -1;
    tag = 0;
    for (i = 0; i < N; i++) DL_INSERT(i);
    for (k = 0; k < N; k++) {
        I pv;
        while (mindeg < N && head[mindeg] < 0) mindeg++;
        pv = head[mindeg];
        DL_REMOVE(pv);
        perm[k] = pv;
        /* form the new element Lp = adj(pv) U members of adjacent elements, minus pv */
        tag++;
        mark[pv] = tag;
        for (q = 0; q < adj[pv].len; q++) {
            j = adj[pv].v[q];
            if (state[j] == 0 && mark[j] != tag) { mark[j] = tag; ipush(&Le[pv], j); }
        }
        for (q = 0; q < els[pv].len; q++) {
            I e = els[pv].v[q], r;
            if (state[e] != 1) continue;
            for (r = 0; r < Le[e].len; r++) {
                j = Le[e].v[r];
                if (state[j] == 0 && mark[j] != tag) { mark[j] = tag; ipush(&Le[pv], j); }
            }
            state[e] = 2;             /* absorbed */
            free(Le[e].v); Le[e].v = NULL; Le[e].len = Le[e].cap = 0;
        }
        state[pv] = 1;
        free(adj[pv].v); adj[pv].v = NULL; adj[pv].len = adj[pv].cap = 0;
        free(els[pv].v); els[pv].v = NULL; els[pv].len = els[pv].cap = 0;
        /* update each member */
        {
            I lptag = tag;
            for (q = 0; q < Le[pv].len; q++) {
                I w, r, d;
                i = Le[pv].v[q];
                DL_REMOVE(i);
                /* prune variable adjacency: drop dead vars and members of Lp */
                w = 0;
                for (r = 0; r < adj[i].len; r++) {
                    j = adj[i].v[r];
                    if (state[j] == 0 && mark[j] != lptag) adj[i].v[w++] = j;
                }
                adj[i].len = w;
                /* prune element list, add pv */
                w = 0;
                for (r = 0; r < els[i].len; r++) {
                    I e = els[i].v[r];
                    if (state[e] == 1) els[i].v[w++] = e;
                }
                els[i].len = w;
                ipush(&els[i], pv);
            }
            /* exact external degrees */
            for (q = 0; q < Le[pv].len; q++) {
                I r, d = 0, t2;
                i = Le[pv].v[q];
                tag++;
                t2 = tag;
                /* marks for Lp membership were overwritten below only with new tags; re-mark lazily */
                {
                    I *seen = mark;
                    seen[i] = t2;
                    for (r = 0; r < adj[i].len; r++) {
                        j = adj[i].v[r];
                        if (seen[j] != t2) { seen[j] = t2; d++; }
                    }
                    for (r = 0; r < els[i].len; r++) {
                        I e = els[i].v[r], s2;
                        for (s2 = 0; s2 < Le[e].len; s2++) {
                            j = Le[e].v[s2];
                            if (state[j] == 0 && seen[j] != t2) { seen[j] = t2; d++; }
                        }
                    }
                }
                deg[i] = d;
            }
            for (q = 0; q < Le[pv].len; q++) {
                i = Le[pv].v[q];
                DL_INSERT(i);
                if (deg[i] < mindeg) mindeg = deg[i];
            }
            (void)lptag;
        }
    }
#undef DL_INSERT
#undef DL_REMOVE
    for (i = 0; i < N; i++) { free(adj[i].v); free(els[i].v); free(Le[i].v); }
    free(adj); free(els); free(Le); free(deg); free(mark); free(state);
    free(head); free(next); free(prev);
}

/* ------------------------------------------------------------ KKT assembly
 * triu K = [P+diag  A'  0 ; .  -Hs  V ; . . D]   (directldl_kkt_assembly.jl:15-175).
 * Within every column rows ascend and the diagonal is the last entry. */
static void assemble_kkt(orc_kkt *k, const I *Pp, const I *Pi, const double *Px,
                         const I *Ap, const I *Ai, const double *Ax)
{
    I n = k->n, m = k->m, N = k->N, c, q, j;
    I *cnt = (I *)calloc((size_t)N + 1, sizeof(I));
    I nnz_diagP = 0, nnzK, pcol;
    /* column counts (directldl_kkt_assembly.jl:52-101) */
    for (j = 0; j < n; j++) {
        I len = Pp[j + 1] - Pp[j];
        int hasdiag = (len > 0 && Pi[Pp[j + 1] - 1] == j);
        cnt[j] += len + (hasdiag ? 0 : 1);       /* csc_assembly.jl:36-48 */
        nnz_diagP += hasdiag;
    }
    for (q = 0; q < Ap[n]; q++) cnt[n + Ai[q]] += 1;   /* A transposed: csc_assembly.jl:76-90 */
    pcol = n + m;
    for (c = 0; c < k->ncones; c++) {
        cone_t *K = &k->cones[c];
        I row = n + K->off, t;
        if (K->kind == ORC_PSD || (K->kind == ORC_SOC && !K->sparse)) {
            for (t = 0; t < K->numel; t++) cnt[row + t] += t + 1;   /* dense triu block */
        } else {
            for (t = 0; t < K->numel; t++) cnt[row + t] += 1;       /* diagonal block */
        }
        if (K->sparse) {
            cnt[pcol] += K->numel + 1;       /* v column + diag (directldl_datamaps.jl:24-40) */
            cnt[pcol + 1] += K->numel + 1;   /* u column + diag */
            pcol += 2;
        }
    }
    k->Kp = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->Kp[0] = 0;
    for (j = 0; j < N; j++) k->Kp[j + 1] = k->Kp[j] + cnt[j];
    nnzK = k->Kp[N];
    k->nnzK = nnzK;
    (void)nnz_diagP;
    k->Ki = (I *)malloc((size_t)(nnzK > 0 ? nnzK : 1) * sizeof(I));
    k->Kx = (double *)calloc((size_t)(nnzK > 0 ? nnzK : 1), sizeof(double));
    k->nnzP = Pp[n]; k->nnzA = Ap[n];
    k->mapP = (I *)malloc((size_t)(k->nnzP + 1) * sizeof(I));
    k->mapA = (I *)malloc((size_t)(k->nnzA + 1) * sizeof(I));
    k->mapHs = (I *)malloc((size_t)(k->nHs + 1) * sizeof(I));
    k->mapDiag = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->mapU = (I *)malloc((size_t)(k->sparse_len + 1) * sizeof(I));
    k->mapV = (I *)malloc((size_t)(k->sparse_len + 1) * sizeof(I));
    k->mapD = (I *)malloc((size_t)(2 * k->nsparse + 1) * sizeof(I));
    /* fill, using cnt as next-free pointer (directldl_kkt_assembly.jl:104-175) */
    for (j = 0; j < N; j++) cnt[j] = k->Kp[j];
    for (j = 0; j < n; j++) {
        for (q = Pp[j]; q < Pp[j + 1]; q++) {
            I dest = cnt[j]++;
            k->Ki[dest] = Pi[q]; k->Kx[dest] = Px[q]; k->mapP[q] = dest;
        }
        if (!(Pp[j + 1] > Pp[j] && Pi[Pp[j + 1] - 1] == j)) {   /* csc_assembly.jl:207-220 */
            I dest = cnt[j]++;
            k->Ki[dest] = j; k->Kx[dest] = 0.0;
        }
    }
    for (j = 0; j < n; j++)
        for (q = Ap[j]; q < Ap[j + 1]; q++) {
            I col = n + Ai[q], dest = cnt[col]++;
            k->Ki[dest] = j; k->Kx[dest] = Ax[q]; k->mapA[q] = dest;
        }
    pcol = n + m;
    for (c = 0; c < k->ncones; c++) {
        cone_t *K = &k->cones[c];
        I row = n + K->off, t, r, kidx = 0;
        I *block = k->mapHs + K->boff;
        if (K->kind == ORC_PSD || (K->kind == ORC_SOC && !K->sparse)) {
            for (t = 0; t < K->numel; t++)              /* csc_assembly.jl:160-173 */
                for (r = 0; r <= t; r++) {
                    I dest = cnt[row + t]++;
                    k->Ki[dest] = row + r; block[kidx++] = dest;
                }
        } else {
            for (t = 0; t < K->numel; t++) {            /* csc_assembly.jl:192-202 */
                I dest = cnt[row + t]++;
                k->Ki[dest] = row + t; block[t] = dest;
            }
        }
        if (K->sparse) {                                 /* directldl_datamaps.jl:42-59 */
            for (t = 0; t < K->numel; t++) {
                I dest = cnt[pcol]++;
                k->Ki[dest] = row + t; k->mapV[K->soff + t] = dest;
            }
            for (t = 0; t < K->numel; t++) {
                I dest = cnt[pcol + 1]++;
                k->Ki[dest] = row + t; k->mapU[K->soff + t] = dest;
            }
            for (t = 0; t < 2; t++) {
                I dest = cnt[pcol + t]++;
                k->Ki[dest] = pcol + t; k->mapD[2 * K->sidx + t] = dest;
            }
            pcol += 2;
        }
    }
    for (j = 0; j < N; j++) k->mapDiag[j] = k->Kp[j + 1] - 1;   /* :161-165 */
    free(cnt);
}

/* ------------------------------------------------------------ LDL engine */
/* symmetric permutation of a triu CSC into triu(P K P^T) with the entry map
 * (QDLDL.jl keeps this as AtoPAPt; call site directldl_qdldl.jl:46-68) */
static void permute_symmetric(orc_kkt *k)
{
    I N = k->N, j, q;
    I *cnt = (I *)calloc((size_t)N + 1, sizeof(I));
    for (j = 0; j < N; j++)
        for (q = k->Kp[j]; q < k->Kp[j + 1]; q++) {
            I i2 = k->iperm[k->Ki[q]], j2 = k->iperm[j];
            cnt[i2 > j2 ? i2 : j2]++;
        }
    k->Tp = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->Tp[0] = 0;
    for (j = 0; j < N; j++) k->Tp[j + 1] = k->Tp[j] + cnt[j];
    k->Ti = (I *)malloc((size_t)(k->nnzK + 1) * sizeof(I));
    k->Tx = (double *)calloc((size_t)(k->nnzK + 1), sizeof(double));
    k->AtoPAPt = (I *)malloc((size_t)(k->nnzK + 1) * sizeof(I));
    for (j = 0; j < N; j++) cnt[j] = k->Tp[j];
    for (j = 0; j < N; j++)
        for (q = k->Kp[j]; q < k->Kp[j + 1]; q++) {
            I i2 = k->iperm[k->Ki[q]], j2 = k->iperm[j];
            I col = i2 > j2 ? i2 : j2, row = i2 > j2 ? j2 : i2;
            I dest = cnt[col]++;
            k->Ti[dest] = row; k->Tx[dest] = k->Kx[q]; k->AtoPAPt[q] = dest;
        }
    free(cnt);
}

/* elimination tree and column counts of L */
static I ldl_etree(I N, const I *Ap, const I *Ai, I *work, I *Lnz, I *etree)
{
    I i, j, q, sum = 0;
    for (i = 0; i < N; i++) { work[i] = 0; Lnz[i] = 0; etree[i] = -1; }
    for (j = 0; j < N; j++) {
        work[j] = j;
        for (q = Ap[j]; q < Ap[j + 1]; q++) {
            i = Ai[q];
            while (work[i] != j) {
                if (etree[i] == -1) etree[i] = j;
                Lnz[i]++;
                work[i] = j;
                i = etree[i];
            }
        }
    }
    for (i = 0; i < N; i++) sum += Lnz[i];
    return sum;
}

/* up-looking numeric LDL^T with sign-guided dynamic regularisation
 * (kwargs at directldl_qdldl.jl:18-25) */
static void ldl_factor(orc_kkt *k)
{
    I N = k->N, i, j, q, kk;
    const I *Ap = k->Tp, *Ai = k->Ti; const double *Ax = k->Tx;
    I *yIdx = k->iwork, *elim = k->iwork + N, *Lnext = k->iwork + 2 * N;
    unsigned char *ymark = k->bwork;
    double *yv = k->fwork, *D = k->D, *Dinv = k->Dinv;
    double eps = k->st.dynamic_reg_eps, delta = k->st.dynamic_reg_delta;
    k->ndyn = 0;
    for (i = 0; i < N; i++) { ymark[i] = 0; yv[i] = 0.0; D[i] = 0.0; Lnext[i] = k->Lp[i]; }
    for (kk = 0; kk < N; kk++) {
        I nnzY = 0;
        for (q = Ap[kk]; q < Ap[kk + 1]; q++) {
            I b = Ai[q], nx, nnzE;
            if (b == kk) { D[kk] = Ax[q]; continue; }
            yv[b] = Ax[q];
            nx = b;
            if (!ymark[nx]) {
                ymark[nx] = 1; elim[0] = nx; nnzE = 1;
                nx = k->etree[b];
                while (nx != -1 && nx < kk) {
                    if (ymark[nx]) break;
                    ymark[nx] = 1; elim[nnzE++] = nx; nx = k->etree[nx];
                }
                while (nnzE) yIdx[nnzY++] = elim[--nnzE];
            }
        }
        for (i = nnzY - 1; i >= 0; i--) {
            I c = yIdx[i], tmp = Lnext[c];
            double yc = yv[c];
            for (j = k->Lp[c]; j < tmp; j++) yv[k->Li[j]] -= k->Lx[j] * yc;
            k->Li[tmp] = kk;
            k->Lx[tmp] = yc * Dinv[c];
            D[kk] -= yc * k->Lx[tmp];
            Lnext[c]++;
            yv[c] = 0.0; ymark[c] = 0;
        }
        if (D[kk] * (double)k->pdsigns[kk] < eps) {
            D[kk] = delta * (double)k->pdsigns[kk];
            k->ndyn++;
        }
        Dinv[kk] = 1.0 / D[kk];
    }
}

int orc_ldl_refactor(orc_kkt *k)
{
    I i;
    ldl_factor(k);
    for (i = 0; i < k->N; i++)
        if (!isfinite(k->Dinv[i])) return 0;      /* directldl_qdldl.jl:79 */
    return 1;
}

/* x = K_reg^{-1} b: permute, L, D, L^T, inverse permute (Appendix C item 5) */
void orc_ldl_solve(const orc_kkt *k, double *x, const double *b)
{
    I N = k->N, i, j;
    double *t = k->solvetmp;
    for (i = 0; i < N; i++) t[i] = b[k->perm[i]];
    for (i = 0; i < N; i++) {
        double ti = t[i];
        for (j = k->Lp[i]; j < k->Lp[i + 1]; j++) t[k->Li[j]] -= k->Lx[j] * ti;
    }
    for (i = 0; i < N; i++) t[i] *= k->Dinv[i];
    for (i = N - 1; i >= 0; i--) {
        double ti = t[i];
        for (j = k->Lp[i]; j < k->Lp[i + 1]; j++) ti -= k->Lx[j] * t[k->Li[j]];
        t[i] = ti;
    }
    for (i = 0; i < N; i++) x[k->perm[i]] = t[i];
}

/* value updates mirrored into the permuted copy (kktsolver_directldl.jl:130-188) */
static void update_values(orc_kkt *k, const I *idx, const double *vals, I cnt)
{
    I t;
    for (t = 0; t < cnt; t++) { k->Kx[idx[t]] = vals[t]; k->Tx[k->AtoPAPt[idx[t]]] = vals[t]; }
}
static void scale_values(orc_kkt *k, const I *idx, double scale, I cnt)
{
    I t;
    for (t = 0; t < cnt; t++) { k->Kx[idx[t]] *= scale; k->Tx[k->AtoPAPt[idx[t]]] *= scale; }
}

/* --------------------------------------------------------- small dense helpers */
static void svec_to_mat(double *M, const double *x, I n)   /* coneops_psdtrianglecone.jl:469-483 */
{
    const double is2 = 1.0 / sqrt(2.0);
    I idx = 0, r, c;
    for (c = 0; c < n; c++)
        for (r = 0; r <= c; r++) {
            if (r == c) M[r + c * n] = x[idx];
            else { M[r + c * n] = x[idx] * is2; M[c + r * n] = x[idx] * is2; }
            idx++;
        }
}
static void mat_to_svec(double *x, const double *M, I n)   /* :486-497 */
{
    const double is2 = 1.0 / sqrt(2.0);
    I idx = 0, r, c;
    for (c = 0; c < n; c++)
        for (r = 0; r <= c; r++) {
            x[idx] = (r == c) ? M[r + c * n] : (M[r + c * n] + M[c + r * n]) * is2;
            idx++;
        }
}
/* lower Cholesky in place (col-major), upper part zeroed; returns 0 on failure */
static int chol_lower(double *A, I n)
{
    I i, j, q;
    for (j = 0; j < n; j++) {
        double d = A[j + j * n];
        for (q = 0; q < j; q++) d -= A[j + q * n] * A[j + q * n];
        if (!(d > 0.0)) return 0;
        d = sqrt(d);
        A[j + j * n] = d;
        for (i = j + 1; i < n; i++) {
            double v = A[i + j * n];
            for (q = 0; q < j; q++) v -= A[i + q * n] * A[j + q * n];
            A[i + j * n] = v / d;
        }
        for (i = 0; i < j; i++) A[i + j * n] = 0.0;
    }
    return 1;
}
/* C = op(A) * op(B), n x n col-major; ta/tb = 1 for transpose */
static void mm(double *C, const double *A, int ta, const double *B, int tb, I n)
{
    I i, j, q;
    for (j = 0; j < n; j++)
        for (i = 0; i < n; i++) {
            double s = 0.0;
            for (q = 0; q < n; q++) {
                double a = ta ? A[q + i * n] : A[i + q * n];
                double b = tb ? B[j + q * n] : B[q + j * n];
                s += a * b;
            }
            C[i + j * n] = s;
        }
}
/* one-sided Jacobi SVD: A (n x n, col-major) = U diag(s) V^T.  A is overwritten by U*diag(s)
 * during the sweeps; stands in for LAPACK gesdd (utils/dense_algebra.jl:219). */
static void jacobi_svd(double *A, double *U, double *s, double *V, I n)
{
    I i, j, q, sweep;
    for (i = 0; i < n * n; i++) V[i] = 0.0;
    for (i = 0; i < n; i++) V[i + i * n] = 1.0;
    for (sweep = 0; sweep < 60; sweep++) {
        double off = 0.0;
        for (i = 0; i < n - 1; i++)
            for (j = i + 1; j < n; j++) {
                double a = 0, b = 0, c = 0, zeta, t, cs, sn;
                for (q = 0; q < n; q++) {
                    a += A[q + i * n] * A[q + i * n];
                    b += A[q + j * n] * A[q + j * n];
                    c += A[q + i * n] * A[q + j * n];
                }
                if (fabs(c) <= 1e-300 || fabs(c) <= 1e-17 * sqrt(a * b)) continue;
                off = fmax(off, fabs(c) / sqrt(a * b));
                zeta = (b - a) / (2.0 * c);
                t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                cs = 1.0 / sqrt(1.0 + t * t); sn = cs * t;
                for (q = 0; q < n; q++) {
                    double ai = A[q + i * n], aj = A[q + j * n];
                    A[q + i * n] = cs * ai - sn * aj; A[q + j * n] = sn * ai + cs * aj;
                    ai = V[q + i * n]; aj = V[q + j * n];
                    V[q + i * n] = cs * ai - sn * aj; V[q + j * n] = sn * ai + cs * aj;
                }
            }
        if (off < 1e-15) break;
    }
    for (j = 0; j < n; j++) {
        double nrm = 0.0;
        for (q = 0; q < n; q++) nrm += A[q + j * n] * A[q + j * n];
        nrm = sqrt(nrm);
        s[j] = nrm;
        for (q = 0; q < n; q++) U[q + j * n] = nrm > 0 ? A[q + j * n] / nrm : 0.0;
    }
    /* singular values in descending order, as LAPACK returns them (stable insertion sort of the triplets) */
    for (j = 1; j < n; j++)
        for (i = j; i > 0 && s[i] > s[i - 1]; i--) {
            double tv = s[i]; s[i] = s[i - 1]; s[i - 1] = tv;
            for (q = 0; q < n; q++) {
                tv = U[q + i * n]; U[q + i * n] = U[q + (i - 1) * n]; U[q + (i - 1) * n] = tv;
                tv = V[q + i * n]; V[q + i * n] = V[q + (i - 1) * n]; V[q + (i - 1) * n] = tv;
            }
        }
}
/* triu(A (x)_s A) for symmetric A, t x t col-major out (coneops_psdtrianglecone.jl:502-540) */
static void skron(double *out, const double *A, I n)
{
    const double s2 = sqrt(2.0);
    I t = tri(n), col = 0, l, kq, i, j;
    for (l = 0; l < n; l++)
        for (kq = 0; kq <= l; kq++) {
            I row = 0;
            int kl = (kq == l);
            for (j = 0; j < n && row <= col; j++) {
                double Ajl = A[j + l * n], Ajk = A[j + kq * n];
                for (i = 0; i <= j; i++) {
                    int ij = (i == j);
                    double v;
                    if (row > col) break;
                    if (!ij && !kl) v = A[i + kq * n] * Ajl + A[i + l * n] * Ajk;
                    else if (ij && !kl) v = s2 * Ajl * Ajk;
                    else if (!ij && kl) v = s2 * A[i + l * n] * Ajk;
                    else v = Ajl * Ajl;
                    out[row + col * t] = v;
                    row++;
                }
            }
            col++;
        }
}

/* ---------------------------------------------------------------- cone ops */
static double soc_residual(const double *z, I n)   /* coneops_socone.jl:415-419 */
{
    double nr = 0.0; I i;
    for (i = 1; i < n; i++) nr += z[i] * z[i];
    nr = sqrt(nr);
    return (z[0] - nr) * (z[0] + nr);
}
static double sqrt_soc_residual(const double *z, I n)   /* :421-425 */
{
    double r = soc_residual(z, n);
    return r > 0.0 ? sqrt(r) : 0.0;
}

static int soc_update_scaling(cone_t *K, const double *s, const double *z)   /* :75-154 */
{
    I n = K->numel, i;
    double zscale = sqrt_soc_residual(z, n), sscale = sqrt_soc_residual(s, n);
    double wscale, w1sq = 0.0, gamma, denom, *w = K->w, *lam = K->lam;
    if (zscale == 0.0 || sscale == 0.0) return 0;
    K->eta = sqrt(sscale / zscale);
    for (i = 0; i < n; i++) w[i] = s[i] / sscale;
    w[0] += z[0] / zscale;
    for (i = 1; i < n; i++) w[i] -= z[i] / zscale;
    wscale = sqrt_soc_residual(w, n);
    if (wscale == 0.0) return 0;
    for (i = 0; i < n; i++) w[i] /= wscale;
    for (i = 1; i < n; i++) w1sq += w[i] * w[i];
    w[0] = sqrt(1.0 + w1sq);
    gamma = 0.5 * wscale;
    lam[0] = gamma;
    denom = 1.0 / (s[0] / sscale + z[0] / zscale + 2.0 * gamma);
    for (i = 1; i < n; i++)
        lam[i] = (((gamma + z[0] / zscale) / sscale) * s[i] +
                  ((gamma + s[0] / sscale) / zscale) * z[i]) * denom;
    {
        double sc = sqrt(sscale * zscale);
        for (i = 0; i < n; i++) lam[i] *= sc;
    }
    if (K->sparse) {
        double alpha = 2.0 * w[0], wsq = w[0] * w[0] + w1sq, wsqinv = 1.0 / wsq;
        double u0, u1, v1;
        K->d = wsqinv / 2.0;
        u0 = sqrt(wsq - K->d);
        u1 = alpha / u0;
        v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
        K->u[0] = u0; K->v[0] = 0.0;
        for (i = 1; i < n; i++) { K->u[i] = u1 * w[i]; K->v[i] = v1 * w[i]; }
    }
    return 1;
}

static int psd_update_scaling(cone_t *K, const double *s, const double *z)   /* psd :78-143 */
{
    I n = K->dim, nn = n * n, i, j;
    double *S, *Z, *tmp, *U, *V, *sv, *work;
    if (K->numel == 0) return 1;
    work = (double *)malloc((size_t)(5 * nn + n) * sizeof(double));
    S = work; Z = work + nn; tmp = work + 2 * nn; U = work + 3 * nn; V = work + 4 * nn;
    sv = work + 5 * nn;
    svec_to_mat(S, s, n); svec_to_mat(Z, z, n);
    if (!chol_lower(S, n) || !chol_lower(Z, n)) { free(work); return 0; }
    mm(tmp, Z, 1, S, 0, n);             /* L2' * L1 */
    jacobi_svd(tmp, U, sv, V, n);
    for (i = 0; i < n; i++) K->plam[i] = sv[i];
    /* R = L1 * V * Lam^{-1/2};  Rinv = Lam^{-1/2} * U' * L2' */
    mm(K->R, S, 0, V, 0, n);
    for (j = 0; j < n; j++) {
        double sc = 1.0 / sqrt(sv[j]);
        for (i = 0; i < n; i++) K->R[i + j * n] *= sc;
    }
    mm(K->Rinv, U, 1, Z, 1, n);
    for (i = 0; i < n; i++) {
        double sc = 1.0 / sqrt(sv[i]);
        for (j = 0; j < n; j++) K->Rinv[i + j * n] *= sc;
    }
    mm(tmp, K->R, 0, K->R, 1, n);        /* R R' */
    /* symmetrise exactly as a triu-only syrk + Symmetric view would */
    for (j = 0; j < n; j++)
        for (i = j + 1; i < n; i++) tmp[i + j * n] = tmp[j + i * n];
    skron(K->Hs, tmp, n);
    free(work);
    return 1;
}

void orc_cones_set_identity_scaling(orc_kkt *k)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {
        cone_t *K = &k->cones[c];
        if (K->kind == ORC_NN) for (i = 0; i < K->numel; i++) K->w[i] = 1.0;
        else if (K->kind == ORC_SOC) {              /* coneops_socone.jl:56-73 */
            for (i = 0; i < K->numel; i++) K->w[i] = 0.0;
            K->w[0] = 1.0; K->eta = 1.0;
            if (K->sparse) {
                K->d = 0.5;
                for (i = 0; i < K->numel; i++) { K->u[i] = 0.0; K->v[i] = 0.0; }
                K->u[0] = sqrt(0.5);
            }
        } else if (K->kind == ORC_PSD) {            /* psd :65-75 */
            I n = K->dim, t = K->numel;
            memset(K->R, 0, (size_t)(n * n) * sizeof(double));
            memset(K->Rinv, 0, (size_t)(n * n) * sizeof(double));
            memset(K->Hs, 0, (size_t)(t * t) * sizeof(double));
            for (i = 0; i < n; i++) { K->R[i + i * n] = 1.0; K->Rinv[i + i * n] = 1.0; }
            for (i = 0; i < t; i++) K->Hs[i + i * t] = 1.0;
        }
    }
}

int orc_cones_update_scaling(orc_kkt *k, const double *s, const double *z)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {          /* coneops_compositecone.jl:103-120 */
        cone_t *K = &k->cones[c];
        const double *sc = s + K->off, *zc = z + K->off;
        if (K->kind == ORC_NN) {                /* coneops_nncone.jl:77-89 */
            for (i = 0; i < K->numel; i++) {
                K->lam[i] = sqrt(sc[i] * zc[i]);
                K->w[i] = sqrt(sc[i] / zc[i]);
            }
        } else if (K->kind == ORC_SOC) {
            if (!soc_update_scaling(K, sc, zc)) return 0;
        } else if (K->kind == ORC_PSD) {
            if (!psd_update_scaling(K, sc, zc)) return 0;
        }
    }
    return 1;
}

void orc_cones_get_Hs(const orc_kkt *k, double *Hs)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {          /* coneops_compositecone.jl:123-132 */
        const cone_t *K = &k->cones[c];
        double *blk = Hs + K->boff;
        if (K->kind == ORC_ZERO) {
            for (i = 0; i < K->numel; i++) blk[i] = 0.0;              /* zerocone :91-102 */
        } else if (K->kind == ORC_NN) {
            for (i = 0; i < K->numel; i++) blk[i] = K->w[i] * K->w[i]; /* nncone :91-101 */
        } else if (K->kind == ORC_SOC) {       /* socone :156-192 */
            double e2 = K->eta * K->eta;
            if (K->sparse) {
                for (i = 0; i < K->numel; i++) blk[i] = e2;
                blk[0] *= K->d;
            } else {
                I col, row, h = 1;
                blk[0] = (sqrt(2.0) * K->w[0] - 1.0) * (sqrt(2.0) * K->w[0] + 1.0);
                for (col = 1; col < K->numel; col++) {
                    double wc = K->w[col];
                    for (row = 0; row <= col; row++) blk[h++] = 2.0 * K->w[row] * wc;
                    blk[h - 1] += 1.0;
                }
                for (i = 0; i < K->blen; i++) blk[i] *= e2;
            }
        } else {                               /* psd :153-161, mathutils.jl:402-412 */
            I t = K->numel, col, row, q = 0;
            for (col = 0; col < t; col++)
                for (row = 0; row <= col; row++) blk[q++] = K->Hs[row + col * t];
        }
    }
}

void orc_cones_mul_Hs(const orc_kkt *k, double *y, const double *x)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {          /* coneops_compositecone.jl:138-150 */
        const cone_t *K = &k->cones[c];
        const double *xc = x + K->off; double *yc = y + K->off;
        if (K->kind == ORC_ZERO) for (i = 0; i < K->numel; i++) yc[i] = 0.0;
        else if (K->kind == ORC_NN) for (i = 0; i < K->numel; i++) yc[i] = K->w[i] * (K->w[i] * xc[i]);
        else if (K->kind == ORC_SOC) {         /* socone :201-216 */
            double cdot = 0.0, e2 = K->eta * K->eta;
            for (i = 0; i < K->numel; i++) cdot += K->w[i] * xc[i];
            cdot *= 2.0;
            for (i = 0; i < K->numel; i++) yc[i] = xc[i];
            yc[0] = -xc[0];
            for (i = 0; i < K->numel; i++) yc[i] = (yc[i] + cdot * K->w[i]) * e2;
        } else {                               /* psd :164-187, :409-437 */
            I n = K->dim, nn = n * n;
            double *X = (double *)malloc((size_t)(3 * nn) * sizeof(double));
            double *Y = X + nn, *T = X + 2 * nn;
            svec_to_mat(X, xc, n);
            mm(T, K->R, 1, X, 0, n); mm(Y, T, 0, K->R, 0, n);      /* R' X R */
            mm(T, Y, 0, K->R, 1, n); mm(X, K->R, 0, T, 0, n);      /* R Y R' */
            mat_to_svec(yc, X, n);
            free(X);
        }
    }
}

void orc_cones_soc_sparse(const orc_kkt *k, double *u, double *v, double *eta2, double *d)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {
        const cone_t *K = &k->cones[c];
        if (!K->sparse) continue;
        for (i = 0; i < K->numel; i++) { u[K->soff + i] = K->u[i]; v[K->soff + i] = K->v[i]; }
        eta2[K->sidx] = K->eta * K->eta;
        if (d) d[K->sidx] = K->d;
    }
}

void orc_cones_lambda(const orc_kkt *k, double *lam)
{
    I c, i;
    for (c = 0; c < k->ncones; c++) {
        const cone_t *K = &k->cones[c];
        double *l = lam + K->off;
        for (i = 0; i < K->numel; i++) l[i] = 0.0;
        if (K->kind == ORC_NN || K->kind == ORC_SOC) for (i = 0; i < K->numel; i++) l[i] = K->lam[i];
        else if (K->kind == ORC_PSD) for (i = 0; i < K->dim; i++) l[i] = K->plam[i];
    }
}

void orc_cones_psd_scaling(const orc_kkt *k, double *R, double *Rinv)
{
    I c, i, o = 0;
    for (c = 0; c < k->ncones; c++) {
        const cone_t *K = &k->cones[c];
        if (K->kind != ORC_PSD) continue;
        for (i = 0; i < K->dim * K->dim; i++) { R[o + i] = K->R[i]; Rinv[o + i] = K->Rinv[i]; }
        o += K->dim * K->dim;
    }
}

/* ---------------------------------------------------------- construction */
orc_kkt *orc_kkt_new(I n, I m, const I *Pp, const I *Pi, const double *Px,
                     const I *Ap, const I *Ai, const double *Ax,
                     I ncones, const int *kinds, const I *dims,
                     const I *perm, const orc_settings *settings)
{
    orc_kkt *k = (orc_kkt *)calloc(1, sizeof(orc_kkt));
    I c, off = 0, boff = 0, i, N;
    if (settings) k->st = *settings; else orc_default_settings(&k->st);
    k->n = n; k->m = m; k->ncones = ncones;
    k->cones = (cone_t *)calloc((size_t)(ncones > 0 ? ncones : 1), sizeof(cone_t));
    for (c = 0; c < ncones; c++) {            /* compositecone_type.jl:114-141 */
        cone_t *K = &k->cones[c];
        K->kind = kinds[c]; K->dim = dims[c];
        K->numel = (K->kind == ORC_PSD) ? tri(dims[c]) : dims[c];
        K->off = off; off += K->numel;
        K->sparse = (K->kind == ORC_SOC && K->dim > 4);
        K->boff = boff;
        if (K->kind == ORC_PSD || (K->kind == ORC_SOC && !K->sparse)) K->blen = tri(K->numel);
        else K->blen = K->numel;
        boff += K->blen;
        if (K->sparse) { K->sidx = k->nsparse++; K->soff = k->sparse_len; k->sparse_len += K->numel; }
        if (K->kind == ORC_NN || K->kind == ORC_SOC) {
            K->w = (double *)calloc((size_t)K->numel + 1, sizeof(double));
            K->lam = (double *)calloc((size_t)K->numel + 1, sizeof(double));
        }
        if (K->sparse) {
            K->u = (double *)calloc((size_t)K->numel, sizeof(double));
            K->v = (double *)calloc((size_t)K->numel, sizeof(double));
        }
        if (K->kind == ORC_PSD) {
            I nn = K->dim * K->dim, t = K->numel;
            K->R = (double *)calloc((size_t)nn + 1, sizeof(double));
            K->Rinv = (double *)calloc((size_t)nn + 1, sizeof(double));
            K->Hs = (double *)calloc((size_t)(t * t) + 1, sizeof(double));
            K->plam = (double *)calloc((size_t)K->dim + 1, sizeof(double));
        }
    }
    if (off != m) { orc_kkt_free(k); return NULL; }
    k->nHs = boff;
    k->p = 2 * k->nsparse;
    N = k->N = n + m + k->p;
    assemble_kkt(k, Pp, Pi, Px, Ap, Ai, Ax);
    /* Dsigns (kktsolver_directldl.jl:112-126; Dsigns(SOC)=(-1,1) datamaps.jl:21) */
    k->dsigns = (I *)malloc((size_t)(N + 1) * sizeof(I));
    for (i = 0; i < n; i++) k->dsigns[i] = 1;
    for (i = n; i < n + m; i++) k->dsigns[i] = -1;
    for (i = 0; i < k->nsparse; i++) { k->dsigns[n + m + 2 * i] = -1; k->dsigns[n + m + 2 * i + 1] = 1; }
    /* ordering */
    k->perm = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->iperm = (I *)malloc((size_t)(N + 1) * sizeof(I));
    if (perm) memcpy(k->perm, perm, (size_t)N * sizeof(I));
    else orc_min_degree(N, k->Kp, k->Ki, k->perm);
    for (i = 0; i < N; i++) k->iperm[k->perm[i]] = i;
    permute_symmetric(k);
    k->pdsigns = (I *)malloc((size_t)(N + 1) * sizeof(I));
    for (i = 0; i < N; i++) k->pdsigns[i] = k->dsigns[k->perm[i]];
    /* symbolic ("logical") factorisation */
    k->iwork = (I *)malloc((size_t)(3 * N + 1) * sizeof(I));
    k->bwork = (unsigned char *)malloc((size_t)N + 1);
    k->fwork = (double *)malloc((size_t)(N + 1) * sizeof(double));
    k->etree = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->Lnz = (I *)malloc((size_t)(N + 1) * sizeof(I));
    k->Lp = (I *)malloc((size_t)(N + 2) * sizeof(I));
    k->nnzL = ldl_etree(N, k->Tp, k->Ti, k->iwork, k->Lnz, k->etree);
    k->Lp[0] = 0;
    for (i = 0; i < N; i++) k->Lp[i + 1] = k->Lp[i] + k->Lnz[i];
    k->Li = (I *)malloc((size_t)(k->nnzL + 1) * sizeof(I));
    k->Lx = (double *)malloc((size_t)(k->nnzL + 1) * sizeof(double));
    k->D = (double *)calloc((size_t)N + 1, sizeof(double));
    k->Dinv = (double *)calloc((size_t)N + 1, sizeof(double));
    k->x = (double *)calloc((size_t)N + 1, sizeof(double));
    k->b = (double *)calloc((size_t)N + 1, sizeof(double));
    k->work1 = (double *)calloc((size_t)N + 1, sizeof(double));
    k->work2 = (double *)calloc((size_t)N + 1, sizeof(double));
    k->solvetmp = (double *)calloc((size_t)N + 1, sizeof(double));
    k->Hsvals = (double *)calloc((size_t)k->nHs + 1, sizeof(double));
    orc_cones_set_identity_scaling(k);
    return k;
}

void orc_kkt_free(orc_kkt *k)
{
    I c;
    if (!k) return;
    for (c = 0; c < k->ncones; c++) {
        cone_t *K = &k->cones[c];
        free(K->w); free(K->lam); free(K->u); free(K->v);
        free(K->R); free(K->Rinv); free(K->Hs); free(K->plam);
    }
    free(k->cones); free(k->Kp); free(k->Ki); free(k->Kx);
    free(k->mapP); free(k->mapA); free(k->mapHs); free(k->mapDiag);
    free(k->mapU); free(k->mapV); free(k->mapD); free(k->dsigns);
    free(k->perm); free(k->iperm); free(k->Tp); free(k->Ti); free(k->Tx); free(k->AtoPAPt);
    free(k->pdsigns); free(k->etree); free(k->Lnz); free(k->Lp); free(k->Li); free(k->Lx);
    free(k->D); free(k->Dinv); free(k->iwork); free(k->bwork); free(k->fwork);
    free(k->x); free(k->b); free(k->work1); free(k->work2); free(k->solvetmp); free(k->Hsvals);
    free(k);
}

void orc_kkt_sizes(const orc_kkt *k, I *out)
{
    out[0] = k->n; out[1] = k->m; out[2] = k->p; out[3] = k->N; out[4] = k->nnzK;
    out[5] = k->nHs; out[6] = k->nnzL; out[7] = k->nsparse; out[8] = k->sparse_len;
}
const I *orc_kkt_colptr(const orc_kkt *k) { return k->Kp; }
const I *orc_kkt_rowval(const orc_kkt *k) { return k->Ki; }
const double *orc_kkt_nzval(const orc_kkt *k) { return k->Kx; }
const I *orc_kkt_map_P(const orc_kkt *k) { return k->mapP; }
const I *orc_kkt_map_A(const orc_kkt *k) { return k->mapA; }
const I *orc_kkt_map_Hs(const orc_kkt *k) { return k->mapHs; }
const I *orc_kkt_map_diag_full(const orc_kkt *k) { return k->mapDiag; }
const I *orc_kkt_map_soc_u(const orc_kkt *k) { return k->mapU; }
const I *orc_kkt_map_soc_v(const orc_kkt *k) { return k->mapV; }
const I *orc_kkt_map_soc_D(const orc_kkt *k) { return k->mapD; }
const I *orc_kkt_dsigns(const orc_kkt *k) { return k->dsigns; }
const I *orc_kkt_perm(const orc_kkt *k) { return k->perm; }
double orc_kkt_last_regularizer(const orc_kkt *k) { return k->last_eps; }
I orc_kkt_last_ir_iters(const orc_kkt *k) { return k->last_ir; }
I orc_kkt_num_dyn_regularized(const orc_kkt *k) { return k->ndyn; }
const double *orc_ldl_Dinv(const orc_kkt *k) { return k->Dinv; }

/* ------------------------------------------------- update / regularise / refactor */
static int regularize_and_refactor(orc_kkt *k)      /* kktsolver_directldl.jl:247-294 */
{
    I N = k->N, i;
    int ok;
    double *diag_kkt = k->work1, *diag_shifted = k->work2;
    if (k->st.static_reg_enable) {
        double maxdiag = 0.0, eps;
        for (i = 0; i < N; i++) {
            diag_kkt[i] = k->Kx[k->mapDiag[i]];
            if (fabs(diag_kkt[i]) > maxdiag) maxdiag = fabs(diag_kkt[i]);
        }
        eps = k->st.static_reg_constant + k->st.static_reg_proportional * maxdiag;  /* :297-310 */
        for (i = 0; i < N; i++)
            diag_shifted[i] = diag_kkt[i] + (k->dsigns[i] == 1 ? eps : -eps);
        update_values(k, k->mapDiag, diag_shifted, N);
        k->last_eps = eps;
    }
    ok = orc_ldl_refactor(k);
    if (k->st.static_reg_enable)
        for (i = 0; i < N; i++) k->Kx[k->mapDiag[i]] = diag_kkt[i];   /* K only, not the LDL copy */
    return ok;
}

int orc_kkt_update_values(orc_kkt *k, const double *Hs, const double *soc_u,
                          const double *soc_v, const double *soc_eta2)
{
    I i, c;
    for (i = 0; i < k->nHs; i++) k->Hsvals[i] = -Hs[i];          /* :225-228 */
    update_values(k, k->mapHs, k->Hsvals, k->nHs);
    for (c = 0; c < k->ncones; c++) {                             /* :235-241, datamaps :61-79 */
        const cone_t *K = &k->cones[c];
        double e2, dd[2];
        if (!K->sparse) continue;
        e2 = soc_eta2[K->sidx];
        update_values(k, k->mapU + K->soff, soc_u + K->soff, K->numel);
        update_values(k, k->mapV + K->soff, soc_v + K->soff, K->numel);
        scale_values(k, k->mapU + K->soff, -e2, K->numel);
        scale_values(k, k->mapV + K->soff, -e2, K->numel);
        dd[0] = -e2; dd[1] = e2;
        update_values(k, k->mapD + 2 * K->sidx, dd, 2);
    }
    return regularize_and_refactor(k);
}

int orc_kkt_update(orc_kkt *k)
{
    double *Hs = (double *)malloc((size_t)(k->nHs + 1) * sizeof(double));
    double *u = (double *)malloc((size_t)(k->sparse_len + 1) * sizeof(double));
    double *v = (double *)malloc((size_t)(k->sparse_len + 1) * sizeof(double));
    double *e2 = (double *)malloc((size_t)(k->nsparse + 1) * sizeof(double));
    int ok;
    orc_cones_get_Hs(k, Hs);
    orc_cones_soc_sparse(k, u, v, e2, NULL);
    ok = orc_kkt_update_values(k, Hs, u, v, e2);
    free(Hs); free(u); free(v); free(e2);
    return ok;
}

void orc_kkt_update_P(orc_kkt *k, const double *Px) { update_values(k, k->mapP, Px, k->nnzP); }
void orc_kkt_update_A(orc_kkt *k, const double *Ax) { update_values(k, k->mapA, Ax, k->nnzA); }

/* ------------------------------------------------------------- solve + IR */
void orc_kkt_setrhs(orc_kkt *k, const double *rhsx, const double *rhsz)   /* :313-327 */
{
    I i;
    for (i = 0; i < k->n; i++) k->b[i] = rhsx[i];
    for (i = 0; i < k->m; i++) k->b[k->n + i] = rhsz[i];
    for (i = 0; i < k->p; i++) k->b[k->n + k->m + i] = 0.0;
}

double orc_kkt_residual(const orc_kkt *k, double *e, const double *b, const double *x)
{
    /* e = b - K_sym x with K stored triu (kktsolver_directldl.jl:455-466) */
    I N = k->N, j, q;
    double nrm = 0.0;
    for (j = 0; j < N; j++) e[j] = b[j];
    for (j = 0; j < N; j++) {
        double xj = x[j], acc = 0.0;
        for (q = k->Kp[j]; q < k->Kp[j + 1]; q++) {
            I i = k->Ki[q];
            double v = k->Kx[q];
            if (i == j) acc += v * xj;
            else { e[i] -= v * xj; acc += v * x[i]; }
        }
        e[j] -= acc;
    }
    for (j = 0; j < N; j++) {
        double a = fabs(e[j]);
        if (!(a <= nrm)) nrm = a;      /* propagates NaN like norm(e,Inf) */
    }
    return nrm;
}

static int iterative_refinement(orc_kkt *k)      /* kktsolver_directldl.jl:389-449 */
{
    I N = k->N, i;
    double *x = k->x, *b = k->b, *e = k->work1, *dx = k->work2, *tmp;
    double normb = 0.0, norme, lastnorme;
    int it;
    k->last_ir = 0;
    for (i = 0; i < N; i++) if (fabs(b[i]) > normb) normb = fabs(b[i]);
    norme = orc_kkt_residual(k, e, b, x);
    if (!isfinite(norme)) return 0;
    for (it = 0; it < k->st.ir_max_iter; it++) {
        double ratio;
        if (norme <= k->st.ir_abstol + k->st.ir_reltol * normb) break;
        lastnorme = norme;
        orc_ldl_solve(k, dx, e);
        for (i = 0; i < N; i++) dx[i] += x[i];
        norme = orc_kkt_residual(k, e, b, dx);
        k->last_ir++;
        if (!isfinite(norme)) return 0;
        ratio = lastnorme / norme;
        if (ratio < k->st.ir_stop_ratio) {
            if (ratio > 1.0) { tmp = x; x = dx; dx = tmp; }
            break;
        }
        tmp = x; x = dx; dx = tmp;
    }
    k->x = x; k->work2 = dx;
    return 1;
}

int orc_kkt_solve(orc_kkt *k, double *lhsx, double *lhsz)    /* :346-371 */
{
    I i;
    int ok;
    orc_ldl_solve(k, k->x, k->b);
    if (k->st.ir_enable) ok = iterative_refinement(k);
    else {
        ok = 1;
        for (i = 0; i < k->N; i++) if (!isfinite(k->x[i])) { ok = 0; break; }
    }
    if (ok) {
        if (lhsx) for (i = 0; i < k->n; i++) lhsx[i] = k->x[i];
        if (lhsz) for (i = 0; i < k->m; i++) lhsz[i] = k->x[k->n + i];
    }
    return ok;
}
