// Triangular solves on the supernodal tree (gfx950, wave64).
//
// Replaces QDLDL.solve! (call site /root/reference/src/kktsolvers/direct-ldl/directldl_qdldl.jl:
// 85-96): permute, L \, D^{-1}, L' \, inverse permute.
//
// Forward sweep, multifrontal style: each front gathers b and its children's contribution
// vectors, solves its unit-lower diagonal block, and leaves (what passed through) - L21*y for its
// parent.  No scatter conflicts, fixed summation order, hence bit-reproducible.  Backward sweep:
// each front gathers the ancestors' solution entries it needs.
//
// Two kernels per direction:
//   *_wave   fronts with f <= 64: one wave per front, everything in registers / readlane.
//   *_block  larger fronts: one 256-thread workgroup per front.  The diagonal block is staged in
//            LDS (packed lower triangle) and solved by ONE wave with v_readlane broadcasts (no
//            workgroup barrier per column); the off-diagonal panel is a dense GEMV streamed from
//            HBM/L2 with 8 independent loads in flight per lane.
#include "kernels.hpp"
#include "knobs.hpp"
#include "solve_common.hpp"
#include <algorithm>
#include <cstdlib>

namespace hipkkt {


// The static description of a front comes from its packed record (A.recs; rec_of) in one round of loads -- header and
// this lane's row slots side by side -- or, without records, from the legacy chain FrontDesc -> perm / rows / gl_ptr -> gl_src.
struct FrontLoc { int c0, nc, nb; int64_t rp, mat; };
__device__ __forceinline__ FrontLoc front_loc(const SolveHdr& h) { return FrontLoc{h.c0, h.nc, h.nb, h.rp, h.mat_off}; }

template <int NR>
__device__ __forceinline__ void fwd_wave_body(const SolveArgs& A, const RecSeg& R, int begin, int count, int bx, bool leaf = false)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = bx * (int)(blockDim.x >> 6) + wv;
    if (item >= count) return;
    const TreeDev& T = A.T;
    FrontLoc L;
    int pi = 0;
    RowGather G;
    G.cnt = 0;
    if (const char* rec = rec_of(A, R, 1, item)) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        pi = rec_idx(rec, lane);
        if (!leaf) G = rec_gather(rec, R.fmax[1], lane);
        L = front_loc(h);
    } else {
        const FrontDesc fd = T.desc[begin + item];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.front_off};
        pi = (lane < L.nc) ? T.perm[L.c0 + lane] : 0;
        if (!leaf && lane < L.nc + L.nb) G = row_gather_lists(T, (int64_t)L.c0 + L.rp + lane);
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int64_t rp = L.rp;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + L.mat;

    // gather: right-hand side entry plus the children's contributions to this row, in child order
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (lane < nc) ? A.b[c * A.ld_b + pi] : 0.0;
    if (!leaf && lane < f) gather_add<NR, false>(A, G, y);     // (a leaf has no gather lists to look at)
    // column sweep: y_l -= L(l,k) y_k
    for (int k0 = 0; k0 < nc; k0 += 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            lv[q] = (k < nc && lane > k && lane < f) ? F[lane + (int64_t)k * f] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            if (k < nc) {
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const double yk = readlane_f64(y[c], k);
                    y[c] = fma(-lv[q], yk, y[c]);
                }
            }
        }
    }
    if (lane < nc) stv<NR>(A.xp, c0 + lane, y);
    else if (lane < f) stv<NR>(A.uvec, rp + lane - nc, y);
}

template <int NR>
__device__ __forceinline__ void bwd_wave_body(const SolveArgs& A, const RecSeg& R, int begin, int count, int bx)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = bx * (int)(blockDim.x >> 6) + wv;
    if (item >= count) return;
    const TreeDev& T = A.T;
    FrontLoc L;
    int idx = 0;                   // own columns: the caller's row of the final store; rows below: the ancestor's entry of xp
    if (const char* rec = rec_of(A, R, 1, item)) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        idx = rec_idx(rec, lane);
        L = front_loc(h);
    } else {
        const FrontDesc fd = T.desc[begin + item];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.front_off};
        // (the store's row is fetched here, beside the other static loads: behind the sweep it was one more memory round
        //  trip at the end of every wave's life)
        idx = (lane < L.nc) ? T.perm[L.c0 + lane] : ((lane < L.nc + L.nb) ? T.rows[L.rp + lane - L.nc] : 0);
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + L.mat;

    // lane = row: y_r = D^{-1} x_r for the front's own columns, the ancestors' solution below
    double y[NR];
    {
        const double di = (lane < nc) ? A.Dinv[c0 + lane] : 0.0;
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c] = 0.0;
        if (lane < f) ldv<NR>(A.xp, lane < nc ? c0 + lane : idx, y);
        if (lane < nc) {
#pragma unroll
            for (int c = 0; c < NR; ++c) y[c] *= di;
        }
    }
    // x_j = y_j - sum_{r > j} L(r,j) x_r, j = nc-1 .. 0: column loads (coalesced over lanes) issued eight
    // at a time up front; one wave reduction per column
    for (int j1 = nc; j1 > 0; j1 -= 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j1 - 1 - q;
            lv[q] = (j >= 0 && lane > j && lane < f) ? F[lane + (int64_t)j * f] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j1 - 1 - q;
            if (j >= 0) {
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const double sum = wave_reduce_sum(lv[q] * y[c]);
                    if (lane == j) y[c] -= sum;
                }
            }
        }
    }
    if (lane < nc) {
        stv<NR>(A.xp, c0 + lane, y);
#pragma unroll
        for (int c = 0; c < NR; ++c) A.out[c * A.ld_out + idx] = y[c];
    }
}

// ------------------------------------------------------------------ tiny fronts, eight to a wave
// Most leaves of a KKT elimination tree are single columns with a handful of rows (cfg2: 82 000 of the
// 92 000 one-wave fronts have f <= 8).  A whole wave for each wastes 7/8 of the machine's wave slots, and
// these kernels are bound by how many waves are in flight: eight fronts share a wave, eight lanes each.
constexpr int kTinyFront = 8;
template <int NR>
__device__ __forceinline__ void fwd_tiny_body(const SolveArgs& A, const RecSeg& R, int begin, int count, int bx, bool leaf = false)
{
    const int sub = threadIdx.x & 7;                                 // row inside the front
    const int item = bx * (int)(blockDim.x >> 3) + (threadIdx.x >> 3);
    const bool live = item < count;
    const TreeDev& T = A.T;
    FrontLoc L;
    int pi = 0;
    RowGather G;
    G.cnt = 0;
    if (const char* rec = rec_of(A, R, 2, live ? item : count - 1)) {
        const SolveHdr* h = reinterpret_cast<const SolveHdr*>(rec);
        L = FrontLoc{h->c0, h->nc, h->nb, h->rp, h->mat_off};
        pi = rec_idx(rec, sub);
        if (!leaf) G = rec_gather(rec, R.fmax[2], sub);
    } else {
        const FrontDesc fd = T.desc[begin + (live ? item : count - 1)];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.front_off};
        pi = (sub < L.nc) ? T.perm[L.c0 + sub] : 0;
        if (!leaf && sub < L.nc + L.nb) G = row_gather_lists(T, (int64_t)L.c0 + L.rp + sub);
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int64_t rp = L.rp;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + L.mat;
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (sub < nc) ? A.b[c * A.ld_b + pi] : 0.0;
    if (!leaf && sub < f) gather_add<NR, false>(A, G, y);
    double lv[kTinyFront];
#pragma unroll
    for (int k = 0; k < kTinyFront; ++k) lv[k] = (k < nc && sub > k && sub < f) ? F[sub + k * f] : 0.0;
#pragma unroll
    for (int k = 0; k < kTinyFront; ++k) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double yk = __shfl(y[c], k, 8);                    // every lane takes part (no divergence here)
            if (k < nc) y[c] = fma(-lv[k], yk, y[c]);
        }
    }
    if (live) {
        if (sub < nc) stv<NR>(A.xp, c0 + sub, y);
        else if (sub < f) stv<NR>(A.uvec, rp + sub - nc, y);
    }
}
template <int NR>
__device__ __forceinline__ void bwd_tiny_body(const SolveArgs& A, const RecSeg& R, int begin, int count, int bx)
{
    const int sub = threadIdx.x & 7;
    const int item = bx * (int)(blockDim.x >> 3) + (threadIdx.x >> 3);
    const bool live = item < count;
    const TreeDev& T = A.T;
    FrontLoc L;
    int idx = 0;
    if (const char* rec = rec_of(A, R, 2, live ? item : count - 1)) {
        const SolveHdr* h = reinterpret_cast<const SolveHdr*>(rec);
        L = FrontLoc{h->c0, h->nc, h->nb, h->rp, h->mat_off};
        idx = rec_idx(rec, sub);
    } else {
        const FrontDesc fd = T.desc[begin + (live ? item : count - 1)];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.front_off};
        idx = (sub < L.nc) ? T.perm[L.c0 + sub] : ((sub < L.nc + L.nb) ? T.rows[L.rp + sub - L.nc] : 0);
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + L.mat;
    double y[NR];
    {
        const double di = (sub < nc) ? A.Dinv[c0 + sub] : 0.0;
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c] = 0.0;
        if (sub < f) ldv<NR>(A.xp, sub < nc ? c0 + sub : idx, y);
        if (sub < nc) {
#pragma unroll
            for (int c = 0; c < NR; ++c) y[c] *= di;
        }
    }
    double lv[kTinyFront];
#pragma unroll
    for (int j = 0; j < kTinyFront; ++j) lv[j] = (j < nc && sub > j && sub < f) ? F[sub + j * f] : 0.0;
#pragma unroll
    for (int j = kTinyFront - 1; j >= 0; --j) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double sum = group8_sum(lv[j] * y[c]);
            if (j < nc && sub == j) y[c] -= sum;
        }
    }
    if (live && sub < nc) {
        stv<NR>(A.xp, c0 + sub, y);
#pragma unroll
        for (int c = 0; c < NR; ++c) A.out[c * A.ld_out + idx] = y[c];
    }
}
// A level's one-wave and tiny fronts are independent of each other: one launch for both (the first nwb workgroups
// take the one-wave fronts [begin, begin + nwave), the others the tiny fronts behind them) saves a launch per sweep
// leaf: the fronts have no children (tree level 0), so no gather lists are read
template <int NR>
__global__ __launch_bounds__(256) void k_fwd_small(SolveArgs A, RecSeg R, int begin, int nwave, int ntiny, int leaf)
{
    const int nwb = (nwave + 3) >> 2;
    if ((int)blockIdx.x < nwb) fwd_wave_body<NR>(A, R, begin, nwave, blockIdx.x, leaf != 0);
    else fwd_tiny_body<NR>(A, R, begin + nwave, ntiny, blockIdx.x - nwb, leaf != 0);
}
template <int NR>
__global__ __launch_bounds__(256) void k_bwd_small(SolveArgs A, RecSeg R, int begin, int nwave, int ntiny)
{
    const int nwb = (nwave + 3) >> 2;
    if ((int)blockIdx.x < nwb) bwd_wave_body<NR>(A, R, begin, nwave, blockIdx.x);
    else bwd_tiny_body<NR>(A, R, begin + nwave, ntiny, blockIdx.x - nwb);
}

// ------------------------------------------------------------------ larger fronts, one block each
// After the factorisation k_winv forms, per supernode, the "solve matrix"
//        W = [ T ; M ],   T = L11^{-1} (unit lower, nc x nc),   M = L21 * T  (nb x nc),
// stored f x nc col-major.  With it a front's solve is ONE streaming matrix-vector phase per
// sweep (no substitution, no second dependent phase):
//   forward :  y <- gather ;  [x_s ; w] = W y_s ;           contribution = y_below - w
//   backward:  z = [D^{-1} y_s ; -x_below] ;  x_s = W' z
// (L^{-1} restricted to a front is [T 0; -M I], and its transpose gives the backward form.)
// Work is cut into items of 8 columns x 64 rows, 8 independent loads per lane in flight; partial
// sums are combined in a fixed order (bit-reproducible).
// LDS (doubles): forward NR * (1 + nks) * fpad  (y, then the partial sums per column slice),
//                backward NR * (fpad + nrs * ncpad); column c's share sits behind column c - 1's.

template <int BS, int NR>
__device__ __forceinline__ void fwd_block_body(const SolveArgs& A, const RecSeg& R, int begin, int bx)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    // the header and this thread's row slots in ONE round of loads (packed records), or the legacy chain
    const char* rec = rec_of(A, R, 0, bx);
    FrontLoc L;
    int pi = 0;
    RowGather G;
    G.cnt = 0;
    if (rec) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        if (tid < R.fmax[0]) { pi = rec_idx(rec, tid); G = rec_gather(rec, R.fmax[0], tid); }
        L = front_loc(h);
    } else {
        const FrontDesc fd = T.desc[begin + bx];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.w_off};
        if (tid < L.nc + L.nb) {
            pi = (tid < L.nc) ? T.perm[L.c0 + tid] : 0;
            G = row_gather_lists(T, (int64_t)L.c0 + L.rp + tid);
        }
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int64_t rp = L.rp;
    const int f = nc + nb;
    const double* __restrict__ W = A.tinv + L.mat;                     // f x nc, ld f
    const int fpad = (f + 3) & ~3;
    const int nks = (nc + 7) >> 3, nrb = (f + 63) >> 6;
    const int cst = (1 + nks) * fpad;        // LDS doubles per column
    double* y = smem;                        // column c: y at c * cst, its partial sums behind it
    double* part = smem + fpad;

    // The wave's FIRST batch of matrix items is fetched with the values of the gather: both depend on the header only.
    constexpr int U = kItemsInFlight;
    double m[U][8];
    auto load_items = [&](int it0) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NW;
            const int ks = it / nrb, rb = it - ks * nrb;
            const int r = rb * 64 + lane, k0 = 8 * ks;
            const bool live = it < nrb * nks && !(rb * 64 + 63 < k0);      // (wholly above T's diagonal: zeros)
#pragma unroll
            for (int q = 0; q < 8; ++q) m[u][q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
        }
    };
    load_items(wv);
    // gather: right-hand side entry plus the children's contributions to each row, in child order
    for (int i = tid; i < f; i += BS) {
        if (i != tid) {                       // (fronts taller than the workgroup: the further rows' slots)
            if (rec) { pi = rec_idx(rec, i); G = rec_gather(rec, R.fmax[0], i); }
            else { pi = (i < nc) ? T.perm[c0 + i] : 0; G = row_gather_lists(T, (int64_t)c0 + rp + i); }
        }
        double v[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) v[c] = (i < nc) ? A.b[c * A.ld_b + pi] : 0.0;
        gather_add<NR, false>(A, G, v);
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c * cst + i] = v[c];
    }
    __syncthreads();
    // kItemsInFlight items at a time: their loads are independent, so a wave keeps 8 x kItemsInFlight of them in flight
    for (int it0 = wv; it0 < nrb * nks; it0 += U * NW) {
        if (it0 != wv) load_items(it0);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NW;
            if (it < nrb * nks) {
                const int ks = it / nrb, rb = it - ks * nrb;
                const int r = rb * 64 + lane, k0 = 8 * ks;
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = fma(m[u][q], (k0 + q < nc) ? y[c * cst + k0 + q] : 0.0, acc);
                    if (r < f) part[c * cst + ks * fpad + r] = acc;
                }
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < f; i += BS) {
        double w[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double v = lds_sum_strided(part + c * cst + i, nks, fpad);
            w[c] = (i < nc) ? v : y[c * cst + i] - v;
        }
        if (i < nc) stv<NR>(A.xp, c0 + i, w);
        else stv<NR>(A.uvec, rp + i - nc, w);
    }
}

// backward items: 64 columns (lanes) x 8 rows of W' (= Wt, nc x f col-major), partial sums per row slice;
// column c's z / part at c * cst
struct BwdBatch { double m[kItemsInFlight][8]; };
// the batch of matrix items starting at it0 (the first one is fetched before z is built: bwd_block_body)
__device__ inline void bwd_load_items(BwdBatch& B, const double* __restrict__ Wt, int nc, int f, int it0, int NW, int lane)
{
    const int ncb = (nc + 63) >> 6, nrs = (f + 7) >> 3;
#pragma unroll
    for (int u = 0; u < kItemsInFlight; ++u) {
        const int it = it0 + u * NW;
        const int rs = it / ncb, cb = it - rs * ncb;
        const int j = cb * 64 + lane, r0 = 8 * rs;
        const bool live = it < ncb * nrs && !(r0 + 7 < cb * 64);       // (rows above the column block's diagonal: zeros)
#pragma unroll
        for (int q = 0; q < 8; ++q) B.m[u][q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
    }
}
template <int NR>
__device__ inline void bwd_items(BwdBatch& B, const double* __restrict__ Wt, int nc, int f, const double* z, double* part, int ncpad,
                                 int cst, int wv, int NW, int lane)
{
    const int ncb = (nc + 63) >> 6, nrs = (f + 7) >> 3;
    constexpr int U = kItemsInFlight;
    for (int it0 = wv; it0 < ncb * nrs; it0 += U * NW) {
        if (it0 != wv) bwd_load_items(B, Wt, nc, f, it0, NW, lane);
        double (&m)[U][8] = B.m;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int it = it0 + u * NW;
            if (it < ncb * nrs) {
                const int rs = it / ncb, cb = it - rs * ncb;
                const int j = cb * 64 + lane, r0 = 8 * rs;
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = fma(m[u][q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
                    if (j < nc) part[c * cst + rs * ncpad + j] = acc;
                }
            }
        }
    }
}

template <int BS, int NR>
__global__ __launch_bounds__(BS) void k_fwd_block(SolveArgs A, RecSeg R, int begin)
{
    fwd_block_body<BS, NR>(A, R, begin, blockIdx.x);
}

template <int BS, int NR>
__device__ __forceinline__ void bwd_block_body(const SolveArgs& A, const RecSeg& R, int begin, int bx)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    const char* rec = rec_of(A, R, 0, bx);
    FrontLoc L;
    int idx = 0;                  // own columns: the caller's row of the final store; rows below: the ancestor's entry of xp
    if (rec) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        if (tid < R.fmax[0]) idx = rec_idx(rec, tid);
        L = front_loc(h);
    } else {
        const FrontDesc fd = T.desc[begin + bx];
        L = FrontLoc{fd.c0, fd.nc, fd.nb, fd.rp, fd.w_off};
        if (tid < L.nc + L.nb) idx = (tid < L.nc) ? T.perm[L.c0 + tid] : T.rows[L.rp + tid - L.nc];
    }
    const int c0 = L.c0, nc = L.nc, nb = L.nb;
    const int64_t rp = L.rp;
    const int f = nc + nb;
    const double* __restrict__ Wt = A.tinv + L.mat + (int64_t)f * nc;          // W'(j, r) at j + r*nc
    const int fpad = (f + 3) & ~3, ncpad = (nc + 3) & ~3;
    const int nrs = (f + 7) >> 3;
    const int cst = fpad + nrs * ncpad;
    double* z = smem;
    double* part = smem + fpad;

    BwdBatch first;                           // (in flight while z is built)
    bwd_load_items(first, Wt, nc, f, wv, NW, lane);
    // z = [D^{-1} y_s ; -x_below]
    for (int i = tid; i < f; i += BS) {
        const double di = (i < nc) ? A.Dinv[c0 + i] : 0.0;
        int ri = idx;
        if (i != tid && i >= nc) ri = rec ? rec_idx(rec, i) : T.rows[rp + i - nc];
        double w[NR];
        ldv<NR>(A.xp, i < nc ? c0 + i : ri, w);
#pragma unroll
        for (int c = 0; c < NR; ++c) z[c * cst + i] = (i < nc) ? w[c] * di : -w[c];
    }
    __syncthreads();
    bwd_items<NR>(first, Wt, nc, f, z, part, ncpad, cst, wv, NW, lane);
    __syncthreads();
    for (int j = tid; j < nc; j += BS) {
        const int pi = j == tid ? idx : (rec ? rec_idx(rec, j) : T.perm[c0 + j]);
        double w[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double v = lds_sum_strided(part + c * cst + j, nrs, ncpad);
            w[c] = v;
            A.out[c * A.ld_out + pi] = v;
        }
        stv<NR>(A.xp, c0 + j, w);
    }
}
template <int BS, int NR>
__global__ __launch_bounds__(BS) void k_bwd_block(SolveArgs A, RecSeg R, int begin)
{
    bwd_block_body<BS, NR>(A, R, begin, blockIdx.x);
}
// A level's block-class, one-wave and tiny fronts are independent of each other: one launch for all three
// -- workgroups [0, nblock) take a block-class front each, the next ones BS/64 one-wave fronts
// each, the last ones BS/8 tiny fronts each.  Saves a launch (~5 us of pure latency) per sweep and level.
template <int BS, int NR>
__global__ __launch_bounds__(BS) void k_fwd_level(SolveArgs A, RecSeg R, int begin, int nblock, int nwave, int ntiny)
{
    const int bx = blockIdx.x, nwb = (nwave + BS / 64 - 1) / (BS / 64);
    if (bx < nblock) fwd_block_body<BS, NR>(A, R, begin, bx);
    else if (bx < nblock + nwb) fwd_wave_body<NR>(A, R, begin + nblock, nwave, bx - nblock);
    else fwd_tiny_body<NR>(A, R, begin + nblock + nwave, ntiny, bx - nblock - nwb);
}
template <int BS, int NR>
__global__ __launch_bounds__(BS) void k_bwd_level(SolveArgs A, RecSeg R, int begin, int nblock, int nwave, int ntiny)
{
    const int bx = blockIdx.x, nwb = (nwave + BS / 64 - 1) / (BS / 64);
    if (bx < nblock) bwd_block_body<BS, NR>(A, R, begin, bx);
    else if (bx < nblock + nwb) bwd_wave_body<NR>(A, R, begin + nblock, nwave, bx - nblock);
    else bwd_tiny_body<NR>(A, R, begin + nblock + nwave, ntiny, bx - nblock - nwb);
}

// ------------------------------------------------------------------ top of the tree, persistent
// The top levels of the assembly tree hold few fronts each (a chain of separators), so a launch per
// level is pure latency: ~10 us per level against ~3 us of work.  One launch covers them all:
// a workgroup per front, every workgroup resident (host guarantees count <= kTopMaxFronts), forward
// sweep then backward sweep inside the same kernel.  Fronts hand over through agent-scope flags
// (cdna_hip_programming.md Guideline 16, form R1 without fences): the payload (uvec / xp entries)
// is stored write-through (sc1 = relaxed agent-scope atomic stores), every storing wave drains
// vmcnt, the workgroup barriers, ONE lane stores the flag; the consumer polls the flag with
// relaxed agent-scope loads and reads the payload with agent-scope (L1-bypassing) loads only.
// Flags carry an epoch (a kernel argument, incremented per call), so nothing is re-zeroed.
// Every spin is bounded by wall clock; on expiry the abort word is set and everyone leaves.

// Everything that does not depend on other fronts is fetched BEFORE the flag wait and parked in
// registers: the wave's matrix items (up to PF per sweep), the row's gather-list indices, b, D^{-1},
// the ancestors' row indices.  After the flag only the handed-over values themselves are loaded.
// Two builds: 1024 threads with 4 items per wave parked (4 waves per SIMD, 128 VGPRs; 5 spill) -- the default: sixteen
// waves keep twice the loads in flight and park 64 items per workgroup (cfg2 0.303 -> 0.291 ms per solve, cfg3 0.263
// -> 0.246, cfg5 with its 1.2 MB fronts 1.56 -> 1.31); and 512 threads with 7 parked (2 waves per SIMD, ~177 VGPRs),
// kept selectable (HIPKKT_TOP_TALL=0).  NR right-hand sides: the parked matrix items and indices serve all of them;
// column c's LDS vectors sit at c * cst.
template <int BS, int kTopPF, int kTopPB, int NR>
__global__ __launch_bounds__(BS, BS == 1024 ? 4 : 2) void k_top_solve(SolveArgs A, int begin, int* flags, int epoch, int ntop, int nflag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    // Workgroup `me` owns the fronts at positions me, me + G, me + 2G, ... of the set (positions are in level order,
    // leaves first): forward in ascending, then backward in descending order.  Every task depends only on tasks that
    // come earlier in that global order and every workgroup is resident, so the earliest unfinished task can always
    // run -- no deadlock whatever the number of fronts.  A level of the set never holds more fronts than G (host), so
    // a workgroup has at most one front per level.
    const int me = blockIdx.x, G = gridDim.x;
    int* flag_f = flags;                 // forward done
    int* flag_b = flags + nflag;         // backward done (nflag >= ntop: the layout does not move with the set's size)
    int* abort_word = flags + 2 * nflag;
    const long long t0 = wall_clock64();
    const long long limit = A.top_limit; // (50 ms at 100 MHz: far beyond any real sweep)

#define TOP_STAMP(dir, slot) do { if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)(dir) * ntop + pos) * 8 + (slot)] = wall_clock64(); } while (0)
    // ================= forward =================
    int pos = me;
    for (; pos < ntop; pos += G) {
        TOP_STAMP(0, 0);
        const FrontDesc fd = T.desc[begin + pos];
        const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
        const int64_t rp = fd.rp;
        const int f = nc + nb;
        const double* __restrict__ W = A.tinv + fd.w_off;
        const int fpad = (f + 3) & ~3;
        // ---- preload
        const int nks = (nc + 7) >> 3, nrb = (f + 63) >> 6, nitF = nrb * nks;
        const int cst = (1 + nks) * fpad;
        double* y = smem;
        double* part = smem + fpad;
        ItemRegs rf[kTopPF];
#pragma unroll
        for (int p = 0; p < kTopPF; ++p) {
            const int it = wv + p * NW;
            const int ks = it / nrb, rb = it - ks * nrb;
            const int r = rb * 64 + lane, k0 = 8 * ks;
            const bool live = it < nitF && !(rb * 64 + 63 < k0);
#pragma unroll
            for (int q = 0; q < 8; ++q) rf[p].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
        }
        // gather sources per row whose indices are fetched before the wait: the separator rows near the top collect a
        // dozen small children each (fewer with several right-hand sides: the values of all columns are in flight together)
        constexpr int GP = NR == 1 ? 12 : (NR == 2 ? 8 : 4);
        int gsrc[GP];
#pragma unroll
        for (int q = 0; q < GP; ++q) gsrc[q] = -1;
        int64_t g0 = 0, g1 = 0;
        double bmine[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) bmine[c] = 0.0;
        if (tid < f) {
            const int64_t lc = (int64_t)c0 + rp + tid;
            g0 = T.gl_ptr[lc];
            g1 = T.gl_ptr[lc + 1];
#pragma unroll
            for (int q = 0; q < GP; ++q) gsrc[q] = (g0 + q < g1) ? T.gl_src[g0 + q] : -1;
            if (tid < nc) {
                const int pi = T.perm[c0 + tid];
#pragma unroll
                for (int c = 0; c < NR; ++c) bmine[c] = A.b[c * A.ld_b + pi];
            }
        }
        if (tid == 0) sh_ok = 1;
        __syncthreads();
        TOP_STAMP(0, 1);
        if (wv == 0) {
            // children inside the persistent set: poll their forward flags, lanes over children
            bool ok = true;
            for (int e = T.child_ptr[s] + lane; e < T.child_ptr[s + 1]; e += 64) {
                const int cp = T.spos[T.child_idx[e]] - begin;
                if (cp >= 0) ok = wait_flag(flag_f + cp, epoch, abort_word, t0, limit) && ok;
            }
            if (!ok) sh_ok = 0;
        }
        __syncthreads();
        if (!sh_ok) return;
        TOP_STAMP(0, 2);
        // ---- gather (only the handed-over values are loaded now)
        if (tid < f) {
            double u[NR][GP];              // every column's values in flight before the first sum
#pragma unroll
            for (int q = 0; q < GP; ++q)
#pragma unroll
                for (int c = 0; c < NR; ++c) u[c][q] = gsrc[q] >= 0 ? LD_AGENT_F64(A.uvec + (int64_t)(gsrc[q]) * NR + c) : 0.0;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                double v = bmine[c];
#pragma unroll
                for (int q = 0; q < GP; ++q) v += u[c][q];
                bmine[c] = v;
            }
            // rows with more sources than the parked ones (a few separator rows collect up to ~50): the same two load
            // rounds per GP sources -- indices, then all columns' values -- instead of two dependent loads per source;
            // the order of the sums is the list's order as before
            for (int64_t gb = g0 + GP; gb < g1; gb += GP) {
#pragma unroll
                for (int q = 0; q < GP; ++q) gsrc[q] = (gb + q < g1) ? T.gl_src[gb + q] : -1;
#pragma unroll
                for (int q = 0; q < GP; ++q)
#pragma unroll
                    for (int c = 0; c < NR; ++c) u[c][q] = gsrc[q] >= 0 ? LD_AGENT_F64(A.uvec + (int64_t)(gsrc[q]) * NR + c) : 0.0;
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double v = bmine[c];
#pragma unroll
                    for (int q = 0; q < GP; ++q) if (gsrc[q] >= 0) v += u[c][q];
                    bmine[c] = v;
                }
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) y[c * cst + tid] = bmine[c];
        }
        for (int i = tid + BS; i < f; i += BS) {          // fronts taller than the workgroup (rare)
            const int64_t lc = (int64_t)c0 + rp + i;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                double v = (i < nc) ? A.b[c * A.ld_b + T.perm[c0 + i]] : 0.0;
                for (int64_t g = T.gl_ptr[lc]; g < T.gl_ptr[lc + 1]; ++g) v += LD_AGENT_F64(A.uvec + (int64_t)(T.gl_src[g]) * NR + c);
                y[c * cst + i] = v;
            }
        }
        __syncthreads();
        TOP_STAMP(0, 3);
#pragma unroll
        for (int p = 0; p < kTopPF; ++p) {
            const int it = wv + p * NW;
            if (it < nitF) {
#pragma unroll
                for (int c = 0; c < NR; ++c) item_apply(rf[p], y + c * cst, f, nc, part + c * cst, fpad, it, nrb, lane);
            }
        }
        // (tall fronts: the items beyond the parked ones, three at a time -- 24 loads per lane in flight)
        for (int it0 = wv + kTopPF * NW; it0 < nitF; it0 += 3 * NW) {
            ItemRegs rr[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int it = it0 + u * NW;
                const int ks = it / nrb, rb = it - ks * nrb;
                const int r = rb * 64 + lane, k0 = 8 * ks;
                const bool live = it < nitF && !(rb * 64 + 63 < k0);
#pragma unroll
                for (int q = 0; q < 8; ++q) rr[u].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
                if (it0 + u * NW < nitF) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) item_apply(rr[u], y + c * cst, f, nc, part + c * cst, fpad, it0 + u * NW, nrb, lane);
                }
        }
        __syncthreads();
        TOP_STAMP(0, 4);
        for (int i = tid; i < f; i += BS) {
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const double v = lds_sum_strided(part + c * cst + i, nks, fpad);
                if (i < nc) ST_AGENT_F64(A.xp + (int64_t)(c0 + i) * NR + c, v);
                else ST_AGENT_F64(A.uvec + (int64_t)(rp + i - nc) * NR + c, y[c * cst + i] - v);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
        __syncthreads();
        TOP_STAMP(0, 5);
        if (tid == 0) {
            __hip_atomic_store(flag_f + pos, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (A.chain_cnt) A.chain_cnt[s] = 0;          // (its bottom children were swept by an earlier, chained kernel)
        }
    }

    // ================= backward =================
    for (pos -= G; pos >= 0; pos -= G) {
        const FrontDesc fd = T.desc[begin + pos];
        const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
        const int64_t rp = fd.rp;
        const int f = nc + nb;
        TOP_STAMP(1, 0);
        const double* __restrict__ Wt = A.tinv + fd.w_off + (int64_t)f * nc;
        const int fpad = (f + 3) & ~3, ncpad = (nc + 3) & ~3;
        // ---- preload
        const int ncb = (nc + 63) >> 6, nrs = (f + 7) >> 3, nitB = ncb * nrs;
        const int cst = fpad + nrs * ncpad;
        double* z = smem;
        double* part = smem + fpad;
        ItemRegs rbk[kTopPB];
#pragma unroll
        for (int p = 0; p < kTopPB; ++p) {
            const int it = wv + p * NW;
            const int rs = it / ncb, cb = it - rs * ncb;
            const int j = cb * 64 + lane, r0 = 8 * rs;
            const bool live = it < nitB && !(r0 + 7 < cb * 64);
#pragma unroll
            for (int q = 0; q < 8; ++q) rbk[p].m[q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
        }
        int ridx = -1;
        double dinv = 0.0;
        if (tid < nc) { dinv = A.Dinv[c0 + tid]; ridx = T.perm[c0 + tid]; }     // (own columns: the caller's row of the final store)
        else if (tid < f) ridx = T.rows[rp + tid - nc];
        if (tid == 0) sh_ok = 1;
        __syncthreads();
        TOP_STAMP(1, 1);
        if (wv == 0) {
            bool ok = true;
            const int par = T.sn_parent[s];
            if (lane == 0 && par >= 0) {
                const int pp = T.spos[par] - begin;        // the parent of a front of the set is in the set
                ok = wait_flag(flag_b + pp, epoch, abort_word, t0, limit);
            }
            if (!ok) sh_ok = 0;
        }
        __syncthreads();
        if (!sh_ok) return;
        TOP_STAMP(1, 2);
        {
            double zv[NR];                 // every column's value in flight before the first use
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const double* xc = A.xp + c;          // (columns interleaved: entry i of column c at i * NR + c)
                zv[c] = (tid < nc) ? LD_AGENT_F64(xc + (int64_t)(c0 + tid) * NR) : (tid < f ? LD_AGENT_F64(xc + (int64_t)ridx * NR) : 0.0);
            }
#pragma unroll
            for (int c = 0; c < NR; ++c)
                if (tid < f) z[c * cst + tid] = (tid < nc) ? zv[c] * dinv : -zv[c];
        }
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double* xc = A.xp + c;
            for (int i = tid + BS; i < f; i += BS)
                z[c * cst + i] = (i < nc) ? LD_AGENT_F64(xc + (int64_t)(c0 + i) * NR) * A.Dinv[c0 + i]
                                          : -LD_AGENT_F64(xc + (int64_t)T.rows[rp + i - nc] * NR);
        }
        __syncthreads();
        TOP_STAMP(1, 3);
#pragma unroll
        for (int p = 0; p < kTopPB; ++p) {
            const int it = wv + p * NW;
            if (it < nitB) {
                const int rs = it / ncb, cb = it - rs * ncb;
                const int j = cb * 64 + lane, r0 = 8 * rs;
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = fma(rbk[p].m[q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
                    if (j < nc) part[c * cst + rs * ncpad + j] = acc;
                }
            }
        }
        for (int it0 = wv + kTopPB * NW; it0 < nitB; it0 += 3 * NW) {
            double m[3][8];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int it = it0 + u * NW;
                const int rs = it / ncb, cb = it - rs * ncb;
                const int j = cb * 64 + lane, r0 = 8 * rs;
                const bool live = it < nitB && !(r0 + 7 < cb * 64);
#pragma unroll
                for (int q = 0; q < 8; ++q) m[u][q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int it = it0 + u * NW;
                if (it < nitB) {
                    const int rs = it / ncb, cb = it - rs * ncb;
                    const int j = cb * 64 + lane, r0 = 8 * rs;
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc = fma(m[u][q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
                        if (j < nc) part[c * cst + rs * ncpad + j] = acc;
                    }
                }
            }
        }
        __syncthreads();
        TOP_STAMP(1, 4);
        for (int j = tid; j < nc; j += BS) {
            const int pi = j == tid ? ridx : T.perm[c0 + j];       // (fetched before the wait: it sat on every hop's path)
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const double v = lds_sum_strided(part + c * cst + j, nrs, ncpad);
                ST_AGENT_F64(A.xp + (int64_t)(c0 + j) * NR + c, v);
                A.out[c * A.ld_out + pi] = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        TOP_STAMP(1, 5);
        if (tid == 0) __hip_atomic_store(flag_b + pos, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#undef TOP_STAMP

// ------------------------------------------------------------------ persistent kernel, sliced fronts
// The same hand-over scheme as k_top_solve, for sets with very TALL fronts (cfg5: 1531 x 96 panels, 1.2 MB of W per
// sweep through one CU).  Tasks are (front, slice) pairs (SolveArgs::tk_*): forward, slice sl of R owns the rows
// [R0, R1) of W -- the top nc rows plus the first share of the rows below for slice 0, a share of the rows below for
// the others -- and gathers the top nc entries of y itself; backward, it owns a run of columns of x and reads them
// from W with lanes over rows (x_j = sum_r W(r,j) z_r: per-lane partial sums over the wave's row blocks, one wave
// reduction per column, waves combined through LDS in a fixed order), building z itself, the top part from xf (the
// forward solution slice 0 left there) so that siblings may overwrite xp meanwhile.  The slices of a front never
// exchange anything; a parent waits for all slices of a child, a child for all slices of its parent.  As in
// k_top_solve everything static (matrix entries, gather indices, b, D^{-1}, row indices) is parked before the wait.
constexpr int kBdColsSolveMax = 160;     // columns of a front the tall kernels keep in LDS (panels have at most 144)
constexpr int kSlPF = 2;         // forward items per wave parked (the kernel must fit 128 VGPRs)
constexpr int kSlGP = 4;         // gather indices per row parked
// (front, slice) tasks: wave 0 waits for every task of the children of s inside the set.  Lanes over children for each
// child's FIRST task, then -- children cut into several tasks, one after the other -- lanes over the remaining tasks:
// a chain's only child has up to a dozen tasks, and polling them one by one cost a dozen L2 round trips per hop.
__device__ inline bool wait_children_tasks(const TreeDev& T, const int* __restrict__ tbase, int s, int begin, int pos0, int* flag,
                                           int epoch, int* abort_word, long long t0, long long limit, int lane)
{
    bool ok = true;
    const int ce1 = T.child_ptr[s + 1];
    for (int e0 = T.child_ptr[s]; e0 < ce1; e0 += 64) {
        int a = 0, b = 0;
        if (e0 + lane < ce1) {
            const int cp = T.spos[T.child_idx[e0 + lane]] - begin;
            if (cp >= pos0) { a = tbase[cp]; b = tbase[cp + 1]; }
        }
        if (b > a) ok = wait_flag(flag + a, epoch, abort_word, t0, limit) && ok;
        unsigned long long more = __ballot(b - a > 1);
        while (more) {
            const int c = __ffsll((long long)more) - 1;
            more &= more - 1;
            const int ca = __builtin_amdgcn_readlane(a, c), cb = __builtin_amdgcn_readlane(b, c);
            for (int q = ca + 1 + lane; q < cb; q += 64) ok = wait_flag(flag + q, epoch, abort_word, t0, limit) && ok;
        }
    }
    return ok;
}
// ... and for every task of the parent (backward), lanes over its tasks
__device__ inline bool wait_parent_tasks(const TreeDev& T, const int* __restrict__ tbase, int s, int begin, int* flag, int epoch,
                                         int* abort_word, long long t0, long long limit, int lane)
{
    bool ok = true;
    const int par = T.sn_parent[s];
    if (par >= 0) {
        const int pp = T.spos[par] - begin;        // the parent of a front of the set is in the set
        for (int q = tbase[pp] + lane; q < tbase[pp + 1]; q += 64) ok = wait_flag(flag + q, epoch, abort_word, t0, limit) && ok;
    }
    return ok;
}

// gather_rest for NR interleaved columns (entry e of column c at e * NR + c): GPB sources per pair of load rounds,
// every column's values in flight together, sums in list order
template <int GPB, int NR>
__device__ inline void gather_rest_nr(const TreeDev& T, const double* uvec, int64_t g, int64_t g1, double (&v)[NR])
{
    for (; g < g1; g += GPB) {
        int src[GPB];
        double u[NR][GPB];
#pragma unroll
        for (int q = 0; q < GPB; ++q) src[q] = (g + q < g1) ? T.gl_src[g + q] : -1;
#pragma unroll
        for (int q = 0; q < GPB; ++q)
#pragma unroll
            for (int c = 0; c < NR; ++c) u[c][q] = src[q] >= 0 ? LD_AGENT_F64(uvec + (int64_t)src[q] * NR + c) : 0.0;
#pragma unroll
        for (int c = 0; c < NR; ++c)
#pragma unroll
            for (int q = 0; q < GPB; ++q) if (src[q] >= 0) v[c] += u[c][q];
    }
}

// NR = 1 or 2 right-hand sides per sweep (column c of the work vectors interleaved: entry i at i * NR + c, as in
// k_top_solve): the parked matrix entries, gather indices and row indices serve both columns; column c's LDS vectors
// sit one column stride behind column c - 1's.
template <int BS, int NR>
__global__ __launch_bounds__(BS, 4) void k_top_solve_sliced(SolveArgs A, int begin, int pos0, int task0, int task1, int* flags,
                                                             int epoch, int nflag)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    const int me = blockIdx.x, G = gridDim.x;
    int* flag_f = flags;
    int* flag_b = flags + nflag;
    int* abort_word = flags + 2 * nflag;
    const long long t0 = wall_clock64();
    const long long limit = A.top_limit;

#define SL_STAMP(dir, slot) do { if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)(dir) * (task1 - task0) + (tk - task0)) * 8 + (slot)] = wall_clock64(); } while (0)
    // ================= forward =================
    int tk = task0 + me;
    for (; tk < task1; tk += G) {
        SL_STAMP(0, 0);
        const int pos = A.tk_pos[tk];
        const int slr = A.tk_sl[tk];
        const int sl = slr & 0xff, R = slr >> 8;
        if (R == 1) {
            // an ordinary front of the set: exactly k_top_solve's forward step (flags per task)
            const FrontDesc fd = T.desc[begin + pos];
            const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
            const int64_t rp = fd.rp;
            const int f = nc + nb;
            const double* __restrict__ W = A.tinv + fd.w_off;
            const int fpad = (f + 3) & ~3;
            // ---- preload
            const int nks = (nc + 7) >> 3, nrb = (f + 63) >> 6, nitF = nrb * nks;
            const int cst = (1 + nks) * fpad;
            double* y = smem;
            double* part = smem + fpad;
            ItemRegs rf[4];
    #pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int it = wv + p * NW;
                const int ks = it / nrb, rb = it - ks * nrb;
                const int r = rb * 64 + lane, k0 = 8 * ks;
                const bool live = it < nitF && !(rb * 64 + 63 < k0);
    #pragma unroll
                for (int q = 0; q < 8; ++q) rf[p].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
            }
            constexpr int GP = NR == 1 ? 12 : 8;   // gather sources per row whose indices are fetched before the wait: the
            int gsrc[GP];                          // separator rows near the top collect a dozen small children each
    #pragma unroll
            for (int q = 0; q < GP; ++q) gsrc[q] = -1;
            int64_t g0 = 0, g1 = 0;
            double bmine[NR];
    #pragma unroll
            for (int c = 0; c < NR; ++c) bmine[c] = 0.0;
            if (tid < f) {
                const int64_t lc = (int64_t)c0 + rp + tid;
                g0 = T.gl_ptr[lc];
                g1 = T.gl_ptr[lc + 1];
    #pragma unroll
                for (int q = 0; q < GP; ++q) gsrc[q] = (g0 + q < g1) ? T.gl_src[g0 + q] : -1;
                if (tid < nc) {
                    const int pi = T.perm[c0 + tid];
    #pragma unroll
                    for (int c = 0; c < NR; ++c) bmine[c] = A.b[c * A.ld_b + pi];
                }
            }
            if (tid == 0) sh_ok = 1;
            __syncthreads();
            if (wv == 0) {
                // children inside the persistent set: poll their forward flags, lanes over children
                if (!wait_children_tasks(T, A.tbase, s, begin, pos0, flag_f, epoch, abort_word, t0, limit, lane)) sh_ok = 0;
            }
            __syncthreads();
            if (!sh_ok) return;
            SL_STAMP(0, 2);
            // ---- gather (only the handed-over values are loaded now)
            if (tid < f) {
                double u[NR][GP];
    #pragma unroll
                for (int q = 0; q < GP; ++q)
    #pragma unroll
                    for (int c = 0; c < NR; ++c) u[c][q] = gsrc[q] >= 0 ? LD_AGENT_F64(A.uvec + (int64_t)gsrc[q] * NR + c) : 0.0;
    #pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double v = bmine[c];
    #pragma unroll
                    for (int q = 0; q < GP; ++q) v += u[c][q];
                    bmine[c] = v;
                }
                if (NR == 1) bmine[0] = gather_rest<8>(T, A.uvec, g0 + GP, g1, bmine[0]);
                else gather_rest_nr<8, NR>(T, A.uvec, g0 + GP, g1, bmine);
    #pragma unroll
                for (int c = 0; c < NR; ++c) y[c * cst + tid] = bmine[c];
            }
            for (int i = tid + BS; i < f; i += BS) {          // fronts taller than the workgroup (rare)
                const int64_t lc = (int64_t)c0 + rp + i;
    #pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double v = (i < nc) ? A.b[c * A.ld_b + T.perm[c0 + i]] : 0.0;
                    for (int64_t g = T.gl_ptr[lc]; g < T.gl_ptr[lc + 1]; ++g) v += LD_AGENT_F64(A.uvec + (int64_t)T.gl_src[g] * NR + c);
                    y[c * cst + i] = v;
                }
            }
            __syncthreads();
    #pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int it = wv + p * NW;
                if (it < nitF) {
    #pragma unroll
                    for (int c = 0; c < NR; ++c) item_apply(rf[p], y + c * cst, f, nc, part + c * cst, fpad, it, nrb, lane);
                }
            }
            // (tall fronts: the items beyond the parked ones, three at a time -- 24 loads per lane in flight)
            for (int it0 = wv + 4 * NW; it0 < nitF; it0 += 3 * NW) {
                ItemRegs rr[3];
    #pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int it = it0 + u * NW;
                    const int ks = it / nrb, rb = it - ks * nrb;
                    const int r = rb * 64 + lane, k0 = 8 * ks;
                    const bool live = it < nitF && !(rb * 64 + 63 < k0);
    #pragma unroll
                    for (int q = 0; q < 8; ++q) rr[u].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
                }
    #pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (it0 + u * NW < nitF) {
    #pragma unroll
                        for (int c = 0; c < NR; ++c) item_apply(rr[u], y + c * cst, f, nc, part + c * cst, fpad, it0 + u * NW, nrb, lane);
                    }
            }
            __syncthreads();
            for (int i = tid; i < f; i += BS) {
    #pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const double v = lds_sum_strided(part + c * cst + i, nks, fpad);
                    if (i < nc) ST_AGENT_F64(A.xp + (int64_t)(c0 + i) * NR + c, v);
                    else ST_AGENT_F64(A.uvec + (int64_t)(rp + i - nc) * NR + c, y[c * cst + i] - v);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
            __syncthreads();
            SL_STAMP(0, 5);
            if (tid == 0) {
                __hip_atomic_store(flag_f + tk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (A.chain_cnt) A.chain_cnt[s] = 0;
            }
            continue;
        }
        const FrontDesc fd = T.desc[begin + pos];
        const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
        const int64_t rp = fd.rp;
        const int f = nc + nb;
        const double* __restrict__ W = A.tinv + fd.w_off;
        const int rsz = (nb + R - 1) / R;
        const int r_lo = nc + sl * rsz;
        const int rs = max(0, min(rsz, f - r_lo));
        const int R0 = sl == 0 ? 0 : r_lo, R1 = r_lo + rs;          // this slice's rows of W
        const int nloc = R1 - R0, nlocp = (nloc + 3) & ~3, ncp = (nc + 3) & ~3;
        const int nks = (nc + 7) >> 3, nrb = (nloc + 63) >> 6, nit = nrb * nks;
        const int cst = ncp + nlocp + nks * nlocp;                   // one column's LDS vectors
        double* ytop = smem;                                         // y of the top nc rows
        double* yloc = smem + ncp;                                   // y of the slice's rows (slice 0: starts with the top ones)
        double* part = yloc + nlocp;
        // ---- parked: matrix items
        ItemRegs rf[kSlPF];
#pragma unroll
        for (int p = 0; p < kSlPF; ++p) {
            const int it = wv + p * NW;
            const int ks = it / nrb, rb = it - ks * nrb;
            const int lr = rb * 64 + lane, k0 = 8 * ks;
            const bool live = it < nit && lr < nloc;
#pragma unroll
            for (int q = 0; q < 8; ++q) rf[p].m[q] = (live && k0 + q < nc) ? W[(R0 + lr) + (int64_t)(k0 + q) * f] : 0.0;
        }
        // ---- parked: gather lists of the slice's rows (thread = row, two rows per thread at most) and, for the other
        // slices, of the top rows (threads 0 .. nc-1)
        int gs[2][kSlGP];
        int64_t g0[2] = {0, 0}, g1[2] = {0, 0};
        double bm[2][NR];
#pragma unroll
        for (int x = 0; x < 2; ++x) {
#pragma unroll
            for (int q = 0; q < kSlGP; ++q) gs[x][q] = -1;
#pragma unroll
            for (int c = 0; c < NR; ++c) bm[x][c] = 0.0;
            const int lr = tid + x * BS;
            if (lr < nloc) {
                const int row = R0 + lr;
                const int64_t lc = (int64_t)c0 + rp + row;
                g0[x] = T.gl_ptr[lc];
                g1[x] = T.gl_ptr[lc + 1];
#pragma unroll
                for (int q = 0; q < kSlGP; ++q) gs[x][q] = (g0[x] + q < g1[x]) ? T.gl_src[g0[x] + q] : -1;
                if (row < nc) {
                    const int pi = T.perm[c0 + row];
#pragma unroll
                    for (int c = 0; c < NR; ++c) bm[x][c] = A.b[c * A.ld_b + pi];
                }
            }
        }
        int gt[kSlGP];
        int64_t gt0 = 0, gt1 = 0;
        double bt[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) bt[c] = 0.0;
#pragma unroll
        for (int q = 0; q < kSlGP; ++q) gt[q] = -1;
        if (sl != 0 && tid < nc) {
            const int64_t lc = (int64_t)c0 + rp + tid;
            gt0 = T.gl_ptr[lc];
            gt1 = T.gl_ptr[lc + 1];
#pragma unroll
            for (int q = 0; q < kSlGP; ++q) gt[q] = (gt0 + q < gt1) ? T.gl_src[gt0 + q] : -1;
            const int pi = T.perm[c0 + tid];
#pragma unroll
            for (int c = 0; c < NR; ++c) bt[c] = A.b[c * A.ld_b + pi];
        }
        if (tid == 0) sh_ok = 1;
        __syncthreads();
        if (wv == 0) {
            if (!wait_children_tasks(T, A.tbase, s, begin, pos0, flag_f, epoch, abort_word, t0, limit, lane)) sh_ok = 0;
        }
        __syncthreads();
        if (!sh_ok) return;
        SL_STAMP(0, 2);
        // ---- gather: only the handed-over values are loaded now
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int lr = tid + x * BS;
            if (lr < nloc) {
                double u[NR][kSlGP];
#pragma unroll
                for (int q = 0; q < kSlGP; ++q)
#pragma unroll
                    for (int c = 0; c < NR; ++c) u[c][q] = gs[x][q] >= 0 ? LD_AGENT_F64(A.uvec + (int64_t)gs[x][q] * NR + c) : 0.0;
                double v[NR];
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    v[c] = bm[x][c];
#pragma unroll
                    for (int q = 0; q < kSlGP; ++q) v[c] += u[c][q];
                }
                if (NR == 1) v[0] = gather_rest<8>(T, A.uvec, g0[x] + kSlGP, g1[x], v[0]);
                else gather_rest_nr<8, NR>(T, A.uvec, g0[x] + kSlGP, g1[x], v);
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    yloc[c * cst + lr] = v[c];
                    if (sl == 0 && lr < nc) ytop[c * cst + lr] = v[c];
                }
            }
        }
        for (int lr = tid + 2 * BS; lr < nloc; lr += BS) {              // (slices taller than 2 x BS rows: none with R <= 8)
            const int row = R0 + lr;
            const int64_t lc = (int64_t)c0 + rp + row;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                double v = (row < nc) ? A.b[c * A.ld_b + T.perm[c0 + row]] : 0.0;
                for (int64_t g = T.gl_ptr[lc]; g < T.gl_ptr[lc + 1]; ++g) v += LD_AGENT_F64(A.uvec + (int64_t)T.gl_src[g] * NR + c);
                yloc[c * cst + lr] = v;
                if (sl == 0 && lr < nc) ytop[c * cst + lr] = v;
            }
        }
        if (sl != 0 && tid < nc) {
            double u[NR][kSlGP];
#pragma unroll
            for (int q = 0; q < kSlGP; ++q)
#pragma unroll
                for (int c = 0; c < NR; ++c) u[c][q] = gt[q] >= 0 ? LD_AGENT_F64(A.uvec + (int64_t)gt[q] * NR + c) : 0.0;
            double v[NR];
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                v[c] = bt[c];
#pragma unroll
                for (int q = 0; q < kSlGP; ++q) v[c] += u[c][q];
            }
            if (NR == 1) v[0] = gather_rest<8>(T, A.uvec, gt0 + kSlGP, gt1, v[0]);
            else gather_rest_nr<8, NR>(T, A.uvec, gt0 + kSlGP, gt1, v);
#pragma unroll
            for (int c = 0; c < NR; ++c) ytop[c * cst + tid] = v[c];
        }
        __syncthreads();
#pragma unroll
        for (int p = 0; p < kSlPF; ++p) {
            const int it = wv + p * NW;
            if (it < nit) {
                const int ks = it / nrb, rb = it - ks * nrb;
                const int lr = rb * 64 + lane, k0 = 8 * ks;
#pragma unroll
                for (int c = 0; c < NR; ++c) {
                    double acc = 0.0;
#pragma unroll
                    for (int q = 0; q < 8; ++q) acc = fma(rf[p].m[q], (k0 + q < nc) ? ytop[c * cst + k0 + q] : 0.0, acc);
                    if (lr < nloc) part[c * cst + ks * nlocp + lr] = acc;
                }
            }
        }
        for (int it0 = wv + kSlPF * NW; it0 < nit; it0 += 3 * NW) {
            double m[3][8];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int it = it0 + u * NW;
                const int ks = it / nrb, rb = it - ks * nrb;
                const int lr = rb * 64 + lane, k0 = 8 * ks;
                const bool live = it < nit && lr < nloc;
#pragma unroll
                for (int q = 0; q < 8; ++q) m[u][q] = (live && k0 + q < nc) ? W[(R0 + lr) + (int64_t)(k0 + q) * f] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int it = it0 + u * NW;
                if (it < nit) {
                    const int ks = it / nrb, rb = it - ks * nrb;
                    const int lr = rb * 64 + lane, k0 = 8 * ks;
#pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        double acc = 0.0;
#pragma unroll
                        for (int q = 0; q < 8; ++q) acc = fma(m[u][q], (k0 + q < nc) ? ytop[c * cst + k0 + q] : 0.0, acc);
                        if (lr < nloc) part[c * cst + ks * nlocp + lr] = acc;
                    }
                }
            }
        }
        __syncthreads();
        for (int lr = tid; lr < nloc; lr += BS) {
            const int row = R0 + lr;
#pragma unroll
            for (int c = 0; c < NR; ++c) {
                const double v = lds_sum_strided(part + c * cst + lr, nks, nlocp);
                if (row < nc) {
                    ST_AGENT_F64(A.xp + (int64_t)(c0 + row) * NR + c, v);
                    ST_AGENT_F64(A.xf + (int64_t)(c0 + row) * NR + c, v);
                } else {
                    ST_AGENT_F64(A.uvec + (int64_t)(rp + row - nc) * NR + c, yloc[c * cst + lr] - v);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SL_STAMP(0, 5);
        if (tid == 0) {
            __hip_atomic_store(flag_f + tk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (A.chain_cnt && sl == 0) A.chain_cnt[s] = 0;
        }
    }

    // ================= backward =================
    for (tk -= G; tk >= task0; tk -= G) {
        SL_STAMP(1, 0);
        const int pos = A.tk_pos[tk];
        const int slr = A.tk_sl[tk];
        const int sl = slr & 0xff, R = slr >> 8;
        if (R == 1) {
            // an ordinary front: exactly k_top_solve's backward step.  (Its own forward solution is still in xp: only
            // sliced fronts are overwritten by siblings.)
            const FrontDesc fd = T.desc[begin + pos];
            const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
            const int64_t rp = fd.rp;
            const int f = nc + nb;
            const double* __restrict__ Wt = A.tinv + fd.w_off + (int64_t)f * nc;
            const int fpad = (f + 3) & ~3, ncpad = (nc + 3) & ~3;
            // ---- preload
            const int ncb = (nc + 63) >> 6, nrs = (f + 7) >> 3, nitB = ncb * nrs;
            const int cst = fpad + nrs * ncpad;
            double* z = smem;
            double* part = smem + fpad;
            ItemRegs rbk[4];
    #pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int it = wv + p * NW;
                const int rs = it / ncb, cb = it - rs * ncb;
                const int j = cb * 64 + lane, r0 = 8 * rs;
                const bool live = it < nitB && !(r0 + 7 < cb * 64);
    #pragma unroll
                for (int q = 0; q < 8; ++q) rbk[p].m[q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
            }
            int ridx = -1;
            double dinv = 0.0;
            if (tid < nc) dinv = A.Dinv[c0 + tid];
            else if (tid < f) ridx = T.rows[rp + tid - nc];
            if (tid == 0) sh_ok = 1;
            __syncthreads();
            if (wv == 0) {
                if (!wait_parent_tasks(T, A.tbase, s, begin, flag_b, epoch, abort_word, t0, limit, lane)) sh_ok = 0;
            }
            __syncthreads();
            if (!sh_ok) return;
            SL_STAMP(1, 2);
            {
                double zv[NR];                 // every column's value in flight before the first use
    #pragma unroll
                for (int c = 0; c < NR; ++c)
                    zv[c] = (tid < nc) ? LD_AGENT_F64(A.xp + (int64_t)(c0 + tid) * NR + c)
                                       : (tid < f ? LD_AGENT_F64(A.xp + (int64_t)ridx * NR + c) : 0.0);
    #pragma unroll
                for (int c = 0; c < NR; ++c)
                    if (tid < f) z[c * cst + tid] = (tid < nc) ? zv[c] * dinv : -zv[c];
            }
    #pragma unroll
            for (int c = 0; c < NR; ++c)
                for (int i = tid + BS; i < f; i += BS)
                    z[c * cst + i] = (i < nc) ? LD_AGENT_F64(A.xp + (int64_t)(c0 + i) * NR + c) * A.Dinv[c0 + i]
                                              : -LD_AGENT_F64(A.xp + (int64_t)T.rows[rp + i - nc] * NR + c);
            __syncthreads();
    #pragma unroll
            for (int p = 0; p < 4; ++p) {
                const int it = wv + p * NW;
                if (it < nitB) {
                    const int rs = it / ncb, cb = it - rs * ncb;
                    const int j = cb * 64 + lane, r0 = 8 * rs;
    #pragma unroll
                    for (int c = 0; c < NR; ++c) {
                        double acc = 0.0;
    #pragma unroll
                        for (int q = 0; q < 8; ++q) acc = fma(rbk[p].m[q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
                        if (j < nc) part[c * cst + rs * ncpad + j] = acc;
                    }
                }
            }
            for (int it0 = wv + 4 * NW; it0 < nitB; it0 += 3 * NW) {
                double m[3][8];
    #pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int it = it0 + u * NW;
                    const int rs = it / ncb, cb = it - rs * ncb;
                    const int j = cb * 64 + lane, r0 = 8 * rs;
                    const bool live = it < nitB && !(r0 + 7 < cb * 64);
    #pragma unroll
                    for (int q = 0; q < 8; ++q) m[u][q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
                }
    #pragma unroll
                for (int u = 0; u < 3; ++u) {
                    const int it = it0 + u * NW;
                    if (it < nitB) {
                        const int rs = it / ncb, cb = it - rs * ncb;
                        const int j = cb * 64 + lane, r0 = 8 * rs;
    #pragma unroll
                        for (int c = 0; c < NR; ++c) {
                            double acc = 0.0;
    #pragma unroll
                            for (int q = 0; q < 8; ++q) acc = fma(m[u][q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
                            if (j < nc) part[c * cst + rs * ncpad + j] = acc;
                        }
                    }
                }
            }
            __syncthreads();
            for (int j = tid; j < nc; j += BS) {
                const int pi = T.perm[c0 + j];
    #pragma unroll
                for (int c = 0; c < NR; ++c) {
                    const double v = lds_sum_strided(part + c * cst + j, nrs, ncpad);
                    ST_AGENT_F64(A.xp + (int64_t)(c0 + j) * NR + c, v);
                    A.out[c * A.ld_out + pi] = v;
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            SL_STAMP(1, 5);
            if (tid == 0) __hip_atomic_store(flag_b + tk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

            continue;
        }
        const FrontDesc fd = T.desc[begin + pos];
        const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
        const int64_t rp = fd.rp;
        const int f = nc + nb;
        const double* __restrict__ W = A.tinv + fd.w_off;
        const int csz = (nc + R - 1) / R;
        const int j0 = sl * csz, ncl = max(0, min(csz, nc - j0));      // this slice's columns of x
        const int fpad = (f + 3) & ~3;
        const int cst = fpad + NW * 16;                                 // one right-hand side's LDS vectors
        double* z = smem;
        double* part = smem + fpad;                                     // NW x 16
        const int nrb = (f + 63) >> 6;
        // ---- parked: the first 16-column chunk's entries of the wave's first row block; row indices; D^{-1}
        constexpr int CW = 8;                                           // columns per pass (register budget)
        double pm[CW];
        {
            const int r = wv * 64 + lane;
#pragma unroll
            for (int c = 0; c < CW; ++c) pm[c] = (c < ncl && r < f) ? W[r + (int64_t)(j0 + c) * f] : 0.0;
        }
        int ridx[2] = {-1, -1};
        double dinv[2] = {0.0, 0.0};
#pragma unroll
        for (int x = 0; x < 2; ++x) {
            const int i = tid + x * BS;
            if (i < nc) dinv[x] = A.Dinv[c0 + i];
            else if (i < f) ridx[x] = T.rows[rp + i - nc];
        }
        if (tid == 0) sh_ok = 1;
        __syncthreads();
        if (wv == 0) {
            if (!wait_parent_tasks(T, A.tbase, s, begin, flag_b, epoch, abort_word, t0, limit, lane)) sh_ok = 0;
        }
        __syncthreads();
        if (!sh_ok) return;
        SL_STAMP(1, 2);
        {
            double zv[2][NR];              // every value in flight before the first use
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int i = tid + x * BS;
#pragma unroll
                for (int c = 0; c < NR; ++c)
                    zv[x][c] = (i < nc) ? LD_AGENT_F64(A.xf + (int64_t)(c0 + i) * NR + c)
                                        : (i < f ? LD_AGENT_F64(A.xp + (int64_t)ridx[x] * NR + c) : 0.0);
            }
#pragma unroll
            for (int x = 0; x < 2; ++x) {
                const int i = tid + x * BS;
#pragma unroll
                for (int c = 0; c < NR; ++c)
                    if (i < f) z[c * cst + i] = (i < nc) ? zv[x][c] * dinv[x] : -zv[x][c];
            }
        }
#pragma unroll
        for (int c = 0; c < NR; ++c)
            for (int i = tid + 2 * BS; i < f; i += BS)
                z[c * cst + i] = (i < nc) ? LD_AGENT_F64(A.xf + (int64_t)(c0 + i) * NR + c) * A.Dinv[c0 + i]
                                          : -LD_AGENT_F64(A.xp + (int64_t)T.rows[rp + i - nc] * NR + c);
        __syncthreads();
        for (int cc = 0; cc < ncl; cc += CW) {
            const int ncc = min(CW, ncl - cc);
            double acc[NR][CW];
#pragma unroll
            for (int b = 0; b < NR; ++b)
#pragma unroll
                for (int c = 0; c < CW; ++c) acc[b][c] = 0.0;
            int x = 0;
            for (int rb = wv; rb < nrb; rb += NW, ++x) {
                const int r = rb * 64 + lane;
                double zl[NR];
#pragma unroll
                for (int b = 0; b < NR; ++b) zl[b] = r < f ? z[b * cst + r] : 0.0;
                if (cc == 0 && x == 0) {
#pragma unroll
                    for (int b = 0; b < NR; ++b)
#pragma unroll
                        for (int c = 0; c < CW; ++c) acc[b][c] = fma(pm[c], zl[b], acc[b][c]);
                } else {
                    double m[CW];
#pragma unroll
                    for (int c = 0; c < CW; ++c) m[c] = (c < ncc && r < f) ? W[r + (int64_t)(j0 + cc + c) * f] : 0.0;
#pragma unroll
                    for (int b = 0; b < NR; ++b)
#pragma unroll
                        for (int c = 0; c < CW; ++c) acc[b][c] = fma(m[c], zl[b], acc[b][c]);
                }
            }
#pragma unroll
            for (int b = 0; b < NR; ++b)
#pragma unroll
                for (int c = 0; c < CW; ++c) {
                    const double sum = wave_reduce_sum(acc[b][c]);
                    if (lane == 0) part[b * cst + wv * 16 + c] = sum;
                }
            __syncthreads();
            if (tid < ncc * NR) {
                const int b = tid / ncc, jj = tid - b * ncc;
                double v = 0.0;
                for (int w = 0; w < NW; ++w) v += part[b * cst + w * 16 + jj];
                const int j = j0 + cc + jj;
                ST_AGENT_F64(A.xp + (int64_t)(c0 + j) * NR + b, v);
                A.out[b * A.ld_out + T.perm[c0 + j]] = v;
            }
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        SL_STAMP(1, 5);
            if (tid == 0) __hip_atomic_store(flag_b + tk, epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
#undef SL_STAMP

// ------------------------------------------------------------------ W = [L11^{-1} ; L21 L11^{-1}]
// Runs once after the factorisation, all supernodes in parallel (off the critical path of the tree).
// L11 (unit lower, nc x nc) is staged in LDS; T = L11^{-1} is built in 16 x 16 blocks:
//   1. every diagonal block is inverted by 16 threads (thread = column, forward substitution), in place;
//   2. the blocks below the diagonal by the block recurrence
//          T(k,b) = -T_k * sum_{j=b..k-1} L(k,j) T(j,b),     k = b+1, b+2, ...
//      one WAVE per block column b (the columns are independent of each other), every product on
//      v_mfma_f64_16x16x4_f64 with operands from LDS; T(k,b) goes to the transposed slot (b,k) in the unused upper
//      triangle, so that the L(k,j) the other block columns still need stay where they are.  The accumulator layout
//      of the first product (row = lane>>4 + 4q) is exactly the B operand layout of the second (k = 4t + lane>>4):
//      no exchange between the two.
// Then M = L21 * T, a wave per strip of 16 rows against T in LDS (A operands of a strip fetched ahead of the products).
constexpr int kWinvThreads = 512;
constexpr int kWinvSmallThreads = 128, kWinvSmallNc = 32;     // narrow supernodes: k_winv's small build (below)
template <int NT>
__device__ inline void winv_one(const TreeDev& T, const double* __restrict__ fronts, double* __restrict__ wst, int s,
                                double* smem)
{
    typedef double d4_t __attribute__((ext_vector_type(4)));
    constexpr int NWV = NT / 64;
    const int tid = threadIdx.x;
    const int lane = tid & 63, wvi = tid >> 6;
    const int ml = lane & 15, mk = lane >> 4;
    const int c0 = T.sn_start[s];
    const int nc = T.sn_start[s + 1] - c0;
    const int nb = (int)(T.rowptr[s + 1] - T.rowptr[s]);
    const int f = nc + nb;
    const double* __restrict__ F = fronts + T.front_off[s];
    double* __restrict__ W = wst + T.tinv_off[s];
    const int ld = nc | 1;
    const int nblk = (nc + 15) >> 4;
    double* Ls = smem;                 // entry (i,k), i > k, at i*ld + k
    // scratch / off-diagonal T(i,c), i > c, lives in the unused upper triangle at the transposed slot c*ld + i
#define WK(i, c) Ls[(c) * ld + (i)]
    for (int idx = tid; idx < nc * nc; idx += NT) {
        const int k = idx / nc, i = idx - k * nc;
        if (i > k) Ls[i * ld + k] = F[i + (int64_t)k * f];
    }
    __syncthreads();
    // ---- 1. diagonal 16 x 16 blocks: thread (block b, column j) solves L_bb t = e_j in registers
    {
        for (int item = tid; item < nblk * 16; item += NT) {
            const int b = item >> 4, j = item & 15;
            const int o = 16 * b, w = min(16, nc - o);
            if (j >= w) continue;
            double t[16];
#pragma unroll
            for (int i = 0; i < 16; ++i) t[i] = 0.0;
#pragma unroll
            for (int i = 1; i < 16; ++i) {
                if (i > j && i < w) {
                    double acc = Ls[(o + i) * ld + o + j];
#pragma unroll
                    for (int k = 1; k < 16; ++k)
                        if (k > j && k < i) acc = fma(Ls[(o + i) * ld + o + k], t[k], acc);
                    t[i] = -acc;
                }
            }
#pragma unroll
            for (int i = 1; i < 16; ++i)
                if (i > j && i < w) WK(o + i, o + j) = t[i];
        }
        __syncthreads();
        for (int item = tid; item < nblk * 256; item += NT) {
            const int b = item >> 8, i = (item >> 4) & 15, j = item & 15;
            const int o = 16 * b;
            if (i > j && o + i < nc) Ls[(o + i) * ld + o + j] = WK(o + i, o + j);
        }
        __syncthreads();
    }
    // ---- 2. block column b of T, top to bottom (wave = block column)
    for (int b = wvi; b + 1 < nblk; b += NWV) {
        const int ob = 16 * b;
        for (int k = b + 1; k < nblk; ++k) {
            const int ok = 16 * k;
            const int ri = ok + ml;                    // row of block k this lane feeds as A
            d4_t S = (d4_t){0.0, 0.0, 0.0, 0.0};
            for (int j = b; j < k; ++j) {
                const int oj = 16 * j;
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int kk = 4 * t + mk;         // index inside block j
                    const double a = (ri < nc) ? Ls[ri * ld + oj + kk] : 0.0;         // L(k,j)[ml][kk]  (oj + kk < ok <= ri)
                    double bv;                                                        // T(j,b)[kk][ml]
                    if (j == b) bv = (kk > ml) ? Ls[(ob + kk) * ld + ob + ml] : (kk == ml ? 1.0 : 0.0);
                    else bv = WK(oj + kk, ob + ml);
                    S = __builtin_amdgcn_mfma_f64_16x16x4f64(a, bv, S, 0, 0, 0);
                }
            }
            d4_t R = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int kk = 4 * t + mk;
                double a = 0.0;                                                       // T_k[ml][kk], unit lower
                if (ri < nc && ok + kk < nc) a = (ml > kk) ? Ls[ri * ld + ok + kk] : (ml == kk ? 1.0 : 0.0);
                R = __builtin_amdgcn_mfma_f64_16x16x4f64(a, S[t], R, 0, 0, 0);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int i = ok + mk + 4 * q;
                if (i < nc) WK(i, ob + ml) = -R[q];
            }
        }
    }
    __syncthreads();
    // T(i,j), i > j: inside a diagonal block in place, below it in the transposed slot
#define TGET(i, j) ((((i) ^ (j)) & ~15) == 0 ? Ls[(i) * ld + (j)] : WK(i, j))
    // ---- 3. T part of W (unit diagonal, zeros above), and of its transpose copy Wt (nc x f)
    double* __restrict__ Wt = W + (int64_t)f * nc;
    for (int idx = tid; idx < nc * nc; idx += NT) {
        const int j = idx / nc, i = idx - j * nc;
        W[i + (int64_t)j * f] = (i > j) ? TGET(i, j) : (i == j ? 1.0 : 0.0);
    }
    for (int idx = tid; idx < nc * nc; idx += NT) {
        const int i = idx / nc, j = idx - i * nc;
        Wt[j + (int64_t)i * nc] = (i > j) ? TGET(i, j) : (i == j ? 1.0 : 0.0);
    }
    // ---- 4. M = L21 * T on the matrix cores (v_mfma_f64_16x16x4_f64): a wave owns a strip of 16 rows
    //         and all column tiles; A = L21 straight from global (16 contiguous rows per k, a batch of k's in flight),
    //         B = T from LDS with its unit diagonal / zero upper part generated on the fly; column tiles right of k are skipped.
    {
        const int nct = nblk;                           // column tiles (<= 9 for nc <= 144)
        constexpr int KB = 12;                          // k-steps (of 4) fetched together
        for (int strip = wvi; strip * 16 < nb; strip += NWV) {
            const int r = strip * 16 + ml;
            const double* __restrict__ Lr = F + nc + r;
            d4_t acc[9];
#pragma unroll
            for (int jt = 0; jt < 9; ++jt) acc[jt] = (d4_t){0.0, 0.0, 0.0, 0.0};
            for (int kb0 = 0; kb0 < nc; kb0 += 4 * KB) {
                double av[KB];
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k = kb0 + 4 * u + mk;
                    av[u] = (r < nb && k < nc) ? Lr[(int64_t)k * f] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < KB; ++u) {
                    const int k0 = kb0 + 4 * u;
                    if (k0 < nc) {
                        const int k = k0 + mk;
#pragma unroll
                        for (int jt = 0; jt < 9; ++jt) {
                            if (jt < nct && 16 * jt <= k0 + 3) {
                                const int j = 16 * jt + ml;
                                double bv = 0.0;
                                if (k < nc && j < nc) bv = (k > j) ? TGET(k, j) : (k == j ? 1.0 : 0.0);
                                acc[jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(av[u], bv, acc[jt], 0, 0, 0);
                            }
                        }
                    }
                }
            }
#pragma unroll
            for (int jt = 0; jt < 9; ++jt) {
                if (jt < nct) {
                    const int j = 16 * jt + ml;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int rr = strip * 16 + mk + 4 * q;
                        if (rr < nb && j < nc) {
                            W[(nc + rr) + (int64_t)j * f] = acc[jt][q];
                            Wt[j + (int64_t)(nc + rr) * nc] = acc[jt][q];
                        }
                    }
                }
            }
        }
    }
#undef TGET
#undef WK
}

// count supernodes, any grid: a small grid keeps this off most CUs when it runs beside the tree's critical path
// SEL 0: every supernode of the list; 1: those of at most kWinvSmallNc columns (the 128-thread build: a front of a dozen
// columns and a hundred rows keeps a fraction of 512 threads busy, and the bulk of a tree's block-class fronts are such --
// four times the workgroups in flight on the same threads); 2: the wider ones.
template <int NT, int SEL>
__global__ __launch_bounds__(NT) void k_winv(TreeDev T, const double* __restrict__ fronts, double* __restrict__ wst,
                                             const int* __restrict__ list, int count)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    for (int it = blockIdx.x; it < count; it += gridDim.x) {
        const int s = list[it];
        if (SEL != 0) {
            const bool small = T.sn_start[s + 1] - T.sn_start[s] <= kWinvSmallNc;
            if (small != (SEL == 1)) continue;
        }
        winv_one<NT>(T, fronts, wst, s, smem);
        __syncthreads();               // LDS is reused by the next supernode
    }
}


constexpr int kSolveBS = 512;

// NR dispatch: the kernels exist for 1, 2 and 4 right-hand sides
#define HIPKKT_NR_SWITCH(nr, ...)                                                                 \
    do {                                                                                         \
        if ((nr) == 1) { constexpr int NR = 1; __VA_ARGS__; }                                    \
        else if ((nr) == 2) { constexpr int NR = 2; __VA_ARGS__; }                               \
        else { constexpr int NR = 4; __VA_ARGS__; }                                              \
    } while (0)

static void init_solve_lds()
{
    static PerDeviceOnce once;
    once.run([]() {
        hipError_t e = hipSuccess;
        auto set = [&](auto k, int bytes) { if (e == hipSuccess) e = set_max_lds(k, bytes); };
#define HIPKKT_SET_NR(NRV)                                     \
        set(k_fwd_block<128, NRV>, 160 * 1024);                \
        set(k_bwd_block<128, NRV>, 160 * 1024);                \
        set(k_fwd_block<kSolveBS, NRV>, 160 * 1024);           \
        set(k_bwd_block<kSolveBS, NRV>, 160 * 1024);           \
        set(k_fwd_level<128, NRV>, 160 * 1024);                \
        set(k_bwd_level<128, NRV>, 160 * 1024);                \
        set(k_fwd_level<kSolveBS, NRV>, 160 * 1024);           \
        set(k_bwd_level<kSolveBS, NRV>, 160 * 1024);           \
        set(k_top_solve<1024, 4, 4, NRV>, 150 * 1024);         /* (a static LDS word as well: leave room for it) */
        HIPKKT_SET_NR(1)
        HIPKKT_SET_NR(2)
        HIPKKT_SET_NR(4)
#undef HIPKKT_SET_NR
        set(k_winv<kWinvThreads, 0>, 160 * 1024);
        set(k_winv<kWinvThreads, 2>, 160 * 1024);
        set(k_top_solve<512, 7, 7, 1>, 150 * 1024);
        set(k_top_solve_sliced<1024, 1>, 150 * 1024);
        set(k_top_solve_sliced<1024, 2>, 150 * 1024);
        return e;
    });
}

// per right-hand side; the kernels lay NR columns' shares out one after the other
// ------------------------------------------------------------------ very tall fronts (f beyond ~10 000 rows)
// The block kernels keep a front's vector and partial sums in LDS, f (1 + nc / 8) doubles: a front of 14 000 rows does
// not fit whatever its width (long-range KKT graphs: cfg2 with 1 % of A's entries re-drawn over all columns has a
// 14 154-row root).  These fronts take kernels of their own, one workgroup each, nothing of size f in LDS: the top nc
// entries are gathered and solved against T = L11^{-1} in LDS, the rows below stream their row of M = L21 T from W
// (coalesced over the rows) and are stored straight to the contribution vector; backward, a wave owns columns and
// walks the rows with its lanes (the ancestors' x gathered on the way), one wave reduction per column.  Single column
// only; a launch's tall fronts sit at its end (hipkkt.hip, Launch::ntall) and stay out of the persistent kernels.
// r04: a tall front's rows are spread over workgroups -- one workgroup walked 25 088 rows x 96 columns per hop of a
// 261-panel chain (cfg2 with 0.1 % long-range couplings: 519 ms per sweep pair, 15 GB/s) -- and a direction is ONE
// launch per panel: the sweep of such a chain is a launch per panel and kernel, ~10 us each whatever the work.
// Forward (k_fwd_tall; blockIdx.y = front): every workgroup gathers the top nc entries itself (the same sums in the same
// order); workgroup 0 solves them against T, workgroup 1 + i takes block i of BR rows below.  Backward (k_bwd_tall): a
// block of BR rows each -- z for its rows in LDS, a wave per column run, one wave reduction per column and block ->
// partial sums in tall_ws -- and the workgroup that finishes a front's LAST block (a ticket word per front: tall_ws's
// first N doubles seen as ints, zero between sweeps) adds the partial sums up in block order: a fixed order,
// bit-reproducible.  The partial sums are handed over as in chain_kernels.hip: sc1 stores, s_waitcnt, barrier, one
// relaxed agent-scope atomic; the reader uses sc1 loads behind the ticket.
__global__ __launch_bounds__(1024) void k_fwd_tall(SolveArgs A, int begin, int BR)
{
    __shared__ double ytop[kBdColsSolveMax];
    const int tid = threadIdx.x;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + blockIdx.y];
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const int r0 = nc + ((int)blockIdx.x - 1) * BR, r1 = min(f, r0 + BR);
    if (blockIdx.x > 0 && r0 >= f) return;
    const double* __restrict__ W = A.tinv + fd.w_off;
    for (int i = tid; i < nc; i += 1024) {
        double v = A.b[T.perm[c0 + i]];
        const int64_t lc = (int64_t)c0 + rp + i;
        for (int64_t g = T.gl_ptr[lc]; g < T.gl_ptr[lc + 1]; ++g) v += A.uvec[T.gl_src[g]];
        ytop[i] = v;
    }
    __syncthreads();
    if (blockIdx.x == 0) {
        for (int i = tid; i < nc; i += 1024) {
            double acc = 0.0;
            for (int k = 0; k <= i; ++k) acc = fma(W[i + (int64_t)k * f], ytop[k], acc);       // T is unit lower triangular
            A.xp[c0 + i] = acc;
        }
        return;
    }
    for (int r = r0 + tid; r < r1; r += 1024) {
        const int64_t lc = (int64_t)c0 + rp + r;
        double v = 0.0;
        for (int64_t g = T.gl_ptr[lc]; g < T.gl_ptr[lc + 1]; ++g) v += A.uvec[T.gl_src[g]];
        double acc = 0.0;
        for (int k = 0; k < nc; ++k) acc = fma(W[r + (int64_t)k * f], ytop[k], acc);
        A.uvec[rp + r - nc] = v - acc;
    }
}
// partial sums of block bx of front ty (launch-local index) for column j: tall_ws[N + ((ty * nblk + bx) * kBdColsSolveMax) + j]
__global__ __launch_bounds__(1024) void k_bwd_tall(SolveArgs A, int begin, int BR, int N)
{
    extern __shared__ __attribute__((aligned(16))) double zloc[];       // BR doubles
    __shared__ int sh_last;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + blockIdx.y];
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const int r0 = (int)blockIdx.x * BR, r1 = min(f, r0 + BR);
    if (r0 >= f) return;
    const double* __restrict__ W = A.tinv + fd.w_off;
    for (int r = r0 + tid; r < r1; r += 1024)
        zloc[r - r0] = (r < nc) ? A.xp[c0 + r] * A.Dinv[c0 + r] : -A.xp[T.rows[rp + r - nc]];
    __syncthreads();
    double* front_part = A.tall_ws + N + (int64_t)blockIdx.y * gridDim.x * kBdColsSolveMax;
    double* part = front_part + (int64_t)blockIdx.x * kBdColsSolveMax;
    for (int j = wv; j < nc; j += 16) {
        double acc = 0.0;
        const double* __restrict__ Wj = W + (int64_t)j * f;
        for (int r = max(r0, j) + lane; r < r1; r += 64) acc = fma(Wj[r], zloc[r - r0], acc);
        acc = wave_reduce_sum(acc);
        if (lane == 0) ST_AGENT_F64(part + j, acc);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int nb_used = (f + BR - 1) / BR;
    int* ticket = reinterpret_cast<int*>(A.tall_ws) + fd.s;
    if (tid == 0) {
        const int t = __hip_atomic_fetch_add(ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        sh_last = (t == nb_used - 1);
        if (sh_last) __hip_atomic_store(ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);     // (zero again for the next sweep)
    }
    __syncthreads();
    if (!sh_last) return;
    for (int j = tid; j < nc; j += 1024) {
        double v = 0.0;
        for (int bx = j / BR; bx < nb_used; ++bx) v += LD_AGENT_F64(front_part + (int64_t)bx * kBdColsSolveMax + j);   // (blocks above row j hold nothing of column j)
        A.xp[c0 + j] = v;
        A.out[T.perm[c0 + j]] = v;
    }
}
static int tall_block_rows()
{
    const int br = knobs().tall_block_rows;
    return br < 64 ? 64 : (br > 4096 ? 4096 : (br & ~63));
}
size_t tall_ws_doubles(int N, int ntall, int fmax)
{
    const int BR = tall_block_rows();
    return (size_t)N + (size_t)ntall * (size_t)((fmax + BR - 1) / BR) * kBdColsSolveMax;
}
void launch_fwd_tall(const SolveArgs& a, int begin, int count, int fmax, hipStream_t st)
{
    if (count <= 0) return;
    const int BR = tall_block_rows();
    hipLaunchKernelGGL(k_fwd_tall, dim3(1 + (fmax + BR - 1) / BR, count), dim3(1024), 0, st, a, begin, BR);
}
void launch_bwd_tall(const SolveArgs& a, int begin, int count, int fmax, int N, hipStream_t st)
{
    if (count <= 0) return;
    const int BR = tall_block_rows();
    const int nblk = (fmax + BR - 1) / BR;
    hipLaunchKernelGGL(k_bwd_tall, dim3(nblk, count), dim3(1024), (size_t)BR * sizeof(double), st, a, begin, BR, N);
}

size_t solve_lds_bytes(int fmax, int ncmax)
{
    const size_t fpad = (size_t)((fmax + 3) & ~3), ncpad = (size_t)((ncmax + 3) & ~3);
    const size_t nks = (size_t)((ncmax + 7) >> 3), nrs = (size_t)((fmax + 7) >> 3);
    const size_t fwd = fpad + nks * fpad, bwd = fpad + nrs * ncpad;
    return (fwd > bwd ? fwd : bwd) * sizeof(double);
}

void launch_fwd(const SolveArgs& a, const RecSeg& rs, int begin, int count, int bs, size_t lds, hipStream_t st, int nr)
{
    if (count <= 0) return;
    init_solve_lds();
    // bs = 128: a wide level of small fronts -- more fronts in flight per CU matter more than waves per front
    if (bs == 128) HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_block<128, NR>), dim3(count), dim3(128), lds * NR, st, a, rs, begin));
    else HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_block<kSolveBS, NR>), dim3(count), dim3(kSolveBS), lds * NR, st, a, rs, begin));
}
void launch_fwd_level(const SolveArgs& a, const RecSeg& rs, int begin, int nblock, int nwave, int ntiny, int bs, size_t lds, hipStream_t st, int nr)
{
    init_solve_lds();
    if (bs == 128) {
        const int grid = nblock + (nwave + 1) / 2 + (ntiny + 15) / 16;
        HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_level<128, NR>), dim3(grid), dim3(128), lds * NR, st, a, rs, begin, nblock, nwave, ntiny));
    } else {
        const int grid = nblock + (nwave + kSolveBS / 64 - 1) / (kSolveBS / 64) + (ntiny + kSolveBS / 8 - 1) / (kSolveBS / 8);
        HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_level<kSolveBS, NR>), dim3(grid), dim3(kSolveBS), lds * NR, st, a, rs, begin, nblock, nwave, ntiny));
    }
}
void launch_bwd_level(const SolveArgs& a, const RecSeg& rs, int begin, int nblock, int nwave, int ntiny, int bs, size_t lds, hipStream_t st, int nr)
{
    init_solve_lds();
    if (bs == 128) {
        const int grid = nblock + (nwave + 1) / 2 + (ntiny + 15) / 16;
        HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_level<128, NR>), dim3(grid), dim3(128), lds * NR, st, a, rs, begin, nblock, nwave, ntiny));
    } else {
        const int grid = nblock + (nwave + kSolveBS / 64 - 1) / (kSolveBS / 64) + (ntiny + kSolveBS / 8 - 1) / (kSolveBS / 8);
        HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_level<kSolveBS, NR>), dim3(grid), dim3(kSolveBS), lds * NR, st, a, rs, begin, nblock, nwave, ntiny));
    }
}
void launch_fwd_small(const SolveArgs& a, const RecSeg& rs, int begin, int nwave, int ntiny, hipStream_t st, bool leaf, int nr)
{
    if (nwave + ntiny <= 0) return;
    const int grid = (nwave + 3) / 4 + (ntiny + 31) / 32;
    HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_small<NR>), dim3(grid), dim3(256), 0, st, a, rs, begin, nwave, ntiny, leaf ? 1 : 0));
}
void launch_bwd_small(const SolveArgs& a, const RecSeg& rs, int begin, int nwave, int ntiny, hipStream_t st, int nr)
{
    if (nwave + ntiny <= 0) return;
    const int grid = (nwave + 3) / 4 + (ntiny + 31) / 32;
    HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_small<NR>), dim3(grid), dim3(256), 0, st, a, rs, begin, nwave, ntiny));
}
void launch_bwd(const SolveArgs& a, const RecSeg& rs, int begin, int count, int bs, size_t lds, hipStream_t st, int nr)
{
    if (count <= 0) return;
    init_solve_lds();
    if (bs == 128) HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_block<128, NR>), dim3(count), dim3(128), lds * NR, st, a, rs, begin));
    else HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_block<kSolveBS, NR>), dim3(count), dim3(kSolveBS), lds * NR, st, a, rs, begin));
}
// ------------------------------------------------------------------ several right-hand sides
// The single-column kernels above are latency-bound; with many columns the cost is fetching the
// solve matrices, so these variants use every entry they load for a whole block of columns:
//   * the work vectors are ROW-major, N x KP and sum(nb) x KP (KP = column count rounded up to 16,
//     padding columns zero), so that one row of a column block is contiguous -- every gather moves
//     full cache lines; contribution rows are stored in the order of their receiver's gather list
//     (TreeDev::udst), so a receiver sums a contiguous run without the gl_src[] indirection;
//   * k_permute_in / k_permute_out convert from / to the caller's column-major vectors, coalesced
//     on the caller's side and sector-sized on the permuted side;
//   * block fronts: [x_s; w] = W y_s and x_s = W' z are GEMMs with 16 columns -> v_mfma_f64_16x16x4_f64,
//     A operand (W, 16 consecutive rows per k) straight from global, B operand from LDS (forward) or
//     straight from the row-major work vector (backward);
//   * small fronts (one wave each): lane = column, substitution with wave-uniform matrix entries.
constexpr int kMultiCB = 16;         // columns per block-kernel workgroup (one MFMA tile wide)
typedef double d4m_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_permute_in(const double* __restrict__ B, int64_t ldb, double* __restrict__ Xp,
                                                    int KP, const int* __restrict__ iperm, int N, int nrhs)
{
    // thread = (caller's row o, group of 8 columns): column reads coalesced over o, one 64-byte write per thread
    const int groups = KP >> 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)N * groups;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % N), gq = (int)(idx / N);
        const int i = iperm[o];
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) v[c] = (gq * 8 + c < nrhs) ? B[(int64_t)(gq * 8 + c) * ldb + o] : 0.0;
        double2* dst = reinterpret_cast<double2*>(Xp + (int64_t)i * KP + gq * 8);
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = make_double2(v[2 * c], v[2 * c + 1]);
    }
}
__global__ __launch_bounds__(256) void k_permute_out(double* __restrict__ X, int64_t ldx, const double* __restrict__ Xp,
                                                     int KP, const int* __restrict__ iperm, int N, int nrhs)
{
    const int groups = KP >> 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)N * groups;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int o = (int)(idx % N), gq = (int)(idx / N);
        const int i = iperm[o];
        const double2* src = reinterpret_cast<const double2*>(Xp + (int64_t)i * KP + gq * 8);
        double2 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = src[c];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if (gq * 8 + 2 * c < nrhs) X[(int64_t)(gq * 8 + 2 * c) * ldx + o] = v[c].x;
            if (gq * 8 + 2 * c + 1 < nrhs) X[(int64_t)(gq * 8 + 2 * c + 1) * ldx + o] = v[c].y;
        }
    }
}

// One-wave fronts (f <= 64): lane = right-hand-side column (64 per wave, grid.y = KP / 64), so every row
// of the work vectors is one coalesced 512-byte access and all lanes work whatever the front's size; the
// entries of L, the gather lists and the row indices are wave-uniform (scalar loads).  The first kWC own
// unknowns stay in registers; beyond that they are re-read from the work vector (same lane wrote them).
constexpr int kWC = 16;

// xp (N x KP) holds the permuted right-hand sides on entry of the forward sweep
// V columns per lane (2: 16-byte accesses, a wave covers 128 columns; the row loop is a chain of dependent loads, so twice
// the bytes per step is nearly twice the rate)
template <int V>
__global__ __launch_bounds__(256) void k_fwd_wave_m(SolveArgs A, int begin, int count, int KP)
{
    typedef double vec_t __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= count) return;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + item];
    if (fd.pad & 1) return;                  // a pulled leaf: its parent computes its contributions (TreeDev::hp_*)
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + fd.front_off;
    const int col = (blockIdx.y * 64 + lane) * V;
    const bool act = col < KP;
    const int cc = act ? col : KP - V;
    double* __restrict__ xp = A.xp + cc;
    double* __restrict__ uvec = A.uvec + cc;
    const double* __restrict__ bsrc = A.b ? A.b + cc : nullptr;
    auto ld = [](const double* p) -> vec_t { return *reinterpret_cast<const vec_t*>(p); };
    auto st = [](double* p, vec_t v) { *reinterpret_cast<vec_t*>(p) = v; };

    vec_t yk[kWC];
#pragma unroll
    for (int k = 0; k < kWC; ++k) yk[k] = 0.0;
    for (int i = 0; i < f; ++i) {
        // (A.b: the right-hand sides in the caller's row order, row-major like xp -- read through the permutation here
        //  instead of in a pass of its own)
        vec_t v = 0.0;
        if (i < nc) v = bsrc ? ld(bsrc + (int64_t)T.perm[c0 + i] * KP) : ld(xp + (int64_t)(c0 + i) * KP);
        const int64_t lc = (int64_t)c0 + rp + i;
        const int64_t g0 = T.glm_ptr[lc], g1 = T.glm_ptr[lc + 1];
        for (int64_t g = g0; g < g1; g += 4) {
            vec_t u[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) u[q] = (g + q < g1) ? ld(uvec + (g + q) * KP) : (vec_t)0.0;
#pragma unroll
            for (int q = 0; q < 4; ++q) v += u[q];
        }
        const int km = min(i, nc);
        const double* __restrict__ Li = F + i;                 // L(i, k) = Li[k * f]
#pragma unroll
        for (int k = 0; k < kWC; ++k)
            if (k < km) {
                const double l = -Li[(int64_t)k * f];
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = fma(l, yk[k][e], v[e]);
            }
        for (int k = kWC; k < km; ++k) {
            const double l = -Li[(int64_t)k * f];
            const vec_t y = ld(xp + (int64_t)(c0 + k) * KP);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = fma(l, y[e], v[e]);
        }
        if (i < nc) {
#pragma unroll
            for (int k = 0; k < kWC; ++k)
                if (k == i) yk[k] = v;
            if (act) st(xp + (int64_t)(c0 + i) * KP, v);
        } else if (act) {
            st(uvec + (int64_t)T.udst_m[rp + i - nc] * KP, v);
        }
    }
}

// own columns in chunks of kWC from the last one up: acc_j = D^{-1} y_j - sum over the rows below the chunk,
// each such row loaded once; then the triangle inside the chunk from registers
// Forward: the pulled leaves' terms, summed per RECEIVING row before the tree is walked: row k of the list (TreeDev::pr_*)
// gets sum_h -L(r,0) b_c over its pulled leaves and stores it as the first entry of its run in the receiver-ordered
// contribution store -- the parents read it like any other stored contribution.  A gather like the residual's: 16 lanes
// per (row, block of 16 columns), eight leaf rows of B in flight (a leaf's row serves all the parent rows it reaches: L2).
template <int V>                 // columns per lane (1, or 2: double2 accesses, KP a multiple of 32)
__global__ __launch_bounds__(256) void k_pull_leaves_m(SolveArgs A, int nrows, int KP)
{
    typedef double vec_t __attribute__((ext_vector_type(V)));
    const int c = threadIdx.x & 15;
    const int k = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (k >= nrows) return;
    const TreeDev& T = A.T;
    const int64_t h0 = T.pr_ptr[k], h1 = T.pr_ptr[k + 1];
    const int64_t col = ((int64_t)blockIdx.y * 16 + c) * V;
    const int* __restrict__ hrow = A.b ? T.hp_row : T.hp_col;       // the leaf's row of B: caller's order / tree order
    const double* __restrict__ src = A.b ? A.b : A.xp;
    vec_t acc = 0.0;
    for (int64_t h = h0; h < h1; h += 4) {
        vec_t u[4];
        double l[4];
        int cj[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool ok = h + e < h1;
            cj[e] = ok ? hrow[h + e] : -1;
            l[e] = ok ? A.fronts[T.hp_lidx[h + e]] : 0.0;
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) u[e] = cj[e] >= 0 ? *reinterpret_cast<const vec_t*>(src + (int64_t)cj[e] * KP + col) : (vec_t)0.0;
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (cj[e] >= 0) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fma(-l[e], u[e][v], acc[v]);
            }
    }
    *reinterpret_cast<vec_t*>(A.uvec + (int64_t)T.pr_slot[k] * KP + col) = acc;
}
void launch_pull_leaves_multi(const SolveArgs& a, int nrows, int KP, hipStream_t st)
{
    if (nrows <= 0) return;
    const int vmax = knobs().multi_vec;
    if (KP % 64 == 0 && vmax >= 4) hipLaunchKernelGGL(k_pull_leaves_m<4>, dim3((nrows + 15) / 16, KP / 64), dim3(256), 0, st, a, nrows, KP);
    else if (KP % 32 == 0) hipLaunchKernelGGL(k_pull_leaves_m<2>, dim3((nrows + 15) / 16, KP / 32), dim3(256), 0, st, a, nrows, KP);
    else hipLaunchKernelGGL(k_pull_leaves_m<1>, dim3((nrows + 15) / 16, KP / 16), dim3(256), 0, st, a, nrows, KP);
}

// Backward step of the PULLED leaves of a launch (one column, a handful of rows: TreeDev::hp_*): x_c = y_c / d_c - sum_r
// L(r,0) x_r with y_c = b_c -- a gather like the residual's: 16 lanes per (leaf, block of 16 columns), the leaf's rows of
// the ancestors' x all in flight (they are shared by the leaves of a parent: L2), same order of operations as k_bwd_wave_m.
template <int V>
__global__ __launch_bounds__(256) void k_bwd_leaf_m(SolveArgs A, int begin, int count, int KP)
{
    typedef double vec_t __attribute__((ext_vector_type(V)));
    const int c = threadIdx.x & 15;
    const int item = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (item >= count) return;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + item];
    if (!(fd.pad & 1)) return;
    const int c0 = fd.c0, nb = fd.nb;
    const int64_t rp = fd.rp;
    const double* __restrict__ F = A.fronts + fd.front_off;          // column 0: 1, L(1,0), L(2,0), ...
    const int64_t col = ((int64_t)blockIdx.y * 16 + c) * V;
    const int64_t orow = (int64_t)T.perm[c0] * KP + col;
    const vec_t y = A.b ? *reinterpret_cast<const vec_t*>(A.b + orow) : *reinterpret_cast<const vec_t*>(A.xp + (int64_t)c0 * KP + col);
    const double dinv = A.Dinv[c0];
    vec_t acc;
#pragma unroll
    for (int v = 0; v < V; ++v) acc[v] = y[v] * dinv;
    for (int r0 = 0; r0 < nb; r0 += 8) {
        vec_t xr[8];
        double l[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const bool ok = r0 + e < nb;
            l[e] = ok ? F[1 + r0 + e] : 0.0;
            xr[e] = ok ? *reinterpret_cast<const vec_t*>(A.xp + (int64_t)T.rows[rp + r0 + e] * KP + col) : (vec_t)0.0;
        }
#pragma unroll
        for (int e = 0; e < 8; ++e)
            if (r0 + e < nb) {
#pragma unroll
                for (int v = 0; v < V; ++v) acc[v] = fma(-l[e], xr[e][v], acc[v]);
            }
    }
    // (nobody reads a leaf's x from the tree-ordered copy: it is stored there only where that copy IS the result)
    if (A.out) {
        vec_t o = acc;
        if (A.add) { const vec_t a = *reinterpret_cast<const vec_t*>(A.add + orow); o += a; }
        *reinterpret_cast<vec_t*>(A.out + orow) = o;
    } else {
        *reinterpret_cast<vec_t*>(A.xp + (int64_t)c0 * KP + col) = acc;
    }
}

template <int V>
__global__ __launch_bounds__(256) void k_bwd_wave_m(SolveArgs A, int begin, int count, int KP)
{
    typedef double vec_t __attribute__((ext_vector_type(V)));
    const int lane = threadIdx.x & 63;
    const int item = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6));
    if (item >= count) return;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + item];
    if (fd.pad & 1) return;                  // a pulled leaf: k_bwd_leaf_m
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + fd.front_off;
    const int col = (blockIdx.y * 64 + lane) * V;
    const bool act = col < KP;
    const int cc = act ? col : KP - V;
    double* __restrict__ xp = A.xp + cc;
    auto ld = [](const double* p) -> vec_t { return *reinterpret_cast<const vec_t*>(p); };
    auto st = [](double* p, vec_t v) { *reinterpret_cast<vec_t*>(p) = v; };

    for (int jhi = nc; jhi > 0; jhi -= kWC) {
        const int jlo = max(0, jhi - kWC);
        const int w = jhi - jlo;
        vec_t acc[kWC];
#pragma unroll
        for (int q = 0; q < kWC; ++q) {
            acc[q] = 0.0;
            if (q < w) {
                const vec_t y = ld(xp + (int64_t)(c0 + jlo + q) * KP);
                const double d = A.Dinv[c0 + jlo + q];
#pragma unroll
                for (int e = 0; e < V; ++e) acc[q][e] = y[e] * d;
            }
        }
        for (int r0 = jhi; r0 < f; r0 += 4) {
            vec_t xr[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                xr[e] = 0.0;
                if (r < f) xr[e] = ld(xp + (int64_t)(r < nc ? c0 + r : T.rows[rp + r - nc]) * KP);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int r = r0 + e;
                if (r < f) {
                    const double* __restrict__ Lr = F + r + (int64_t)jlo * f;      // L(r, jlo + q) = Lr[q * f]
#pragma unroll
                    for (int q = 0; q < kWC; ++q)
                        if (q < w) {
                            const double l = -Lr[(int64_t)q * f];
#pragma unroll
                            for (int z = 0; z < V; ++z) acc[q][z] = fma(l, xr[e][z], acc[q][z]);
                        }
                }
            }
        }
#pragma unroll
        for (int q = kWC - 1; q >= 0; --q) {
            if (q < w) {
                const vec_t xj = acc[q];
                if (act) {
                    st(xp + (int64_t)(c0 + jlo + q) * KP, xj);
                    if (A.out) {         // the solution in the caller's row order (+ add), stored here instead of in a pass of its own
                        const int64_t o = (int64_t)T.perm[c0 + jlo + q] * KP + cc;
                        vec_t ov = xj;
                        if (A.add) ov += ld(A.add + o);
                        st(A.out + o, ov);
                    }
                }
                const double* __restrict__ Lj = F + (jlo + q) + (int64_t)jlo * f;   // L(jlo + q, jlo + q2) = Lj[q2 * f]
#pragma unroll
                for (int q2 = 0; q2 < kWC; ++q2)
                    if (q2 < q) {
                        const double l = -Lj[(int64_t)q2 * f];
#pragma unroll
                        for (int z = 0; z < V; ++z) acc[q2][z] = fma(l, xj[z], acc[q2][z]);
                    }
            }
        }
    }
}

// forward, block fronts: Ys = gathered own rows (nc x 16, LDS); every wave takes 16-row tiles of W,
// out = W Ys on the matrix cores; rows below the diagonal block gather their children's contributions
// in the epilogue (they are only needed there).
// CT column tiles of 16 per workgroup (2 where 32 | KP: every W entry, gather-list bound, row index and store position is
// fetched once for 32 columns instead of 16, the gathered rows are 256 contiguous bytes)
template <int BS, int CT>
__global__ __launch_bounds__(BS) void k_fwd_block_m(SolveArgs A, int begin, int count, int KP)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 15, mk = lane >> 4;
    const TreeDev& T = A.T;
    // Grid (8, ceil(count / 8) * column blocks): workgroup (x, y) takes front 8 (y / ncb) + x and column block y % ncb.  A
    // launch's workgroups go to the XCDs round-robin by linear index, i.e. by x here: all column blocks of a front run on
    // ONE XCD and share its L2, so the front's W comes from HBM once instead of once per XCD (or per column block).
    constexpr int CB = 16 * CT;
    const int ncb = KP / CB;
    const int fi = 8 * ((int)blockIdx.y / ncb) + (int)blockIdx.x, cbk = (int)blockIdx.y % ncb;
    if (fi >= count) return;
    const FrontDesc fd = T.desc[begin + fi];
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const double* __restrict__ W = A.tinv + fd.w_off;                  // f x nc, ld f
    double* __restrict__ xp = A.xp + cbk * CB;
    double* __restrict__ uvec = A.uvec + cbk * CB;
    const int ncp = (nc + 3) & ~3;
    double* Ys = smem;                       // ncp x CB row-major

    // own rows: thread = (row, column); four rows per thread in flight so that the dependent chain
    // gather-list bounds -> sources -> values is paid once per four rows
    constexpr int NW = BS / 64;
    for (int base = 0; base < ncp * CB; base += 4 * BS) {
        int pg0[4], pg1[4];                  // (gather-list bounds fit 32 bits: the row structure is int32-indexed)
        double pv[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = base + p * BS + tid;
            const int i = idx / CB;
            pg0[p] = pg1[p] = 0;
            pv[p] = 0.0;
            if (i < nc) {
                const int64_t lc = (int64_t)c0 + rp + i;
                pg0[p] = (int)T.glm_ptr[lc];
                pg1[p] = (int)T.glm_ptr[lc + 1];
                pv[p] = A.b ? A.b[(int64_t)T.perm[c0 + i] * KP + (xp - A.xp) + (idx & (CB - 1))] : xp[(int64_t)(c0 + i) * KP + (idx & (CB - 1))];
            }
        }
        // (measured r03: four sources per row and round instead of two -- sixteen loads per thread in flight -- was no
        //  faster: 64 columns 5.23 vs 4.99 ms per solve, 512 columns 28.5 vs 27.4)
        for (int e = 0;; e += 2) {
            bool any = false;
#pragma unroll
            for (int p = 0; p < 4; ++p) any = any || pg0[p] + e < pg1[p];
            if (!any) break;
            double u[4][2];
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int d = 0; d < 2; ++d)
                    u[p][d] = (pg0[p] + e + d < pg1[p]) ? uvec[(int64_t)(pg0[p] + e + d) * KP + (tid & (CB - 1))] : 0.0;
#pragma unroll
            for (int p = 0; p < 4; ++p) pv[p] = (pv[p] + u[p][0]) + u[p][1];
        }
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int idx = base + p * BS + tid;
            if (idx < ncp * CB) Ys[idx] = pv[p];
        }
    }
    __syncthreads();
    const int ntile = (f + 15) >> 4;
    for (int t = wv; t < ntile; t += NW) {
        const int r0 = t * 16;
        // gather lists of this lane's four output rows (r0 + mk + 4q), fetched ahead of the product
        int g0[4], g1[4];
        int ud[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = r0 + mk + 4 * q;
            g0[q] = g1[q] = 0;
            ud[q] = 0;
            if (i >= nc && i < f) {
                const int64_t lc = (int64_t)c0 + rp + i;
                g0[q] = (int)T.glm_ptr[lc];
                g1[q] = (int)T.glm_ptr[lc + 1];
                ud[q] = T.udst_m[rp + i - nc];
            }
        }
        d4m_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (d4m_t){0.0, 0.0, 0.0, 0.0};
        const int kend = min(nc, r0 + 16);               // T = L11^{-1} is lower triangular
        const bool rowok = r0 + ml < f;
        const double* __restrict__ Wr = W + r0 + ml;
        for (int k0 = 0; k0 < kend; k0 += 32) {
            double a[8];
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                const int k = k0 + 4 * s + mk;
                a[s] = (rowok && k < kend) ? Wr[(int64_t)k * f] : 0.0;
            }
#pragma unroll
            for (int s = 0; s < 8; ++s) {
                if (k0 + 4 * s < kend) {
                    const int k = k0 + 4 * s + mk;       // < ncp
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[s], Ys[k * CB + 16 * ct + ml], acc[ct], 0, 0, 0);
                }
            }
        }
        // epilogue: the four rows' gather lists advance together (one dependent chain for all of them)
        double gv[CT][4];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) gv[ct][q] = 0.0;
        for (int e = 0;; e += 2) {
            bool any = false;
#pragma unroll
            for (int q = 0; q < 4; ++q) any = any || g0[q] + e < g1[q];
            if (!any) break;
            double u[CT][4][2];
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int d = 0; d < 2; ++d)
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct)
                        u[ct][q][d] = (g0[q] + e + d < g1[q]) ? uvec[(int64_t)(g0[q] + e + d) * KP + 16 * ct + ml] : 0.0;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct)
#pragma unroll
                for (int q = 0; q < 4; ++q) gv[ct][q] = (gv[ct][q] + u[ct][q][0]) + u[ct][q][1];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int i = r0 + mk + 4 * q;
#pragma unroll
            for (int ct = 0; ct < CT; ++ct) {
                if (i < nc) xp[(int64_t)(c0 + i) * KP + 16 * ct + ml] = acc[ct][q];
                else if (i < f) uvec[(int64_t)ud[q] * KP + 16 * ct + ml] = gv[ct][q] - acc[ct][q];
            }
        }
    }
}

// backward, block fronts: x_s = W' z, z = [D^{-1} y_s ; -x_below].  Items = (16-column tile of W', slice of
// the rows); the B operand z is read straight from the row-major work vector (16 lanes = 128 contiguous
// bytes per row); partial tiles are combined through LDS in slice order.
__device__ inline int bwd_multi_slices(int nt, int f, int nwaves)
{
    int ns = nwaves / nt;
    const int cap = (f + 31) >> 5;
    if (ns > cap) ns = cap;
    return ns < 1 ? 1 : ns;
}
template <int BS, int CT>
__global__ __launch_bounds__(BS) void k_bwd_block_m(SolveArgs A, int begin, int count, int KP)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ml = lane & 15, mk = lane >> 4;
    const TreeDev& T = A.T;
    constexpr int CB = 16 * CT;
    const int ncb = KP / CB;                    // (grid as in k_fwd_block_m: a front's column blocks on one XCD)
    const int fi = 8 * ((int)blockIdx.y / ncb) + (int)blockIdx.x, cbk = (int)blockIdx.y % ncb;
    if (fi >= count) return;
    const FrontDesc fd = T.desc[begin + fi];
    const int c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int64_t rp = fd.rp;
    const int f = nc + nb;
    const double* __restrict__ Wt = A.tinv + fd.w_off + (int64_t)f * nc;       // W'(j, r) at j + r*nc
    double* __restrict__ xp = A.xp + cbk * CB;
    const int nt = (nc + 15) >> 4;
    constexpr int NW = BS / 64;
    const int ns = bwd_multi_slices(nt, f, NW);
    const int sl = ((((f + ns - 1) / ns) + 3) >> 2) << 2;
    double* part = smem;                     // (ns * nt) x CT tiles of 16 x 16

    for (int it = wv; it < nt * ns; it += NW) {
        const int s = it / nt, jt = it - s * nt;
        const int j0 = 16 * jt;
        const int rbeg = max(s * sl, j0 & ~3);           // W'(j, r) = 0 for r < j
        const int rend = min(f, (s + 1) * sl);
        const bool colok = j0 + ml < nc;
        const double* __restrict__ Wc = Wt + j0 + ml;
        d4m_t acc[CT];
#pragma unroll
        for (int ct = 0; ct < CT; ++ct) acc[ct] = (d4m_t){0.0, 0.0, 0.0, 0.0};
        for (int r0 = rbeg; r0 < rend; r0 += 32) {
            int ri[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + 4 * q + mk;
                ri[q] = -1;
                if (r < rend) ri[q] = (r < nc) ? c0 + r : T.rows[rp + r - nc];
            }
            double a[8], b[CT][8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + 4 * q + mk;
                a[q] = (colok && r < rend) ? Wc[(int64_t)r * nc] : 0.0;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[ct][q] = ri[q] >= 0 ? xp[(int64_t)ri[q] * KP + 16 * ct + ml] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int r = r0 + 4 * q + mk;
                const double sc = (r < nc) ? A.Dinv[c0 + r] : -1.0;
#pragma unroll
                for (int ct = 0; ct < CT; ++ct) b[ct][q] = (r < nc) ? b[ct][q] * sc : -b[ct][q];
            }
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (r0 + 4 * q < rend) {
#pragma unroll
                    for (int ct = 0; ct < CT; ++ct) acc[ct] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[q], b[ct][q], acc[ct], 0, 0, 0);
                }
        }
#pragma unroll
        for (int ct = 0; ct < CT; ++ct)
#pragma unroll
            for (int q = 0; q < 4; ++q) part[(it * CT + ct) * 256 + (mk + 4 * q) * 16 + ml] = acc[ct][q];
    }
    __syncthreads();
    for (int idx = tid; idx < nc * CB; idx += BS) {
        const int j = idx / CB, n = idx & (CB - 1);
        const int jt = j >> 4;
        double v = 0.0;
        for (int s = 0; s < ns; ++s) v += part[((s * nt + jt) * CT + (n >> 4)) * 256 + (j & 15) * 16 + (n & 15)];
        xp[(int64_t)(c0 + j) * KP + n] = v;
        if (A.out) {
            const int64_t o = (int64_t)T.perm[c0 + j] * KP + (xp - A.xp) + n;
            A.out[o] = A.add ? v + A.add[o] : v;
        }
    }
}

// add (nullable, dir = 1 only): dst[o] = src[iperm[o]] + add[o] -- the refinement step's x + dx formed on the way out
__global__ __launch_bounds__(256) void k_permute_rows(double* __restrict__ dst, const double* __restrict__ src, int KP,
                                                      const int* __restrict__ iperm, int N, int dir, const double* __restrict__ add)
{
    const int groups = KP >> 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)N * groups;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int g = (int)(idx % groups), o = (int)(idx / groups);          // a row's groups are adjacent threads
        const int i = iperm[o];
        const double2* s2 = reinterpret_cast<const double2*>(src + (int64_t)(dir ? i : o) * KP + 8 * g);
        double2* d2 = reinterpret_cast<double2*>(dst + (int64_t)(dir ? o : i) * KP + 8 * g);
        double2 v[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = s2[c];
        if (add) {
            const double2* a2 = reinterpret_cast<const double2*>(add + (int64_t)o * KP + 8 * g);
#pragma unroll
            for (int c = 0; c < 4; ++c) { const double2 t = a2[c]; v[c].x += t.x; v[c].y += t.y; }
        }
#pragma unroll
        for (int c = 0; c < 4; ++c) d2[c] = v[c];
    }
}
void launch_permute_rows(double* dst, const double* src, int KP, const int* iperm, int N, int dir, hipStream_t st, const double* add)
{
    const int64_t work = (int64_t)N * (KP >> 3);
    int g = (int)std::min<int64_t>((work + 255) / 256, 8192);
    hipLaunchKernelGGL(k_permute_rows, dim3(g < 1 ? 1 : g), dim3(256), 0, st, dst, src, KP, iperm, N, dir, dir ? add : nullptr);
}
void launch_permute_in(const double* B, int64_t ldb, double* Xp, int KP, const int* iperm, int N, int nrhs, hipStream_t st)
{
    const int64_t work = (int64_t)N * (KP >> 3);
    int g = (int)std::min<int64_t>((work + 255) / 256, 8192);
    hipLaunchKernelGGL(k_permute_in, dim3(g < 1 ? 1 : g), dim3(256), 0, st, B, ldb, Xp, KP, iperm, N, nrhs);
}
void launch_permute_out(double* X, int64_t ldx, const double* Xp, int KP, const int* iperm, int N, int nrhs, hipStream_t st)
{
    const int64_t work = (int64_t)N * (KP >> 3);
    int g = (int)std::min<int64_t>((work + 255) / 256, 8192);
    hipLaunchKernelGGL(k_permute_out, dim3(g < 1 ? 1 : g), dim3(256), 0, st, X, ldx, Xp, KP, iperm, N, nrhs);
}
void launch_fwd_multi(const SolveArgs& a, int begin, int count, bool small, int ncmax, int KP, hipStream_t st)
{
    if (count <= 0) return;
    if (small) {
        if (KP % 128 == 0) hipLaunchKernelGGL(k_fwd_wave_m<2>, dim3((count + 3) / 4, KP / 128), dim3(256), 0, st, a, begin, count, KP);
        else hipLaunchKernelGGL(k_fwd_wave_m<1>, dim3((count + 3) / 4, (KP + 63) / 64), dim3(256), 0, st, a, begin, count, KP);
        return;
    }
    // many column blocks: smaller workgroups, more fronts in flight (measured: 256 columns 21.4 vs 23.4 ms)
    const bool wide = knobs().multi_ct != 1;
    if (wide && KP % 32 == 0 && KP >= 256) {         // (measured: 128 columns 6.2 vs 6.4 ms, 512 columns 19.8 vs 19.1)
        const size_t lds = (size_t)((ncmax + 3) & ~3) * 32 * sizeof(double);
        const dim3 grid(8, ((count + 7) / 8) * (KP / 32));
        hipLaunchKernelGGL((k_fwd_block_m<256, 2>), grid, dim3(256), lds, st, a, begin, count, KP);
        return;
    }
    const size_t lds = (size_t)((ncmax + 3) & ~3) * 16 * sizeof(double);
    const dim3 grid(8, ((count + 7) / 8) * (KP / kMultiCB));
    if (KP >= 128) hipLaunchKernelGGL((k_fwd_block_m<256, 1>), grid, dim3(256), lds, st, a, begin, count, KP);
    else hipLaunchKernelGGL((k_fwd_block_m<512, 1>), grid, dim3(512), lds, st, a, begin, count, KP);
}
void launch_bwd_multi(const SolveArgs& a, int begin, int count, bool small, int ncmax, int KP, hipStream_t st, bool leaves)
{
    if (count <= 0) return;
    if (small) {
        if (KP % 128 == 0) hipLaunchKernelGGL(k_bwd_wave_m<2>, dim3((count + 3) / 4, KP / 128), dim3(256), 0, st, a, begin, count, KP);
        else hipLaunchKernelGGL(k_bwd_wave_m<1>, dim3((count + 3) / 4, (KP + 63) / 64), dim3(256), 0, st, a, begin, count, KP);
        // the launch's pulled leaves (tree level 0 only: they have no children)
        if (leaves) {
            const int vmax = knobs().multi_vec;
            if (KP % 64 == 0 && vmax >= 4) hipLaunchKernelGGL(k_bwd_leaf_m<4>, dim3((count + 15) / 16, KP / 64), dim3(256), 0, st, a, begin, count, KP);
            else if (KP % 32 == 0) hipLaunchKernelGGL(k_bwd_leaf_m<2>, dim3((count + 15) / 16, KP / 32), dim3(256), 0, st, a, begin, count, KP);
            else hipLaunchKernelGGL(k_bwd_leaf_m<1>, dim3((count + 15) / 16, KP / 16), dim3(256), 0, st, a, begin, count, KP);
        }
        return;
    }
    // at most max(8, nt) partial tiles of 16 x 16
    const int nt = (ncmax + 15) >> 4;
    const bool wide = knobs().multi_ct != 1;
    if (wide && KP % 32 == 0 && KP >= 256) {         // (measured: 128 columns 6.2 vs 6.4 ms, 512 columns 19.8 vs 19.1)
        const size_t lds = (size_t)std::max(8, nt) * 256 * 2 * sizeof(double);
        const dim3 grid(8, ((count + 7) / 8) * (KP / 32));
        hipLaunchKernelGGL((k_bwd_block_m<256, 2>), grid, dim3(256), lds, st, a, begin, count, KP);
        return;
    }
    const size_t lds = (size_t)std::max(8, nt) * 256 * sizeof(double);
    const dim3 grid(8, ((count + 7) / 8) * (KP / kMultiCB));
    if (KP >= 128) hipLaunchKernelGGL((k_bwd_block_m<256, 1>), grid, dim3(256), lds, st, a, begin, count, KP);
    else hipLaunchKernelGGL((k_bwd_block_m<512, 1>), grid, dim3(512), lds, st, a, begin, count, KP);
}

// resident workgroups the device guarantees for the persistent kernel with `lds` bytes of dynamic LDS (the 1024-thread
// build's NR = 1 instance stands for all NR: same launch bounds, the caller passes NR times the LDS)
int top_solve_capacity(size_t lds, bool tall)
{
    init_solve_lds();
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const void* fn = tall ? reinterpret_cast<const void*>(k_top_solve<1024, 4, 4, 1>) : reinterpret_cast<const void*>(k_top_solve<512, 7, 7, 1>);
    const hipError_t e = tall ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve<1024, 4, 4, 1>, 1024, lds)
                              : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve<512, 7, 7, 1>, 512, lds);
    if (e != hipSuccess) return 0;
    // Either build has ONE workgroup per CU resident: the 1024-thread one fills a CU's wave slots at 128 VGPRs, the
    // 512-thread one is built for 2 waves per SIMD (~177 VGPRs).  (Measured: a 128-VGPR 512-thread build with two
    // workgroups per CU and fewer parked items is 7 % slower than that, with as many it spills: 40 % slower.)  MI355X_MICROARCH.md (residency) caps 256-thread blocks at
    // min(API, 8, floor(800 / (ceil(sgpr/16)*16 + 16))) per CU, i.e. 3 x 512 threads at ~106 SGPRs: the
    // register-limited API answer binds.  Keep 6 % spare.
    if (knobs().verbose) {
        hipFuncAttributes fa;
        if (hipFuncGetAttributes(&fa, fn) == hipSuccess)
            std::fprintf(stderr, "[hipkkt] k_top_solve<%d>: %d registers, %zu B dynamic LDS -> %d workgroup(s) per CU\n", tall ? 1024 : 512,
                         fa.numRegs, lds, per_cu);
    }
    per_cu = per_cu > 3 ? 3 : per_cu;
    return (int)(per_cu * prop.multiProcessorCount * 0.94);
}
// the NR-column instances may need more registers than the one the capacity was asked for: ask each of them once
int top_solve_capacity_nr(size_t lds_total, int nr)
{
    init_solve_lds();
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    hipError_t e;
    if (nr == 1) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve<1024, 4, 4, 1>, 1024, lds_total);
    else if (nr == 2) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve<1024, 4, 4, 2>, 1024, lds_total);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve<1024, 4, 4, 4>, 1024, lds_total);
    if (e != hipSuccess) return 0;
    per_cu = per_cu > 3 ? 3 : per_cu;
    return (int)(per_cu * prop.multiProcessorCount * 0.94);
}
void launch_top_solve(const SolveArgs& a, int begin, int count, int grid, size_t lds, int* flags, int nflag, int epoch,
                      hipStream_t st, bool tall, int nr)
{
    if (count <= 0 || grid <= 0 || nflag < count) return;
    init_solve_lds();
    if (tall || nr > 1)
        HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_top_solve<1024, 4, 4, NR>), dim3(std::min(grid, count)), dim3(1024), lds * NR, st, a,
                                                 begin, flags, epoch, count, nflag));
    else
        hipLaunchKernelGGL((k_top_solve<512, 7, 7, 1>), dim3(std::min(grid, count)), dim3(512), lds, st, a, begin, flags, epoch, count, nflag);
}
int top_solve_sliced_capacity(size_t lds, int nr)
{
    init_solve_lds();
    int per_cu = 0, dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return 0;
    const hipError_t e = nr == 2 ? hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve_sliced<1024, 2>, 1024, lds)
                                 : hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_top_solve_sliced<1024, 1>, 1024, lds);
    if (e != hipSuccess) return 0;
    per_cu = per_cu > 1 ? 1 : per_cu;
    return (int)(per_cu * prop.multiProcessorCount * 0.94);
}
void launch_top_solve_sliced(const SolveArgs& a, int begin, int pos0, int task0, int task1, int grid, size_t lds, int* flags, int nflag,
                             int epoch, hipStream_t st, int nr)
{
    const int ntask = task1 - task0;
    if (ntask <= 0 || grid <= 0 || nflag < task1) return;
    init_solve_lds();
    // (lds: one right-hand side's share)
    if (nr == 2)
        hipLaunchKernelGGL((k_top_solve_sliced<1024, 2>), dim3(std::min(grid, ntask)), dim3(1024), lds * 2, st, a, begin, pos0, task0,
                           task1, flags, epoch, nflag);
    else
        hipLaunchKernelGGL((k_top_solve_sliced<1024, 1>), dim3(std::min(grid, ntask)), dim3(1024), lds, st, a, begin, pos0, task0, task1,
                           flags, epoch, nflag);
}
int winv_small_nc() { return kWinvSmallNc; }
void launch_tinv(const TreeDev& T, const double* fronts, double* tinv, const int* list, int count, int ncmax,
                 hipStream_t st, int max_blocks, int nsmall)
{
    if (count <= 0) return;
    init_solve_lds();
    const size_t lds = (size_t)ncmax * (ncmax | 1) * sizeof(double);
    const int grid = (max_blocks > 0 && max_blocks < count) ? max_blocks : count;
    // (the narrow supernodes on their own grid only where they fill it once at least: a structure of wide fronts -- cfg5 --
    //  would pay the second launch and its four-times-wider grid beside the panels for nothing: factorisation 4.24 -> 4.31 ms)
    if (max_blocks > 0 && max_blocks < count && knobs().winv_split && nsmall >= max_blocks * (kWinvThreads / kWinvSmallThreads)) {
        // a bounded grid beside the tree's critical path: the wide supernodes first (the long ones), then the narrow ones on
        // four times as many, four times smaller workgroups
        hipLaunchKernelGGL((k_winv<kWinvThreads, 2>), dim3(grid), dim3(kWinvThreads), lds, st, T, fronts, tinv, list, count);
        const int nc_s = ncmax < kWinvSmallNc ? ncmax : kWinvSmallNc;
        const size_t lds_s = (size_t)nc_s * (nc_s | 1) * sizeof(double);
        const int grid_s = std::min(count, max_blocks * (kWinvThreads / kWinvSmallThreads));
        hipLaunchKernelGGL((k_winv<kWinvSmallThreads, 1>), dim3(grid_s), dim3(kWinvSmallThreads), lds_s, st, T, fronts, tinv, list, count);
        return;
    }
    hipLaunchKernelGGL((k_winv<kWinvThreads, 0>), dim3(grid), dim3(kWinvThreads), lds, st, T, fronts, tinv, list, count);
}

}  // namespace hipkkt
