// Chained levels of the tree sweeps (gfx950, wave64): several levels of the assembly tree in ONE launch.
//
// The per-level path (solve_kernels.hip) pays a kernel boundary per level and sweep: ~11-18 us each on cfg2's wide
// bottom levels, of which a few us are work -- a level's launch is one front's chain of dependent loads plus launch ramp
// and drain, and every front of level l+1 waits for the SLOWEST front of level l.  Here the levels of a sweep are
// segments of one grid; a front waits only for ITS OWN children (forward) or its parent (backward) through words in
// memory, so the levels overlap and a level costs one hand-over (~3-4 us) instead of a launch.
//
// Replaces the same part of QDLDL.solve! as solve_kernels.hip
// (/root/reference/src/kktsolvers/direct-ldl/directldl_qdldl.jl:85-96); the arithmetic of a front -- gather order,
// products, partial sums -- is that of the per-level kernels, so a chained sweep is bit-identical to a per-level one.
//
// FORWARD PROGRESS.  Workgroups are numbered segment by segment in processing order (forward: leaves first; backward:
// the top level first), so a workgroup waits only for workgroups with a LOWER index.  The hardware deals the workgroups
// of a grid to the eight XCDs round-robin (workgroup i to XCD i mod 8) and every XCD starts its share in index order.
// Take the lowest-index workgroup w that has not finished: everything it waits for has finished; every workgroup before
// it on ITS XCD has finished, so w has been started or is the next one its XCD starts, and the slots it needs there are
// held only by OTHER kernels -- never by later workgroups of this grid on that XCD, which are started after w.  If
// every other kernel on the device ends by itself, w gets its slot and runs to its end, and by induction the grid drains,
// whatever its size: no residency requirement.  The premise matters: beside ANOTHER handle's waiting kernel (a
// persistent sweep kernel, an overlapped factorisation, another chained grid) the argument fails -- that kernel's
// resident workgroups may hold w's XCD while they wait for workgroups of their own that need the CUs this grid's resident
// workgroups hold (seen in round 4 as 50 ms give-ups).  Hence a chained sweep runs only under the device's token
// (hipkkt.hip, DevToken: one waiting kernel in flight per device, whichever handle's); a handle without it sweeps level
// by level.  The in-order start is hardware behaviour, not a language guarantee: every wait is bounded by wall clock
// (SolveArgs::top_limit, 50 ms) and sets the abort word, upon which the host disables the mode for the handle and
// repeats the sweep level by level (counted in hipkkt_profile).
//
// HAND-OVER CONTRACT (the same one k_top_solve and the factorisation's overlap mode use; pinned by
// tests/test_gpu_parity.py::test_handover_litmus):
//   producer   payload stores are relaxed AGENT-scope atomic stores (global_store ... sc1: written through to the
//              level all XCDs see) -- s_waitcnt vmcnt(0) (each such store has been acknowledged, i.e. performed there)
//              -- block-class fronts: workgroup barrier (all waves' stores) -- ONE relaxed agent-scope atomic on the
//              signal word (forward: add 1 to the parent's counter; backward: the front's epoch word).
//   consumer   one lane polls the signal word with relaxed agent-scope loads -- control dependency (and, block class, a
//              workgroup barrier) -- the payload is read with relaxed agent-scope loads (sc1: never served from the
//              CU's L1 or from a line this XCD's L2 fetched before the producer's store).
// No release / acquire FENCE at agent scope anywhere: on this multi-XCD part such a fence writes back / invalidates the
// whole L2 of the XCD (measured in r02: sweeps 2.8x slower).  What makes the fence-free form sound on gfx950: (i) a
// wave's memory operations are issued in program order and s_waitcnt vmcnt(0) returns only when all its earlier stores
// have completed at the level their scope bits name; (ii) sc1 accesses are coherent at that level per access; (iii) the
// consumer's payload loads are issued after the poll's value is known (branch on it).
#include "kernels.hpp"
#include "solve_common.hpp"
#include <cstdlib>

namespace hipkkt {

constexpr int kTinyFrontMax = 8;        // (solve_kernels.hip: kTinyFront)
constexpr int kChainPF = 4;             // matrix items per wave fetched before the wait (block-class fronts)

__device__ __forceinline__ void chain_add(int* w)
{
    (void)__hip_atomic_fetch_add(w, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void chain_set(int* w, int v)
{
    __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void drain_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

// ------------------------------------------------------------------ forward, block-class front: one workgroup
// k_top_solve's forward step: everything that does not depend on other fronts (matrix items, gather indices, b) is
// fetched BEFORE the wait; afterwards only the handed-over values are loaded.
template <int BS, int PF, int NR>
__device__ __forceinline__ void chain_fwd_block(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int bx, long long t0)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    const int pos = begin + bx;
    // header and row slots in one round (packed records), or the legacy chain of loads
    const char* rec = rec_of(A, R, 0, bx);
    int s, c0, nc, nb, par, nchild, pi = 0;
    int64_t rp, w_off;
    RowGather G;
    G.cnt = 0;
    if (rec) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        if (tid < R.fmax[0]) { pi = rec_idx(rec, tid); G = rec_gather(rec, R.fmax[0], tid); }
        s = h.s; c0 = h.c0; nc = h.nc; nb = h.nb; par = h.par; nchild = h.nchild; rp = h.rp; w_off = h.mat_off;
    } else {
        const FrontDesc fd = T.desc[pos];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; rp = fd.rp; w_off = fd.w_off;
        par = T.sn_parent[s];
        nchild = C.nchild[s];            // children that count themselves in: the ones in chained launches
        if (tid < nc + nb) {
            pi = (tid < nc) ? T.perm[c0 + tid] : 0;
            G = row_gather_lists(T, (int64_t)c0 + rp + tid);
        }
    }
    const int f = nc + nb;
    const double* __restrict__ W = A.tinv + w_off;
    const int fpad = (f + 3) & ~3;
    const int nks = (nc + 7) >> 3, nrb = (f + 63) >> 6, nitF = nrb * nks;
    const int cst = (1 + nks) * fpad;
    double* y = smem;
    double* part = smem + fpad;
#define CH_STAMP(dir, slot) do { if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)(dir) * C.nstamp + (pos - C.lo0)) * 8 + (slot)] = wall_clock64(); } while (0)
    if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)0 * C.nstamp + (pos - C.lo0)) * 8 + 0] = t0;
    ItemRegs rf[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int it = wv + p * NW;
        const int ks = it / nrb, rb = it - ks * nrb;
        const int r = rb * 64 + lane, k0 = 8 * ks;
        const bool live = it < nitF && !(rb * 64 + 63 < k0);          // (wholly above T's diagonal: zeros)
#pragma unroll
        for (int q = 0; q < 8; ++q) rf[p].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
    }
    double bmine[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) bmine[c] = (tid < nc) ? A.b[c * A.ld_b + pi] : 0.0;
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    CH_STAMP(0, 1);
    if (nchild > 0) {
        if (tid == 0) {
            if (wait_flag(C.cnt + s, nchild, C.abort_word, t0, A.top_limit)) chain_set(C.cnt + s, 0);    // (zero again for the next sweep)
            else sh_ok = 0;
        }
        __syncthreads();
        if (!sh_ok) return;
    }
    CH_STAMP(0, 2);
    if (tid < f) {
        gather_add<NR, true>(A, G, bmine);
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c * cst + tid] = bmine[c];
    }
    for (int i = tid + BS; i < f; i += BS) {              // fronts taller than the workgroup
        int pj;
        RowGather Gi;
        if (rec) { pj = rec_idx(rec, i); Gi = rec_gather(rec, R.fmax[0], i); }
        else { pj = (i < nc) ? T.perm[c0 + i] : 0; Gi = row_gather_lists(T, (int64_t)c0 + rp + i); }
        double v[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) v[c] = (i < nc) ? A.b[c * A.ld_b + pj] : 0.0;
        gather_add<NR, true>(A, Gi, v);
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c * cst + i] = v[c];
    }
    __syncthreads();
    CH_STAMP(0, 3);
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int it = wv + p * NW;
        if (it < nitF) {
#pragma unroll
            for (int c = 0; c < NR; ++c) item_apply(rf[p], y + c * cst, f, nc, part + c * cst, fpad, it, nrb, lane);
        }
    }
    for (int it0 = wv + PF * NW; it0 < nitF; it0 += 2 * NW) {
        ItemRegs rr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + u * NW;
            const int ks = it / nrb, rb = it - ks * nrb;
            const int r = rb * 64 + lane, k0 = 8 * ks;
            const bool live = it < nitF && !(rb * 64 + 63 < k0);
#pragma unroll
            for (int q = 0; q < 8; ++q) rr[u].m[q] = (live && r < f && k0 + q < nc) ? W[r + (int64_t)(k0 + q) * f] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (it0 + u * NW < nitF) {
#pragma unroll
                for (int c = 0; c < NR; ++c) item_apply(rr[u], y + c * cst, f, nc, part + c * cst, fpad, it0 + u * NW, nrb, lane);
            }
    }
    __syncthreads();
    CH_STAMP(0, 4);
    for (int i = tid; i < f; i += BS) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double v = lds_sum_strided(part + c * cst + i, nks, fpad);
            if (i < nc) ST_AGENT_F64(A.xp + (int64_t)(c0 + i) * NR + c, v);
            else ST_AGENT_F64(A.uvec + (int64_t)(rp + i - nc) * NR + c, y[c * cst + i] - v);
        }
    }
    drain_stores();
    __syncthreads();
    CH_STAMP(0, 5);
    if (tid == 0 && par >= 0) chain_add(C.cnt + par);
}

// ------------------------------------------------------------------ forward, one wave per front (f <= 64)
template <int BS, int NR>
__device__ __forceinline__ void chain_fwd_wave(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int count, int bx, bool leaf, long long t0)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = bx * (BS / 64) + wv;
    if (item >= count) return;
    const TreeDev& T = A.T;
    int s, c0, nc, nb, par, nchild, pi = 0;
    int64_t rp, f_off;
    RowGather G;
    G.cnt = 0;
    if (const char* rec = rec_of(A, R, 1, item)) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        pi = rec_idx(rec, lane);
        if (!leaf) G = rec_gather(rec, R.fmax[1], lane);
        s = h.s; c0 = h.c0; nc = h.nc; nb = h.nb; par = h.par; nchild = leaf ? 0 : h.nchild; rp = h.rp; f_off = h.mat_off;
    } else {
        const FrontDesc fd = T.desc[begin + item];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; rp = fd.rp; f_off = fd.front_off;
        par = T.sn_parent[s];
        nchild = leaf ? 0 : C.nchild[s];
        pi = (lane < nc) ? T.perm[c0 + lane] : 0;
        if (!leaf && lane < nc + nb) G = row_gather_lists(T, (int64_t)c0 + rp + lane);
    }
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + f_off;
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (lane < nc) ? A.b[c * A.ld_b + pi] : 0.0;
    double lv0[8];                         // the first eight columns of L: they depend on the descriptor only
#pragma unroll
    for (int q = 0; q < 8; ++q) lv0[q] = (q < nc && lane > q && lane < f) ? F[lane + (int64_t)q * f] : 0.0;
    if (nchild > 0) {
        int ok = 1;
        if (lane == 0) {
            if (wait_flag(C.cnt + s, nchild, C.abort_word, t0, A.top_limit)) chain_set(C.cnt + s, 0);
            else ok = 0;
        }
        ok = __builtin_amdgcn_readfirstlane(ok);
        if (!ok) return;
        asm volatile("" ::: "memory");
    }
    if (!leaf && lane < f) gather_add<NR, true>(A, G, y);       // the children's contributions to this row, in child order
    // column sweep: y_l -= L(l,k) y_k
    for (int k0 = 0; k0 < nc; k0 += 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int k = k0 + q;
            lv[q] = k0 == 0 ? lv0[q] : ((k < nc && lane > k && lane < f) ? F[lane + (int64_t)k * f] : 0.0);
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
    if (lane < nc) {
#pragma unroll
        for (int c = 0; c < NR; ++c) ST_AGENT_F64(A.xp + (int64_t)(c0 + lane) * NR + c, y[c]);
    } else if (lane < f) {
#pragma unroll
        for (int c = 0; c < NR; ++c) ST_AGENT_F64(A.uvec + (int64_t)(rp + lane - nc) * NR + c, y[c]);
    }
    drain_stores();
    if (lane == 0 && par >= 0) chain_add(C.cnt + par);
}

// ------------------------------------------------------------------ forward, tiny fronts (f <= 8), eight to a wave
template <int BS, int NR>
__device__ __forceinline__ void chain_fwd_tiny(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int count, int bx, bool leaf, long long t0)
{
    const int sub = threadIdx.x & 7;
    const int item = bx * (BS >> 3) + (int)(threadIdx.x >> 3);
    const bool live = item < count;
    if (!__any(live)) return;
    const TreeDev& T = A.T;
    int s, c0, nc, nb, par, nchild, pi = 0;
    int64_t rp, f_off;
    RowGather G;
    G.cnt = 0;
    if (const char* rec = rec_of(A, R, 2, live ? item : count - 1)) {
        const SolveHdr* h = reinterpret_cast<const SolveHdr*>(rec);
        s = h->s; c0 = h->c0; nc = h->nc; nb = h->nb; par = h->par; nchild = leaf ? 0 : h->nchild; rp = h->rp; f_off = h->mat_off;
        pi = rec_idx(rec, sub);
        if (!leaf) G = rec_gather(rec, R.fmax[2], sub);
    } else {
        const FrontDesc fd = T.desc[begin + (live ? item : count - 1)];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; rp = fd.rp; f_off = fd.front_off;
        par = T.sn_parent[s];
        nchild = leaf ? 0 : C.nchild[s];
        pi = (sub < nc) ? T.perm[c0 + sub] : 0;
        if (!leaf && sub < nc + nb) G = row_gather_lists(T, (int64_t)c0 + rp + sub);
    }
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + f_off;
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (sub < nc) ? A.b[c * A.ld_b + pi] : 0.0;
    double lv[kTinyFrontMax];
#pragma unroll
    for (int k = 0; k < kTinyFrontMax; ++k) lv[k] = (k < nc && sub > k && sub < f) ? F[sub + k * f] : 0.0;
    if (!leaf) {
        // every 8-lane group waits for its own front's children; the wave goes on when all of them have
        bool ok = true;
        if (live && nchild > 0 && sub == 0) {
            if (wait_flag(C.cnt + s, nchild, C.abort_word, t0, A.top_limit)) chain_set(C.cnt + s, 0);
            else ok = false;
        }
        if (__any(!ok)) return;
        asm volatile("" ::: "memory");
        if (sub < f) gather_add<NR, true>(A, G, y);
    }
#pragma unroll
    for (int k = 0; k < kTinyFrontMax; ++k) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double yk = __shfl(y[c], k, 8);                    // every lane takes part (no divergence here)
            if (k < nc) y[c] = fma(-lv[k], yk, y[c]);
        }
    }
    if (live) {
        if (sub < nc) {
#pragma unroll
            for (int c = 0; c < NR; ++c) ST_AGENT_F64(A.xp + (int64_t)(c0 + sub) * NR + c, y[c]);
        } else if (sub < f) {
#pragma unroll
            for (int c = 0; c < NR; ++c) ST_AGENT_F64(A.uvec + (int64_t)(rp + sub - nc) * NR + c, y[c]);
        }
    }
    drain_stores();
    if (live && sub == 0 && par >= 0) chain_add(C.cnt + par);
}

// Which segment a workgroup belongs to: the segments' first workgroups ascend (a handful of scalar compares).
__device__ __forceinline__ int chain_find_seg(const ChainArgs& C, int bx)
{
    int k = 0;
    while (k + 1 < C.nseg && bx >= C.seg[k + 1].wg0) ++k;
    return k;
}

template <int BS, int NR, int PF>
__global__ __launch_bounds__(BS) void k_fwd_chain(SolveArgs A, ChainArgs C)
{
    const long long t0 = wall_clock64();
    const int k = chain_find_seg(C, blockIdx.x);
    const int begin = C.seg[k].begin, nblock = C.seg[k].nblock, nwave = C.seg[k].nwave, ntiny = C.seg[k].ntiny;
    const bool leaf = C.seg[k].leaf != 0;
    const int b = (int)blockIdx.x - C.seg[k].wg0;
    const int nwb = (nwave + BS / 64 - 1) / (BS / 64);
    if (b < nblock) { chain_fwd_block<BS, PF, NR>(A, C, C.seg[k].rec, begin, b, t0); return; }
    if (b < nblock + nwb) chain_fwd_wave<BS, NR>(A, C, C.seg[k].rec, begin + nblock, nwave, b - nblock, leaf, t0);
    else chain_fwd_tiny<BS, NR>(A, C, C.seg[k].rec, begin + nblock + nwave, ntiny, b - nblock - nwb, leaf, t0);
}

// ------------------------------------------------------------------ backward
// A front waits for its PARENT's step if the parent is part of this launch (schedule position < C.hi; a parent beyond
// that was swept by an earlier kernel) and reads the ancestors' solution entries it needs: all of them are final by then
// (the parent waited for its parent, and so on up).
__device__ __forceinline__ bool chain_parent_pending(const TreeDev& T, const ChainArgs& C, int par)
{
    return par >= 0 && T.spos[par] < C.hi;
}

template <int BS, int PB, int NR>
__device__ __forceinline__ void chain_bwd_block(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int bx, long long t0)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int sh_ok;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    const int pos = begin + bx;
    const char* rec = rec_of(A, R, 0, bx);
    int s, c0, nc, nb, par;
    int64_t rp, w_off;
    int ridx = -1;                // own columns: the caller's row of the final store; rows below: the ancestor's entry of xp
    if (rec) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        if (tid < R.fmax[0]) ridx = rec_idx(rec, tid);
        s = h.s; c0 = h.c0; nc = h.nc; nb = h.nb; par = h.par; rp = h.rp; w_off = h.mat_off;
    } else {
        const FrontDesc fd = T.desc[pos];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; rp = fd.rp; w_off = fd.w_off;
        par = T.sn_parent[s];
        if (tid < nc) ridx = T.perm[c0 + tid];
        else if (tid < nc + nb) ridx = T.rows[rp + tid - nc];
    }
    const int f = nc + nb;
    const double* __restrict__ Wt = A.tinv + w_off + (int64_t)f * nc;          // W'(j, r) at j + r*nc
    const int fpad = (f + 3) & ~3, ncpad = (nc + 3) & ~3;
    const int ncb = (nc + 63) >> 6, nrs = (f + 7) >> 3, nitB = ncb * nrs;
    const int cst = fpad + nrs * ncpad;
    double* z = smem;
    double* part = smem + fpad;
#define CH_STAMP(dir, slot) do { if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)(dir) * C.nstamp + (pos - C.lo0)) * 8 + (slot)] = wall_clock64(); } while (0)
    ItemRegs rbk[PB];
#pragma unroll
    for (int p = 0; p < PB; ++p) {
        const int it = wv + p * NW;
        const int rs = it / ncb, cb = it - rs * ncb;
        const int j = cb * 64 + lane, r0 = 8 * rs;
        const bool live = it < nitB && !(r0 + 7 < cb * 64);          // (rows above the column block's diagonal: zeros)
#pragma unroll
        for (int q = 0; q < 8; ++q) rbk[p].m[q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
    }
    const double dinv = (tid < nc) ? A.Dinv[c0 + tid] : 0.0;
    const bool pending = chain_parent_pending(T, C, par);
    if (A.top_stamps && tid == 0) A.top_stamps[((int64_t)1 * C.nstamp + (pos - C.lo0)) * 8 + 0] = t0;
    if (tid == 0) sh_ok = 1;
    __syncthreads();
    CH_STAMP(1, 1);
    if (pending) {
        if (tid == 0 && !wait_flag(C.done + par, C.epoch, C.abort_word, t0, A.top_limit)) sh_ok = 0;
        __syncthreads();
        if (!sh_ok) return;
    }
    CH_STAMP(1, 2);
    {
        double zv[NR];
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double* xc = A.xp + c;
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
                                      : -LD_AGENT_F64(xc + (int64_t)(rec ? rec_idx(rec, i) : T.rows[rp + i - nc]) * NR);
    }
    __syncthreads();
    CH_STAMP(1, 3);
    auto apply = [&](const ItemRegs& R, int it) {
        const int rs = it / ncb, cb = it - rs * ncb;
        const int j = cb * 64 + lane, r0 = 8 * rs;
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            double acc = 0.0;
#pragma unroll
            for (int q = 0; q < 8; ++q) acc = fma(R.m[q], (r0 + q < f) ? z[c * cst + r0 + q] : 0.0, acc);
            if (j < nc) part[c * cst + rs * ncpad + j] = acc;
        }
    };
#pragma unroll
    for (int p = 0; p < PB; ++p)
        if (wv + p * NW < nitB) apply(rbk[p], wv + p * NW);
    for (int it0 = wv + PB * NW; it0 < nitB; it0 += 2 * NW) {
        ItemRegs rr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int it = it0 + u * NW;
            const int rs = it / ncb, cb = it - rs * ncb;
            const int j = cb * 64 + lane, r0 = 8 * rs;
            const bool live = it < nitB && !(r0 + 7 < cb * 64);
#pragma unroll
            for (int q = 0; q < 8; ++q) rr[u].m[q] = (live && j < nc && r0 + q < f) ? Wt[j + (int64_t)(r0 + q) * nc] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (it0 + u * NW < nitB) apply(rr[u], it0 + u * NW);
    }
    __syncthreads();
    CH_STAMP(1, 4);
    for (int j = tid; j < nc; j += BS) {
        const int pi = j == tid ? ridx : (rec ? rec_idx(rec, j) : T.perm[c0 + j]);
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double v = lds_sum_strided(part + c * cst + j, nrs, ncpad);
            ST_AGENT_F64(A.xp + (int64_t)(c0 + j) * NR + c, v);
            A.out[c * A.ld_out + pi] = v;
        }
    }
    drain_stores();
    __syncthreads();
    CH_STAMP(1, 5);
    if (tid == 0) chain_set(C.done + s, C.epoch);
}
#undef CH_STAMP

template <int BS, int NR>
__device__ __forceinline__ void chain_bwd_wave(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int count, int bx, long long t0)
{
    const int lane = threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int item = bx * (BS / 64) + wv;
    if (item >= count) return;
    const TreeDev& T = A.T;
    int s, c0, nc, nb, par, idx = 0;
    int64_t f_off;
    if (const char* rec = rec_of(A, R, 1, item)) {
        const SolveHdr h = *reinterpret_cast<const SolveHdr*>(rec);
        idx = rec_idx(rec, lane);
        s = h.s; c0 = h.c0; nc = h.nc; nb = h.nb; par = h.par; f_off = h.mat_off;
    } else {
        const FrontDesc fd = T.desc[begin + item];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; f_off = fd.front_off;
        par = T.sn_parent[s];
        idx = (lane < nc) ? T.perm[c0 + lane] : ((lane < nc + nb) ? T.rows[fd.rp + lane - nc] : 0);
    }
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + f_off;
    const double di = (lane < nc) ? A.Dinv[c0 + lane] : 0.0;
    const int ri = idx, pi = idx;
    double lv0[8];                         // the last eight columns of L (the first ones the sweep uses)
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int j = nc - 1 - q;
        lv0[q] = (j >= 0 && lane > j && lane < f) ? F[lane + (int64_t)j * f] : 0.0;
    }
    if (chain_parent_pending(T, C, par)) {
        int ok = 1;
        if (lane == 0 && !wait_flag(C.done + par, C.epoch, C.abort_word, t0, A.top_limit)) ok = 0;
        ok = __builtin_amdgcn_readfirstlane(ok);
        if (!ok) return;
        asm volatile("" ::: "memory");
    }
    // lane = row: y_r = D^{-1} x_r for the front's own columns, the ancestors' solution below
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (lane < f) ? LD_AGENT_F64(A.xp + (int64_t)(lane < nc ? c0 + lane : ri) * NR + c) : 0.0;
    if (lane < nc) {
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c] *= di;
    }
    for (int j1 = nc; j1 > 0; j1 -= 8) {
        double lv[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int j = j1 - 1 - q;
            lv[q] = j1 == nc ? lv0[q] : ((j >= 0 && lane > j && lane < f) ? F[lane + (int64_t)j * f] : 0.0);
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
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            ST_AGENT_F64(A.xp + (int64_t)(c0 + lane) * NR + c, y[c]);
            A.out[c * A.ld_out + pi] = y[c];
        }
    }
    drain_stores();
    if (lane == 0) chain_set(C.done + s, C.epoch);
}

template <int BS, int NR>
__device__ __forceinline__ void chain_bwd_tiny(const SolveArgs& A, const ChainArgs& C, const RecSeg& R, int begin, int count, int bx, long long t0)
{
    const int sub = threadIdx.x & 7;
    const int item = bx * (BS >> 3) + (int)(threadIdx.x >> 3);
    const bool live = item < count;
    if (!__any(live)) return;
    const TreeDev& T = A.T;
    int s, c0, nc, nb, par, idx = 0;
    int64_t f_off;
    if (const char* rec = rec_of(A, R, 2, live ? item : count - 1)) {
        const SolveHdr* h = reinterpret_cast<const SolveHdr*>(rec);
        s = h->s; c0 = h->c0; nc = h->nc; nb = h->nb; par = h->par; f_off = h->mat_off;
        idx = rec_idx(rec, sub);
    } else {
        const FrontDesc fd = T.desc[begin + (live ? item : count - 1)];
        s = fd.s; c0 = fd.c0; nc = fd.nc; nb = fd.nb; f_off = fd.front_off;
        par = T.sn_parent[s];
        idx = (sub < nc) ? T.perm[c0 + sub] : ((sub < nc + nb) ? T.rows[fd.rp + sub - nc] : 0);
    }
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + f_off;
    const double di = (sub < nc) ? A.Dinv[c0 + sub] : 0.0;
    const int ri = idx, pi = idx;
    double lv[kTinyFrontMax];
#pragma unroll
    for (int j = 0; j < kTinyFrontMax; ++j) lv[j] = (j < nc && sub > j && sub < f) ? F[sub + j * f] : 0.0;
    {
        bool ok = true;
        if (live && sub == 0 && chain_parent_pending(T, C, par) && !wait_flag(C.done + par, C.epoch, C.abort_word, t0, A.top_limit)) ok = false;
        if (__any(!ok)) return;
        asm volatile("" ::: "memory");
    }
    double y[NR];
#pragma unroll
    for (int c = 0; c < NR; ++c) y[c] = (sub < f) ? LD_AGENT_F64(A.xp + (int64_t)(sub < nc ? c0 + sub : ri) * NR + c) : 0.0;
    if (sub < nc) {
#pragma unroll
        for (int c = 0; c < NR; ++c) y[c] *= di;
    }
#pragma unroll
    for (int j = kTinyFrontMax - 1; j >= 0; --j) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            const double sum = group8_sum(lv[j] * y[c]);
            if (j < nc && sub == j) y[c] -= sum;
        }
    }
    if (live && sub < nc) {
#pragma unroll
        for (int c = 0; c < NR; ++c) {
            ST_AGENT_F64(A.xp + (int64_t)(c0 + sub) * NR + c, y[c]);
            A.out[c * A.ld_out + pi] = y[c];
        }
    }
    drain_stores();
    if (live && sub == 0) chain_set(C.done + s, C.epoch);
}

template <int BS, int NR, int PF>
__global__ __launch_bounds__(BS) void k_bwd_chain(SolveArgs A, ChainArgs C)
{
    const long long t0 = wall_clock64();
    const int k = chain_find_seg(C, blockIdx.x);
    const int begin = C.seg[k].begin, nblock = C.seg[k].nblock, nwave = C.seg[k].nwave, ntiny = C.seg[k].ntiny;
    const int b = (int)blockIdx.x - C.seg[k].wg0;
    const int nwb = (nwave + BS / 64 - 1) / (BS / 64);
    if (b < nblock) { chain_bwd_block<BS, PF, NR>(A, C, C.seg[k].rec, begin, b, t0); return; }
    if (b < nblock + nwb) chain_bwd_wave<BS, NR>(A, C, C.seg[k].rec, begin + nblock, nwave, b - nblock, t0);
    else chain_bwd_tiny<BS, NR>(A, C, C.seg[k].rec, begin + nblock + nwave, ntiny, b - nblock - nwb, t0);
}

// ------------------------------------------------------------------ host side
#define HIPKKT_NR_SWITCH(nr, ...)                                                                 \
    do {                                                                                         \
        if ((nr) == 1) { constexpr int NR = 1; __VA_ARGS__; }                                    \
        else if ((nr) == 2) { constexpr int NR = 2; __VA_ARGS__; }                               \
        else { constexpr int NR = 4; __VA_ARGS__; }                                              \
    } while (0)

// (PF = 4 matrix items per wave before the wait: 109 / 93 registers, two workgroups per CU.  PF = 1 -- 66 / 53 registers,
//  three / four per CU -- was measured: the same for two chained levels, much slower up the narrow top, whose fronts
//  then fetch most of their matrix behind the wait: sweep pair 0.307 against 0.268 ms.)
static void init_chain_lds()
{
    static PerDeviceOnce once;
    once.run([]() {
        hipError_t e = hipSuccess;
        auto set = [&](auto k, int bytes) { if (e == hipSuccess) e = set_max_lds(k, bytes); };
#define HIPKKT_SET_CHAIN(NRV) set(k_fwd_chain<kChainBS, NRV, kChainPF>, 150 * 1024); set(k_bwd_chain<kChainBS, NRV, kChainPF>, 150 * 1024);
        HIPKKT_SET_CHAIN(1) HIPKKT_SET_CHAIN(2) HIPKKT_SET_CHAIN(4)
#undef HIPKKT_SET_CHAIN
        return e;
    });
}

void launch_fwd_chain(const SolveArgs& a, const ChainArgs& c, int nwg, size_t lds, hipStream_t st, int nr)
{
    if (nwg <= 0) return;
    init_chain_lds();
    HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_fwd_chain<kChainBS, NR, kChainPF>), dim3(nwg), dim3(kChainBS), lds * NR, st, a, c));
}
void launch_bwd_chain(const SolveArgs& a, const ChainArgs& c, int nwg, size_t lds, hipStream_t st, int nr)
{
    if (nwg <= 0) return;
    init_chain_lds();
    HIPKKT_NR_SWITCH(nr, hipLaunchKernelGGL((k_bwd_chain<kChainBS, NR, kChainPF>), dim3(nwg), dim3(kChainBS), lds * NR, st, a, c));
}

}  // namespace hipkkt
