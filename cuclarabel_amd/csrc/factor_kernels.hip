// Numeric supernodal multifrontal LDL^T (gfx950, wave64).
//
// Replaces QDLDL.refactor! (call site /root/reference/src/kktsolvers/direct-ldl/
// directldl_qdldl.jl:72-81).  Same pivot rule -- D_k*sign_k < eps  =>  D_k = sign_k*delta -- applied
// at pivot time inside the dense panel; no pivoting, static structure.
//
// Per level of the assembly tree, three kernels:
//   k_front_wave    fronts that fit one wave's LDS slice (f <= 64): assemble, factor and form the
//                   update block entirely in LDS, one wave per front, four fronts per workgroup.
//   k_panel         larger fronts, one workgroup each: zero + scatter K (+ static eps*sign) +
//                   gather the children's update entries that land in the panel, then a
//                   right-looking blocked LDL^T with the current block column staged in LDS.
//   k_schur         the update block U = (children pass-through) - L21 D L21^T, tiled 64x64 over
//                   many workgroups per front so the few big fronts near the root still fill CUs.
// Every extend-add is a gather by the owner of the target, children in a fixed order, each child
// through an injective ascending map: sums are reproducible run to run (no atomics).
//
// Loads are issued in explicit batches before first use: a one-load-then-use loop serialises
// on HBM/L2 latency (the first version of these kernels ran 10x slower for that reason).
#include "kernels.hpp"
#include "knobs.hpp"
#include <utility>

namespace hipkkt {

__device__ inline double rl_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}
// Ordering of LDS traffic inside ONE wave: the LDS executes a wave's DS instructions in issue order, so
// a ds_write followed by another lane's ds_read of that address needs no hardware fence -- only the
// compiler must not reorder or cache across the point.  (A seq_cst wavefront fence also drains
// outstanding global stores/loads, which put ~1 us into every pivot step.)
#define WAVE_FENCE() asm volatile("" ::: "memory")

// LDS += without the read-wait-add-write chain: ds_add_f64 (gfx90a+) is fire-and-forget, and the LDS executes one
// wave's operations in issue order, so a wave's adds to one address still land in program order (deterministic sums).
__device__ __forceinline__ void lds_add(double* p, double v)
{
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
// ---- overlap mode (FactorArgs::ov): hand-over between concurrently running kernels, the Guideline-16 recipe of the
// HIP guide in its write-through form -- payload stored sc1 (relaxed agent-scope atomic stores), every storing wave
// drains vmcnt, the workgroup meets, ONE lane bumps / stores the counter; consumers poll the counter with relaxed
// agent-scope loads and read the payload with agent-scope (L1 / L2-bypassing) loads only.  Every spin is bounded by
// wall clock; on expiry flags[2] is set, everyone leaves, and the host repeats the factorisation level by level.
//
// THE MEMORY-ORDERING CONTRACT (shared with k_top_solve / k_top_solve_sliced in solve_kernels.hip and the chained
// sweep kernels in chain_kernels.hip; there is no release / acquire FENCE at agent scope anywhere in csrc/):
//   (P1) every payload store is a relaxed agent-scope atomic store: `global_store ... sc1`, performed at the level all
//        XCDs share, never left dirty in this XCD's L2;
//   (P2) `s_waitcnt vmcnt(0)` in every storing wave BEFORE the signal: on gfx9 stores count in vmcnt and the counter
//        only drops when the store has been acknowledged at the level its scope bits name -- after it, the wave's
//        payload stores are visible device-wide.  This is the instruction that orders payload before signal; without
//        it the signal (a different address, possibly a different channel) can be performed first;
//   (P3) workgroup barrier (block-class producers: all waves have passed P2), then ONE relaxed agent-scope atomic on the
//        signal word;
//   (C1) the consumer polls the signal word with relaxed agent-scope loads, branches on the value (a wave issues its
//        memory operations in program order, and the payload loads are issued behind the branch), block-class:
//        workgroup barrier;
//   (C2) every payload load is a relaxed agent-scope atomic load (`... sc1`: served from the shared level, not from the
//        CU's L1 nor from a line this XCD's L2 fetched before the producer's store).
// Why no fence: an agent-scope release on this multi-XCD part writes the XCD's whole L2 back and an acquire invalidates
// it (measured in r02: the sweeps 2.8x slower); P1 + C2 make the few words that are handed over coherent one by one
// instead, and P2 / C1 order them.  tests/test_gpu_parity.py::test_handover_litmus runs exactly this protocol
// (hipkkt_selftest_handover, below) on workgroup pairs placed on different XCDs: zero stale reads in 10^8 handed-over
// words with the contract; with plain payload accesses instead of P1 / C2 EVERY word is read stale (the consumer's XCD
// serves the line it fetched before), which pins P1 / C2 as necessary.  The variant without P2 has not shown a stale
// word on this part (the memory side performs a wave's stores in issue order in practice); P2 stays because the ISA
// promises no order between stores to different addresses -- it is the one part of the contract that rests on the
// architecture manual rather than on an observed failure, and it costs what the drain costs (~0.5 us per hand-over).
#define OV_LD(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define OV_ST(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
// (who / info: which wait expired first -- 1 a panel for a child's tiles, 2 a tile for its panel's blocks, 3 the gate -- and
//  the supernode or launch it waited for: abort_word[1..4], printed with the fall-back message)
// (seen_min, nullable: lowered to the counter's value as found -- a caller that polls for a sequence of rising targets
//  skips the polls a value already seen covers)
__device__ inline bool ov_wait_ge(const int* counter, int target, int* abort_word, long long t0, long long limit, int who = 0, int info = 0,
                                  int* seen_min = nullptr)
{
    for (;;) {
        const int seen = __hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (seen >= target) { if (seen_min && seen < *seen_min) *seen_min = seen; return true; }
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (wall_clock64() - t0 > limit) {            // (50 ms at 100 MHz: far beyond any real factorisation step)
            if (atomicCAS(abort_word, 0, 1) == 0) { abort_word[1] = who; abort_word[2] = info; abort_word[3] = seen; abort_word[4] = target; }
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}
// first index t in [0, n) with arr[t] >= v  (arr ascending)
__device__ inline int lower_bound_dev(const int* __restrict__ arr, int n, int v)
{
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (arr[mid] < v) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// =====================================================================================
//  small fronts: one wave per front, LDS slice = panel (f x nc) + update block (nb x nb)
// =====================================================================================
__global__ __launch_bounds__(256) void k_front_wave(FactorArgs A, int begin, int count, int slice)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wv;
    if (item >= count) return;
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + item];
    const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int f = nc + nb;
    double* __restrict__ F = A.fronts + fd.front_off;
    double* __restrict__ U = A.upd + fd.upd_off;
    double* P = smem + (size_t)wv * slice;       // f x nc, ld f
    double* Us = P + f * nc;                     // nb x nb, ld nb

    for (int i = lane; i < f * nc + nb * nb; i += 64) P[i] = 0.0;
    WAVE_FENCE();
    {   // K entries
        const int64_t e0 = fd.kptr;
        const int ne = fd.nk;
        for (int e = lane; e < ne; e += 64) P[T.kdst[e0 + e]] = A.Kval[T.ksrc[e0 + e]];
    }
    WAVE_FENCE();
    if (A.eps) {
        const double eps = *A.eps;
        if (lane < nc) P[lane + lane * f] += eps * (double)T.psign[c0 + lane];
    }
    // children, one after the other: panel part and pass-through part in one sweep.  The children's headers (id, row
    // range, update-block offset: a chain of three dependent loads) are fetched for 64 children at once, lane = child,
    // and handed out with v_readlane: per child only the data rounds remain -- this kernel is bound by exactly that
    // latency (a level-1 front has a dozen leaf children).
    for (int ce0 = T.child_ptr[s]; ce0 < T.child_ptr[s + 1]; ce0 += 64) {
        const int nch = min(64, T.child_ptr[s + 1] - ce0);
        int h_crp = 0, h_nbc = 0, h_ulo = 0, h_uhi = 0;
        if (lane < nch) {
            const int c = T.child_idx[ce0 + lane];
            const int64_t crp = T.rowptr[c];
            const int64_t uo = T.upd_off[c];
            h_crp = (int)crp;                                   // (row structure is int32-indexed: hipkkt.hip checks)
            h_nbc = (int)(T.rowptr[c + 1] - crp);
            h_ulo = (int)(uo & 0xffffffff);
            h_uhi = (int)(uo >> 32);
        }
        for (int ci = 0; ci < nch; ++ci) {
            const int crp = __builtin_amdgcn_readlane(h_crp, ci);
            const int nbc = __builtin_amdgcn_readlane(h_nbc, ci);
            const int64_t uo = ((int64_t)__builtin_amdgcn_readlane(h_uhi, ci) << 32) | (uint32_t)__builtin_amdgcn_readlane(h_ulo, ci);
            const double* __restrict__ Uc = A.upd + uo;
            const int* __restrict__ relc = T.rel + crp;
            WAVE_FENCE();
            if (nbc <= 10) {
                // a small child: its whole lower triangle in ONE load round, lane = entry (a, b) -- distinct targets
                int b = 0, rem = lane;                       // entry index -> (a, b): column b holds nbc - b entries
                while (b < nbc && rem >= nbc - b) { rem -= nbc - b; ++b; }
                if (b < nbc) {
                    const int a = b + rem;
                    const int ra = relc[a], rb = relc[b];
                    const double v = Uc[a + (int64_t)b * nbc];
                    if (rb < nc) lds_add(P + ra + rb * f, v);
                    else lds_add(Us + (ra - nc) + (rb - nc) * nb, v);
                }
                continue;
            }
            for (int b = 0; b < nbc; ++b) {
                const int rb = relc[b];
                for (int a = b + lane; a < nbc; a += 64) {
                    const int ra = relc[a];
                    const double v = Uc[a + (int64_t)b * nbc];
                    if (rb < nc) lds_add(P + ra + rb * f, v);
                    else lds_add(Us + (ra - nc) + (rb - nc) * nb, v);
                }
            }
        }
    }
    WAVE_FENCE();
    // factor the panel: lane = row
    const double my_sg = (lane < nc) ? (double)T.psign[c0 + lane] : 1.0;
    for (int k = 0; k < nc; ++k) {
        double d = P[k + k * f];
        const double sg = rl_f64(my_sg, k);
        const bool reg = (d * sg < A.dyn_eps);
        if (reg) d = sg * A.dyn_delta;
        const double dinv = 1.0 / d;
        if (lane == 0) {
            if (reg) atomicAdd(&A.flags[0], 1);
            if (!isfinite(dinv)) A.flags[1] = 1;
            A.Dinv[c0 + k] = dinv;
        }
        if (lane > k && lane < f) {
            const double vik = P[lane + k * f];
            const double lik = vik * dinv;
            const int jmax = min(lane, nc - 1);
            for (int j = k + 1; j <= jmax; ++j) P[lane + j * f] -= lik * P[j + k * f];
        }
        WAVE_FENCE();
        if (lane > k && lane < f) P[lane + k * f] *= dinv;
        if (lane == k) P[k + k * f] = d;
        WAVE_FENCE();
    }
    // write L and D
    for (int i = lane; i < f * nc; i += 64) {
        const int j = i / f, r = i - j * f;
        if (r >= j) F[i] = P[i];
    }
    // update block: U(a,b) = Us(a,b) - sum_k L(a,k) d_k L(b,k), lane = row a
    if (lane < nb) {
        const int a = lane;
        for (int b = 0; b <= a; ++b) {
            double acc = 0.0;
            for (int k = 0; k < nc; ++k) acc = fma(P[nc + a + k * f] * P[k + k * f], P[nc + b + k * f], acc);
            U[a + (int64_t)b * nb] = Us[a + b * nb] - acc;
        }
    }
}

// =====================================================================================
//  tiny fronts (f <= 8): eight lanes per front, eight fronts per wave, 1 KB of LDS each.  Same steps and the
//  same operation order as k_front_wave; these are most of a KKT tree's leaves, and a wave apiece leaves the
//  machine short of wave slots (the kernel is bound by waves in flight, not by bytes).
// =====================================================================================
constexpr int kTinySlice = 128;        // doubles: f*nc + nb*nb <= 8*8 + 7*7 (f <= 8)
__global__ __launch_bounds__(256) void k_front_tiny(FactorArgs A, int begin, int count)
{
    __shared__ __attribute__((aligned(16))) double smem_t[32 * kTinySlice];
    const int sub = threadIdx.x & 7, grp = threadIdx.x >> 3;
    const int item = blockIdx.x * 32 + grp;
    if (item >= count) return;                    // (no wave-wide operation below: groups are independent)
    const TreeDev& T = A.T;
    const FrontDesc fd = T.desc[begin + item];
    const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int f = nc + nb;
    double* __restrict__ F = A.fronts + fd.front_off;
    double* __restrict__ U = A.upd + fd.upd_off;
    double* P = smem_t + grp * kTinySlice;        // f x nc, ld f
    double* Us = P + f * nc;                      // nb x nb, ld nb

    for (int i = sub; i < f * nc + nb * nb; i += 8) P[i] = 0.0;
    WAVE_FENCE();
    {
        const int64_t e0 = fd.kptr;
        const int ne = fd.nk;
        for (int e = sub; e < ne; e += 8) P[T.kdst[e0 + e]] = A.Kval[T.ksrc[e0 + e]];
    }
    WAVE_FENCE();
    if (A.eps) {
        const double eps = *A.eps;
        if (sub < nc) P[sub + sub * f] += eps * (double)T.psign[c0 + sub];
    }
    for (int ce = T.child_ptr[s]; ce < T.child_ptr[s + 1]; ++ce) {
        const int c = T.child_idx[ce];
        const int64_t crp = T.rowptr[c];
        const int nbc = (int)(T.rowptr[c + 1] - crp);
        const double* __restrict__ Uc = A.upd + T.upd_off[c];
        const int* __restrict__ relc = T.rel + crp;
        WAVE_FENCE();
        for (int b = 0; b < nbc; ++b) {
            const int rb = relc[b];
            for (int a = b + sub; a < nbc; a += 8) {
                const int ra = relc[a];
                const double v = Uc[a + (int64_t)b * nbc];
                if (rb < nc) lds_add(P + ra + rb * f, v);
                else lds_add(Us + (ra - nc) + (rb - nc) * nb, v);
            }
        }
    }
    WAVE_FENCE();
    for (int k = 0; k < nc; ++k) {
        double d = P[k + k * f];
        const double sg = (double)T.psign[c0 + k];
        const bool reg = (d * sg < A.dyn_eps);
        if (reg) d = sg * A.dyn_delta;
        const double dinv = 1.0 / d;
        if (sub == 0) {
            if (reg) atomicAdd(&A.flags[0], 1);
            if (!isfinite(dinv)) A.flags[1] = 1;
            A.Dinv[c0 + k] = dinv;
        }
        if (sub > k && sub < f) {
            const double vik = P[sub + k * f];
            const double lik = vik * dinv;
            const int jmax = min(sub, nc - 1);
            for (int j = k + 1; j <= jmax; ++j) P[sub + j * f] -= lik * P[j + k * f];
        }
        WAVE_FENCE();
        if (sub > k && sub < f) P[sub + k * f] *= dinv;
        if (sub == k) P[k + k * f] = d;
        WAVE_FENCE();
    }
    for (int i = sub; i < f * nc; i += 8) {
        const int j = i / f, r = i - j * f;
        if (r >= j) F[i] = P[i];
    }
    if (sub < nb) {
        const int a = sub;
        for (int b = 0; b <= a; ++b) {
            double acc = 0.0;
            for (int k = 0; k < nc; ++k) acc = fma(P[nc + a + k * f] * P[k + k * f], P[nc + b + k * f], acc);
            U[a + (int64_t)b * nb] = Us[a + b * nb] - acc;
        }
    }
}

// =====================================================================================
//  panel kernel: one workgroup per front, the whole f x nc panel resident in LDS
//  (the symbolic phase splits wider supernodes so that f*nc fits: SymbolicOptions::panel_cap).
//  Global traffic: K values and the children's update entries in (batched, independent loads),
//  L and D out once.
//    children   every wave owns the panel columns j = wave (mod #waves) and applies the extend-add
//               items of its columns in order -- no barrier between children, fixed summation order.
//    per 16-column block: diagonal block in the registers of one wave (readlane broadcasts),
//               rows below by one thread per row (TRSM against the d*L copy in LDS), trailing
//               update by 4x4 register tiles whose rows are 64 apart (bank-conflict free).
// =====================================================================================
typedef double d4_t __attribute__((ext_vector_type(4)));

// 1/d from the hardware seed (v_rcp_f64) and two Newton steps: within an ulp of the quotient at a fraction of the
// latency of the IEEE division sequence -- it sits on the serial pivot chain.
__device__ inline double fast_recip(double d)
{
    double r = __builtin_amdgcn_rcp(d);
    r = fma(fma(-d, r, 1.0), r, r);
    r = fma(fma(-d, r, 1.0), r, r);
    return r;          // d = 0 or not finite: NaN (the caller only asks whether the result is finite)
}
constexpr int NB = 16;
// The panel lives in LDS as a TRAPEZOID: column j holds rows j .. f-1, columns back to back.  Entry (i, j), i >= j,
// sits at i + pcol(j, f); nothing above the diagonal is ever stored or read.  That is f*nc - nc(nc-1)/2 doubles
// instead of f*nc: a 231 x 87 panel fits where the rectangle would have to be split (one more tree level).
__device__ __forceinline__ int pcol(int j, int f) { return (j * (2 * f - 1 - j)) >> 1; }
constexpr int kBdCols = 128;       // trailing columns per block whose d*L copy is kept (nc - 16 <= 128 enforced by host)

// apply the extend-add items [i0, i1) (whole columns, owned by this wave) into the LDS panel P (ld f):
// eight items in flight -- one round of descriptor loads, one round of (rel, value) loads, then
// the LDS adds in item order
__device__ inline int64_t uniform_i64(int64_t v)
{
    const int lo = __builtin_amdgcn_readfirstlane((int)(v & 0xffffffff));
    const int hi = __builtin_amdgcn_readfirstlane((int)(v >> 32));
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
// SLICED: the workgroup holds the top nc rows and the front rows [r_lo, r_lo + rs) only (local row nc + r - r_lo)
template <bool SLICED, bool COH>
__device__ inline void apply_items_panel(const TreeDev& T, const double* __restrict__ upd, double* P, int f,
                                         int64_t i0_, int64_t i1_, int lane, int nc, int r_lo, int rs, const ExtItem& first,
                                         long long* dbg = nullptr)
{
    // the range is the same for every lane
    const int64_t i0 = uniform_i64(i0_), i1 = uniform_i64(i1_);
    constexpr int IF = 16;        // pieces in flight
    for (int64_t ii = i0; ii < i1; ii += 64) {
        // the next 64 descriptors in ONE vector load round (lane l fetches item ii + l); their fields are handed out
        // with v_readlane as the pieces are issued -- no scalar-load latency per batch.  The first round was issued by
        // the caller before the panel was zeroed (static data: its latency hides behind the kernel's first phases)
        const ExtItem mine = (ii == i0) ? first : T.items[min(ii + lane, i1 - 1)];
        const int lo = (int)(mine.uoff & 0xffffffff), hi = (int)(mine.uoff >> 32);
        // (sliced: a piece wholly below the top block and outside this slice's rows brings nothing)
        const int mycnt = (SLICED && mine.rfirst >= nc && (mine.rlast < r_lo || mine.rfirst >= r_lo + rs)) ? 0 : mine.cnt;
        const int nhere = (int)min((int64_t)64, i1 - ii);
        for (int q0 = 0; q0 < nhere; q0 += IF) {
            double v[IF];
            int tg[IF];
#pragma unroll
            for (int q = 0; q < IF; ++q) {
                const int j = min(q0 + q, 63);
                const int64_t uoff = ((int64_t)__builtin_amdgcn_readlane(hi, j) << 32) | (uint32_t)__builtin_amdgcn_readlane(lo, j);
                const int relstart = __builtin_amdgcn_readlane(mine.relstart, j);
                const int cnt = (q0 + q < nhere) ? __builtin_amdgcn_readlane(mycnt, j) : 0;
                const int tcol = __builtin_amdgcn_readlane(mine.tcol, j);
                const bool ok = lane < cnt;
                v[q] = ok ? (COH ? OV_LD(upd + uoff + lane) : upd[uoff + lane]) : 0.0;     // (overlap mode: written by a
                                                                                           // kernel that may still be running)
                if (!SLICED) {
                    tg[q] = ok ? T.rel[relstart + lane] + pcol(tcol, f) : -1;
                } else {
                    int r = ok ? T.rel[relstart + lane] : -1;
                    if (r >= nc) r = (r >= r_lo && r < r_lo + rs) ? nc + (r - r_lo) : -1;
                    tg[q] = r >= 0 ? r + pcol(tcol, f) : -1;
                }
            }
            if (dbg && q0 == 0 && ii == i0) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); if (lane == 0) dbg[0] = wall_clock64(); }
#pragma unroll
            for (int q = 0; q < IF; ++q) {
                if (tg[q] >= 0) lds_add(P + tg[q], v[q]);          // (a piece has at most 64 rows; pieces in item order)
            }
        }
    }
}

// lane K of each 16-lane row to every lane of that row (DPP row_newbcast, gfx90a+; two 32-bit moves: the 64-bit DPP
// forms exist on gfx950 but were measured at ~1/16 rate).  K must be a compile-time constant after unrolling.
#define HIPKKT_BCAST16_CASE(K) case K: lo = __builtin_amdgcn_update_dpp(0, lo, 0x150 + K, 0xf, 0xf, false); \
                                       hi = __builtin_amdgcn_update_dpp(0, hi, 0x150 + K, 0xf, 0xf, false); break;
__device__ __forceinline__ double bcast16(double v, const int k)
{
    int lo = __double2loint(v), hi = __double2hiint(v);
    switch (k) {
        HIPKKT_BCAST16_CASE(0) HIPKKT_BCAST16_CASE(1) HIPKKT_BCAST16_CASE(2) HIPKKT_BCAST16_CASE(3)
        HIPKKT_BCAST16_CASE(4) HIPKKT_BCAST16_CASE(5) HIPKKT_BCAST16_CASE(6) HIPKKT_BCAST16_CASE(7)
        HIPKKT_BCAST16_CASE(8) HIPKKT_BCAST16_CASE(9) HIPKKT_BCAST16_CASE(10) HIPKKT_BCAST16_CASE(11)
        HIPKKT_BCAST16_CASE(12) HIPKKT_BCAST16_CASE(13) HIPKKT_BCAST16_CASE(14) HIPKKT_BCAST16_CASE(15)
    }
    return __hiloint2double(hi, lo);
}

// SLICED: a panel too tall for one CU's LDS is cut into ROW slices, one workgroup each (TreeDev::sdesc).  Every slice
// holds the top nc x nc block plus its share of the rows below and factors the top block itself, so the slices
// never talk to each other: the diagonal blocks are computed redundantly (they sit on every slice's critical path
// anyway), the substitution and the trailing update only ever combine a row with the top block.  Inside the kernel
// a slice is simply a front with f = nc + (rows of the slice); only assembly and the final store map rows.
template <int BS, bool SLICED, bool OV>
__global__ __launch_bounds__(BS) void k_panel(FactorArgs A, int begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NW = BS / 64;
    constexpr int NWK = NW - NW / 4;                               // worker waves of the block loop (not on wave 0's SIMD)
    const int widx = wv - 1 - (wv >> 2);                           // this wave's index among them (wv % 4 != 0)
    const TreeDev& T = A.T;
    const FrontDesc fd = SLICED ? T.sdesc[begin + blockIdx.x] : T.desc[begin + blockIdx.x];
    const int s = fd.s, c0 = fd.c0, nc = fd.nc, nb = fd.nb;
    const int ff = nc + nb;                                        // rows of the whole front
    const int nsl = SLICED ? (fd.pad & 0xffff) : 1, sl = SLICED ? (fd.pad >> 16) : 0;
    const int rsmax = (nb + nsl - 1) / nsl;
    const int r_lo = nc + sl * rsmax;                              // front row of this slice's first row below the top
    const int rs = SLICED ? max(0, min(rsmax, ff - r_lo)) : nb;
    const int f = nc + rs;                                         // rows held here
    const bool first = !SLICED || sl == 0;                         // writes what all slices compute alike
    double* __restrict__ F = A.fronts + fd.front_off;
    // overlap mode: this workgroup is resident now (FactorArgs::ov_started; the launch's tiles wait for all of them)
    if (OV && tid == 0) __hip_atomic_fetch_add(A.ov_started + A.ov_slot, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

    // block data is double-buffered by block parity: wave 0 factors diagonal block k+1 while the other waves
    // still read block k's d, 1/d and d*L
    double* sh_d_all = smem;                      // 2 x NB
    double* sh_dinv_all = smem + 2 * NB;          // 2 x NB
    double* Lb_all = smem + 4 * NB;               // 2 x NB x NB: Lb[j*NB + t] = d_t * L(j,t), t < j
    double* colb = smem + 4 * NB + 2 * NB * NB;   // 4*NB (spare)
    double* sgn = colb + 4 * NB;                  // kBdCols + NB: expected pivot signs of this front's columns
    int* lds_cnt = reinterpret_cast<int*>(sgn + kBdCols + NB);   // (2 doubles) arrival counter of the in-block barrier
    double* P = sgn + kBdCols + NB + 2;           // f x nc, ld f (+ 256 doubles of slack behind it)

    HIPKKT_STAMP(A, 0);
    const long long clk0 = A.stamps ? clock64() : 0;
    // (the dense child's offset: a scalar load issued now -- fetched where it is used, behind the waits and barriers, it was
    //  one more memory round trip on every panel workgroup's path: +1.8 us per launch of cfg2's small panels)
    const int64_t dense_off_s = T.dense_off[s];
    // this wave's slice of the children's extend-add items (host-cut, 16 slices) and its first 64 descriptors: issued
    // now, used in phase 3
    int64_t it0 = 0, it1 = 0;
    ExtItem it_first{};
    {
        const int64_t* __restrict__ wc = T.wave_cut + (int64_t)(SLICED ? T.nsuper + begin + (int)blockIdx.x : s) * 17;
        constexpr int SPW = 16 / NW > 0 ? 16 / NW : 1;      // slices per wave
        if (wv * SPW < 16) {
            it0 = wc[wv * SPW];
            it1 = wc[min(16, (wv + 1) * SPW)];
            if (it1 > it0) it_first = T.items[min(it0 + lane, it1 - 1)];
        }
    }
    // ---- 1. zero the panel (and fetch the pivot signs: the serial diagonal step must not wait for global memory)
    for (int k = tid; k < nc; k += BS) sgn[k] = (double)T.psign[c0 + k];
    const int psize = f * nc - ((nc * (nc - 1)) >> 1);
    for (int i = tid; i < psize; i += BS) P[i] = 0.0;
    __syncthreads();
    HIPKKT_STAMP(A, 1);
    // ---- 2. scatter K, four entries per thread in flight
    {
        const int64_t e0 = fd.kptr;
        const int ne = fd.nk;
        for (int base = 0; base < ne; base += 4 * BS) {
            int src[4], dst[4];
            double v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = base + q * BS + tid;
                src[q] = e < ne ? T.ksrc[e0 + e] : -1;
                dst[q] = e < ne ? T.kdst[e0 + e] : -1;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = src[q] >= 0 ? A.Kval[src[q]] : 0.0;
            // (a row slice has a K list of its own with positions in its LDS image: hipkkt.hip, upload)
#pragma unroll
            for (int q = 0; q < 4; ++q) if (dst[q] >= 0) P[dst[q]] = v[q];
        }
    }
    __syncthreads();
    if (A.eps) {
        const double eps = *A.eps;
        for (int k = tid; k < nc; k += BS) P[k + pcol(k, f)] += eps * sgn[k];
    }
    __syncthreads();
    HIPKKT_STAMP(A, 2);
    if (OV) {
        // the children's Schur tiles run beside this kernel: wait until every tile of every child has been stored
        volatile int& sh_ov_ok = reinterpret_cast<volatile int*>(lds_cnt)[2];
        if (tid == 0) sh_ov_ok = 1;
        __syncthreads();
        if (wv == 0) {
            const long long tw = wall_clock64();
            bool ok = true;
            for (int e = T.child_ptr[s] + lane; e < T.child_ptr[s + 1]; e += 64) {
                const int c = T.child_idx[e];
                const int need = A.ov_ntiles[c];
                if (need > 0) ok = ov_wait_ge(A.ov_done + c, need, A.flags + 2, tw, A.ov_limit, 1, c) && ok;
            }
            if (!ok) sh_ov_ok = 0;
        }
        __syncthreads();
        if (!sh_ov_ok) return;
    }
    // ---- 3a. the dense child, if any (TreeDev::dense_off): entry (r, j) of its update block is entry (r, j) of this
    //          front -- column by column, coalesced, eight loads per thread in flight; every thread owns its entries, the
    //          barrier separates them from the waves' item adds below
    {
        const int64_t doff = dense_off_s;
        if (doff >= 0) {
            const double* __restrict__ Uc = A.upd + doff;
            const int total = f * nc;                                  // (rectangle walked, entries above the diagonal skipped)
            for (int base = 0; base < total; base += 8 * BS) {
                double v[8];
                int tg[8];
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int idx = base + q * BS + tid;
                    const int j = idx / f, rl = idx - j * f;
                    const int r = (!SLICED || rl < nc) ? rl : r_lo + (rl - nc);       // row of the whole front
                    const bool ok = idx < total && rl >= j;
                    tg[q] = ok ? rl + pcol(j, f) : -1;
                    v[q] = ok ? (OV ? OV_LD(Uc + r + (int64_t)j * ff) : Uc[r + (int64_t)j * ff]) : 0.0;
                }
#pragma unroll
                for (int q = 0; q < 8; ++q) if (tg[q] >= 0) P[tg[q]] += v[q];
            }
            __syncthreads();
        }
    }
    // ---- 3. children: each wave applies its slice of whole columns' items
    if (A.stamps && blockIdx.x == 0 && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        A.stamps[A.stamp_row * 16 + 6] = wall_clock64();
    }
    if (it1 > it0) apply_items_panel<SLICED, OV>(T, A.upd, P, f, it0, it1, lane, nc, r_lo, rs, it_first,
                                             (A.stamps && blockIdx.x == 0 && wv == 0) ? A.stamps + A.stamp_row * 16 + 15 : nullptr);
    if (A.stamps && blockIdx.x == 0 && tid == 0) A.stamps[A.stamp_row * 16 + 7] = wall_clock64();
    HIPKKT_STAMP(A, 3);
    long long t_i = 0, t_ii = 0, t_iii = 0, t0 = 0;
    // The 16 x 16 diagonal block kb as a callable: with look-ahead it runs on wave 0 while the other waves finish the
    // trailing update of the previous block.  ROW PER LANE: lane i (< 16) keeps row i of the block in 16 registers and,
    // beside it, row i of T = L_bb^{-1} (unit lower), built by the same row operations.  Pivot by pivot, fully
    // unrolled: the pivot, the entries a(j, k) of the pivot column and the entries T(k, c) of T's pivot row come out of
    // their owners' registers as scalar broadcasts (v_readlane) -- no LDS round trip anywhere on the pivot chain, and
    // the reciprocal of the next pivot can start as soon as its entry is updated.  Same operation order per entry as
    // the scalar algorithm; QDLDL's sign rule at pivot time.  T lets every wave turn its rows' substitution against
    // the block into one 16 x 16 x 16 matrix product (trsm_tile below).
    auto diag_block = [&](const int kb, const int w, const int par) {
        double* sh_d = sh_d_all + par * NB;
        double* sh_dinv = sh_dinv_all + par * NB;
        double* Tb = Lb_all + par * NB * NB;         // Tb[i * NB + c] = T(i, c), T = L_bb^{-1}
        // Lane (i = lane & 15, g = lane >> 4) holds a(i, 4g .. 4g+3) and, beside it, T(i, 4g .. 4g+3): T = L_bb^{-1}
        // (unit lower) is built by the same row operations, T(i, :) -= l_i T(k, :), whose pivot row comes from lane
        // (k, g) of the SAME 16-lane row by a DPP row broadcast (row_newbcast, gfx90a+) -- no LDS, no scalar registers.
        // T turns every wave's substitution of its rows against the block into one 16 x 16 x 16 product (trsm_tile).
        // Pivot by pivot, fully unrolled: every lane puts its entry of its group's pivot-column candidate into LDS
        // (slot g * 16 + i: no lane mask), every lane reads back its row's entry and the entries of its own four
        // columns' rows from the pivot column's group -- one LDS round trip per pivot -- while the pivot comes straight
        // out of its owner's register (scalar broadcast) and its reciprocal is formed beside the trip.  This wave is
        // alone on the serial chain and bound by its own instruction issue, so the step is written for few
        // instructions: no operation is guarded by "column already finished" (finished columns of a4 / zero parts of T
        // are simply never read again), and the lane id is made opaque per step so that the compiler computes the two
        // lane conditions where they are used instead of keeping 16 x 3 precomputed masks alive in (spilled) SGPRs.
        // Same operation order per live entry as the scalar algorithm; QDLDL's sign rule at pivot time.
        int lane_o = lane;
        const int i = lane & 15, g = lane >> 4;
        const double my_sg = (lane < w) ? sgn[kb + lane] : 1.0;      // lane k: expected sign of pivot k
        int nreg = 0;
        bool bad = false;
        double a4[4], lout[4], t4[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * g + q;
            a4[q] = (i < w && j <= i) ? P[(kb + i) + pcol(kb + j, f)] : 0.0;
            lout[q] = 0.0;
            t4[q] = (j == i) ? 1.0 : 0.0;
        }
        const double inv_delta = 1.0 / A.dyn_delta;
#pragma unroll
        for (int k = 0; k < NB; ++k) {
            if (k < w) {
                asm volatile("" : "+v"(lane_o));
                const int io = lane_o & 15, go = lane_o >> 4;
                const int gk = k >> 2, qk = k & 3;
                colb[lane_o] = a4[qk];                                // (one wave: its LDS operations execute in order)
                WAVE_FENCE();
                const double ci = colb[gk * NB + io];
                double cj[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) cj[t] = colb[gk * NB + 4 * go + t];
                WAVE_FENCE();
                // the pivot comes straight out of its owner's register (scalar broadcast), so its reciprocal is
                // under way while the column is still on its way through LDS
                double d = rl_f64(a4[qk], k + 16 * gk);
                const double sg = rl_f64(my_sg, k);
                const bool reg = (d * sg < A.dyn_eps);               // QDLDL's sign rule at pivot time
                // the reciprocal starts from the raw pivot; the regularised case has a constant one
                double di = fast_recip(d);
                if (reg) { d = sg * A.dyn_delta; di = sg * inv_delta; }
                nreg += reg ? 1 : 0;
                bad = bad || !isfinite(di);
                const double li = ci * di;
#pragma unroll
                for (int t = 0; t < 4; ++t) a4[t] = fma(-li, cj[t], a4[t]);     // (finished columns: garbage, never read)
                // rows below the pivot of T: T(i, c) -= l_i T(k, c)  (c > k: T(k, c) = 0, the operation adds nothing)
                const double lt = (io > k) ? -li : 0.0;
#pragma unroll
                for (int t = 0; t < 4; ++t) t4[t] = fma(lt, bcast16(t4[t], k), t4[t]);
                // the owner of (row i, column k) keeps L(i, k) (below the pivot) or d (the pivot itself)
                if (go == gk) lout[qk] = (io > k) ? li : d;
                sh_d[k] = d;                                          // (every lane, same value)
                sh_dinv[k] = di;
            }
        }
        for (int k = w + lane; k < NB; k += 64) { sh_d[k] = 1.0; sh_dinv[k] = 1.0; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int j = 4 * g + q;
            if (i < w && j <= i) P[(kb + i) + pcol(kb + j, f)] = lout[q];       // scaled L below, d on the diagonal
            Tb[i * NB + j] = t4[q];
        }
        WAVE_FENCE();
        if (first && lane < w) A.Dinv[c0 + kb + lane] = sh_dinv[lane];
        if (first && lane == 0) {
            if (nreg) atomicAdd(&A.flags[0], nreg);
            if (bad) A.flags[1] = 1;
        }
    };
    // ---- 4. blocked right-looking factorisation inside LDS, software-pipelined across the waves:
    //   wave 0        rows of the NEXT diagonal block against block k (16 rows), the one 16 x 16 tile that updates
    //                 that diagonal block, then its factorisation -- the serial chain of the panel, start to end;
    //   waves 1..     the other rows against block k (thread per row), an LDS arrival counter as their barrier
    //                 (wave 0 signals but does not wait), then the rest of the trailing update on the matrix cores.
    //   One workgroup barrier per block.  The B operand of the trailing tiles is d_k L(j,k), scaled on the fly.
    if (tid == 0) { lds_cnt[0] = 0; lds_cnt[1] = 0; }
    __syncthreads();                     // the panel is assembled (every wave has applied its children's columns)
    // (the loop starts one block early: that prologue pass only factors diagonal block 0 -- ONE copy of the diagonal
    // step's long straight-line code in the kernel instead of two)
    int epoch = 0;
    for (int kb = -NB; kb < nc; kb += NB) {
        const bool pro = kb < 0;
        const int w = pro ? 0 : min(NB, nc - kb);
        const int par = pro ? 1 : ((kb / NB) & 1);
        const double* sh_d = sh_d_all + par * NB;
        const double* sh_dinv = sh_dinv_all + par * NB;
        const double* Lb = Lb_all + par * NB * NB;
        __syncthreads();                 // diagonal block kb is factored, trailing update kb - NB is complete
        if (A.stamps) t0 = wall_clock64();
        if (!pro) ++epoch;
        const int g0 = pro ? 0 : kb + w;
        const int Tc = nc - g0, Tr = f - g0;
        const int ml = lane & 15, mk = lane >> 4;
        // rows below the block, 16 at a time: L(rows, block) = A(rows, block) T' D^{-1} with T = L_bb^{-1} -- one
        // 16 x 16 x 16 product on the matrix cores instead of a 120-term dependent substitution chain per row.
        // A operand A[i = ml][k] = a(row0 + i, kb + k), B operand B[k][j = ml] = T(j, k).
        auto trsm_tile = [&](const int row0) {
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 4 * kk + mk;
                av[kk] = (row0 + ml < f && k < w) ? P[(row0 + ml) + pcol(kb + k, f)] : 0.0;
                bv[kk] = Lb[ml * NB + k];
            }
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
            const double dj = sh_dinv[ml];
            const int ccol = pcol(kb + min(ml, w - 1), f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + mk + 4 * r;
                if (row < f && ml < w) P[row + ccol] = acc[r] * dj;
            }
        };
        // C(16 x 16 tile at rows g0 + 16 tr, columns g0 + 16 tc) -= L(rows, block) * (d L(cols, block))'
        auto tile = [&](const int tr, const int tc) {
            const int i0 = g0 + 16 * tr, j0 = g0 + 16 * tc;
            double av[4], bv[4];
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) {
                const int k = 4 * kk + mk;
                const int cbk = pcol(kb + k, f);
                av[kk] = (i0 + ml < f && k < w) ? P[(i0 + ml) + cbk] : 0.0;
                bv[kk] = (j0 + ml < nc && k < w) ? P[(j0 + ml) + cbk] * sh_d[k] : 0.0;
            }
            d4_t acc = (d4_t){0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[kk], bv[kk], acc, 0, 0, 0);
            const int ccol = pcol(min(j0 + ml, nc - 1), f);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = i0 + mk + 4 * r, col = j0 + ml;
                if (row < f && col < nc && row >= col) P[row + ccol] -= acc[r];
            }
        };
        const int nfirst = pro ? 0 : min(NB, Tr);             // rows of the next diagonal block (or the last rows)
        if (wv == 0) {
            if (!pro) {
                long long ta = A.stamps ? wall_clock64() : 0;
                if (nfirst > 0) trsm_tile(g0);
                WAVE_FENCE();
                if (Tc > 0) {
                    tile(0, 0);
                    WAVE_FENCE();
                }
                if (lane == 0) __hip_atomic_fetch_add(lds_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (A.stamps) t_ii += wall_clock64() - ta;
            }
            if (Tc > 0) {
                long long ta = A.stamps ? wall_clock64() : 0;
                diag_block(g0, min(NB, nc - g0), par ^ 1);
                if (A.stamps) t_i += wall_clock64() - ta;
            }
        } else if (!pro && (wv & 3) != 0) {
            // Waves are dealt round-robin to the CU's four SIMDs, so waves 4, 8, 12 share wave 0's.  Wave 0 carries
            // the serial chain and is bound by its own instruction issue (~55 instructions per pivot): those waves
            // sit this part out at the barrier instead of competing for its issue slots (workers: the other 3/4).
            for (int row0 = g0 + nfirst + 16 * widx; row0 < f; row0 += 16 * NWK) trsm_tile(row0);
            WAVE_FENCE();
            if (lane == 0) __hip_atomic_fetch_add(lds_cnt, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (OV) {
                // Block kb's columns are final once every wave's rows are in place: the workers publish them now
                // (written through), off wave 0's chain, and the last one to finish advances the front's progress
                // counter -- the front's Schur tiles, running beside this kernel, take the block from there.
                while (__hip_atomic_load(lds_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (NWK + 1) * epoch)
                    __builtin_amdgcn_s_sleep(1);
                WAVE_FENCE();
                for (int idx = widx * 64 + lane; idx < w * f; idx += NWK * 64) {
                    const int jj = idx / f, r = idx - jj * f, j = kb + jj;
                    if (!SLICED) {
                        if (r >= j) OV_ST(F + r + (int64_t)j * ff, P[r + pcol(j, f)]);
                    } else if (r >= nc) {                   // (a slice: its own rows below the top block ...
                        OV_ST(F + (r_lo + r - nc) + (int64_t)j * ff, P[r + pcol(j, f)]);
                    } else if (first && r >= j) {           //  ... and the top block from the first slice)
                        OV_ST(F + r + (int64_t)j * ff, P[r + pcol(j, f)]);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if (lane == 0) {
                    const int arrived = __hip_atomic_fetch_add(lds_cnt + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) + 1;
                    if (arrived == NWK * epoch)
                        __hip_atomic_store(SLICED ? A.ov_sprog + begin + blockIdx.x : A.ov_prog + s, kb + w, __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            if (Tc > 0) {
                // every wave's rows are in place once all NW arrivals of this block are counted
                while (__hip_atomic_load(lds_cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < (NWK + 1) * epoch)
                    __builtin_amdgcn_s_sleep(1);
                WAVE_FENCE();
                const int ntr = (Tr + 15) >> 4, ntc = (Tc + 15) >> 4;
                for (int t = widx; t < ntr * ntc; t += NWK) {
                    const int tc = t / ntr, tr = t - tc * ntr;
                    if (tr < tc || (tr == 0 && tc == 0)) continue;      // above the diagonal / wave 0's tile
                    tile(tr, tc);
                }
            }
        }
        if (A.stamps) { long long t1 = wall_clock64(); t_iii += t1 - t0; }
    }
    __syncthreads();
    HIPKKT_STAMP(A, 4);
    // ---- 5. write L and D (lower part of the panel; overlap mode has published every block already)
    if (!OV) for (int idx = tid; idx < f * nc; idx += BS) {
        const int j = idx / f, r = idx - j * f;
        if (!SLICED) {
            if (r >= j) F[idx] = P[r + pcol(j, f)];
        } else if (r >= nc) {
            F[(r_lo + r - nc) + (int64_t)j * ff] = P[r + pcol(j, f)];
        } else if (first && r >= j) {
            F[r + (int64_t)j * ff] = P[r + pcol(j, f)];
        }
    }
    __syncthreads();
    HIPKKT_STAMP(A, 5);
    if (A.stamps && blockIdx.x == 0 && tid == 0) {
        A.stamps[A.stamp_row * 16 + 8] = t_i;
        A.stamps[A.stamp_row * 16 + 9] = t_ii;
        A.stamps[A.stamp_row * 16 + 10] = t_iii;
        A.stamps[A.stamp_row * 16 + 11] = f;
        A.stamps[A.stamp_row * 16 + 12] = nc;
        A.stamps[A.stamp_row * 16 + 13] = T.child_ptr[s + 1] - T.child_ptr[s];
        A.stamps[A.stamp_row * 16 + 14] = clock64() - clk0;
    }
}

// =====================================================================================
//  Schur complement: U(TSxTS tile) = sum over children of their pass-through entries
//                                    - L21 D L21^T,   one workgroup per tile
// =====================================================================================
constexpr int TS = 64;     // tile side
constexpr int KC = 16;     // k-chunk staged in LDS

constexpr int LDA = TS + 16;    // k-rows 80 doubles apart: consecutive k land 32 banks apart (conflict-free MFMA operand reads)

// The rank-nc update runs on the matrix cores: v_mfma_f64_16x16x4_f64, each of the 4 waves owns a
// 32 x 32 quadrant of the tile (2 x 2 MFMA tiles, 4 k-steps per staged chunk).  Operand layout
// (cdna_hip_programming.md section 3): A[i = lane&15][k = lane>>4], B[k = lane>>4][j = lane&15],
// D: col = lane&15, row = (lane>>4) + 4*reg.
// OV (overlap mode): the tile runs BESIDE its front's panel kernel.  The children's pass-through (which does not depend
// on this front's panel at all) is summed into the tile buffer first; then the product takes the panel's 16-column
// blocks one by one as the panel kernel publishes them (FactorArgs::ov_prog), so that when the panel's last block
// arrives only one rank-16 update and the store remain; the tile is stored written-through and counted in ov_done.
// (The kernel can walk its tiles on a grid smaller than their number -- workgroup b takes tiles b, b + G, ... -- which
// is how a bounded tile grid was tried for the overlap mode's forward-progress guarantee; measured +0.1 ms on cfg2's
// factorisation.  The guarantee comes from the gate below instead and the grid is the number of tiles.)
// PIPE: the chunk loop's operand loads run one chunk ahead (always in overlap mode; without it only in launches of
// thousands of tiles, launch_schur below).
template <bool OV, bool PIPE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(OV ? 3 : 4))) void k_schur(FactorArgs A, const int2* __restrict__ tiles, int tile_begin, int ntiles)
{
    // the operand chunks are dead once the product is done: the tile buffer shares their LDS (33 KB per
    // workgroup instead of 53 KB -> one more workgroup per CU)
    // (overlap mode keeps the tile buffer beside the operand chunks: it is filled before the product)
    __shared__ __attribute__((aligned(16))) double smem_s[TS * (TS + 1) + (OV ? 2 * KC * LDA + 2 : 0)];
    double (*As)[LDA] = reinterpret_cast<double (*)[LDA]>(smem_s + (OV ? TS * (TS + 1) : 0));   // As[k][r] = L21(r0 + r, k0 + k)
    double (*Bs)[LDA] = reinterpret_cast<double (*)[LDA]>(smem_s + (OV ? TS * (TS + 1) : 0) + KC * LDA);   // Bs[k][c] = L21(q0 + c, k0 + k) * d_k
    double (*Ct)[TS + 1] = reinterpret_cast<double (*)[TS + 1]>(smem_s);           // the tile, [col][row]
    volatile int* sh_ok = reinterpret_cast<volatile int*>(smem_s + TS * (TS + 1) + 2 * KC * LDA);   // (OV only)
    static_assert(2 * KC * LDA <= TS * (TS + 1), "operand chunks must fit under the tile buffer");
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const TreeDev& T = A.T;
    for (int tix = blockIdx.x; tix < ntiles; tix += gridDim.x) {
    const int2 tl = tiles[tile_begin + tix];
    const int s = tl.x;
    const int ti = tl.y >> 16, tj = tl.y & 0xffff;
    const int c0 = T.sn_start[s];
    const int nc = T.sn_start[s + 1] - c0;
    const int nb = (int)(T.rowptr[s + 1] - T.rowptr[s]);
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + T.front_off[s];
    double* __restrict__ U = A.upd + T.upd_off[s];
    const int r0 = ti * TS, q0 = tj * TS;                 // tile origin inside U
    const int nr = min(TS, nb - r0), nq = min(TS, nb - q0);

    const int wr = wv >> 1, wc = wv & 1;                  // this wave's 32 x 32 quadrant
    // quadrants strictly above the diagonal of a diagonal tile, or wholly outside the front, are idle
    const bool active = !(ti == tj && wr < wc) && 32 * wr < nr && 32 * wc < nq;
    d4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (d4_t){0.0, 0.0, 0.0, 0.0};
    const int ml = lane & 15, mk = lane >> 4;
    // the pass-through work list is static data: fetch this wave's range and its first 64 descriptors now -- lane l
    // takes descriptor l, ONE vector load round -- so that after the product only (rel, value) load rounds remain
    const int64_t* __restrict__ tc = T.tile_cut + 5 * (int64_t)(tile_begin + tix);
    const int64_t i0 = tc[wv], i1 = tc[wv + 1];
    static_assert(sizeof(SubItem) == 16, "SubItem is read as an int4");
    const int4* __restrict__ sit4 = reinterpret_cast<const int4*>(T.sitems);
    int4 mine = (i0 < i1) ? sit4[min(i0 + lane, i1 - 1)] : make_int4(0, 0, 0, 0);
    // children pass-through: this wave's slice of the tile's sub-items (whole columns, child order),
    // PT in flight: the descriptors handed out with v_readlane, one round of (rel, value) loads, then LDS adds
    auto pass_through = [&]() {
        constexpr int PT = 16;
        const int rlo = nc + r0;
        double* ct = &Ct[0][0];
        for (int64_t base = i0; base < i1; base += 64) {
            if (base > i0) mine = sit4[min(base + lane, i1 - 1)];
            const int n = (int)min((int64_t)64, i1 - base);
            for (int z0 = 0; z0 < n; z0 += PT) {
                double v[PT];
                int tg[PT];
#pragma unroll
                for (int z = 0; z < PT; ++z) {
                    const int src = min(z0 + z, 63);
                    const unsigned lo = (unsigned)__builtin_amdgcn_readlane(mine.x, src);
                    const int hi = __builtin_amdgcn_readlane(mine.y, src);
                    const int relstart = __builtin_amdgcn_readlane(mine.z, src);
                    const int cq = __builtin_amdgcn_readlane(mine.w, src);          // cnt | qcol << 8
                    const int64_t uoff = ((int64_t)hi << 32) | lo;
                    const int cnt = (z0 + z < n) ? (cq & 0xff) : 0, qcol = (cq >> 8) & 0xff;
                    const bool ok = lane < cnt;
                    v[z] = ok ? A.upd[uoff + lane] : 0.0;
                    tg[z] = ok ? qcol * (TS + 1) + (T.rel[relstart + lane] - rlo) : -1;
                }
#pragma unroll
                for (int z = 0; z < PT; ++z) if (tg[z] >= 0) lds_add(ct + tg[z], v[z]);
            }
        }
    };
    // the dense child (TreeDev::dense_off): entry (i, j) of this front's update block receives entry (nc + i, nc + j)
    // of the child's (leading dimension f) -- a plain block copy, no lists
    const int64_t dense_off_s = T.dense_off[s];           // (scalar, fetched with the other descriptors)
    auto dense_ptr = [&]() -> const double* {
        return dense_off_s >= 0 ? A.upd + dense_off_s + (int64_t)(nc + r0) + (int64_t)(nc + q0) * f : nullptr;
    };
    if (OV) {
        const double* __restrict__ Ud = dense_ptr();
        auto dense_at = [&](int a, int b) -> double {
            return (Ud && a < nr && b < nq && (r0 + a) >= (q0 + b)) ? Ud[a + (int64_t)b * f] : 0.0;
        };
        for (int idx = tid; idx < TS * (TS + 1); idx += 256) {
            const int b = idx / (TS + 1), a = idx - b * (TS + 1);
            (&Ct[0][0])[idx] = dense_at(a < TS ? a : nr, b);
        }
        if (tid == 0) *sh_ok = 1;
        __syncthreads();
        pass_through();
    }
    const long long tw = OV ? wall_clock64() : 0;
    int known = 0;                 // (thread 0) columns of the panel known to be published by every workgroup this tile reads from
    // Overlap mode: the panel kernel of this front publishes its 16-column blocks as they are finished; thread 0 makes sure
    // that columns [k0, k0 + kw) are there (the workgroup learns the outcome behind its next barrier).  What a poll saw is
    // remembered: on a tall front most tiles start long after their panel has finished (a 14 000-row level has 24 000
    // tiles), and every chunk used to pay three to five dependent polls again -- 20-30 us of a 35 us tile.
    auto ensure = [&](int k0, int kw) {
        if (tid != 0 || k0 + kw <= known) return;
        bool ok = true;
        int seen = 1 << 30;
        const int sb = A.ov_sbase[s];
        if (sb < 0) {
            ok = ov_wait_ge(A.ov_prog + s, k0 + kw, A.flags + 2, tw, A.ov_limit, 2, s, &seen);
        } else {
            // a front factorised in row slices: the slices that hold this tile's two strips, and the first one
            // (it publishes the top block, whose diagonal the product scales with)
            const FrontDesc sd = T.sdesc[sb];
            const int nsl = sd.pad & 0xffff, rsmax = (nb + nsl - 1) / nsl;
            ok = ov_wait_ge(A.ov_sprog + sb, k0 + kw, A.flags + 2, tw, A.ov_limit, 0, 0, &seen);
            for (int q = r0 / rsmax; ok && q <= (r0 + nr - 1) / rsmax; ++q)
                ok = ov_wait_ge(A.ov_sprog + sb + q, k0 + kw, A.flags + 2, tw, A.ov_limit, 0, 0, &seen);
            for (int q = q0 / rsmax; ok && q <= (q0 + nq - 1) / rsmax; ++q)
                ok = ov_wait_ge(A.ov_sprog + sb + q, k0 + kw, A.flags + 2, tw, A.ov_limit, 0, 0, &seen);
        }
        if (!ok) *sh_ok = 0;
        else known = seen;
    };
    // this thread's share of a chunk's operands: row k of the chunk, four entries of each strip, the pivot d_k
    const int ck = tid >> 4, crr = (tid & 15) * 4;
    double av[4], bv[4], dk = 0.0;
    // (overlap mode: the panel entries were written through by a kernel that is still running)
    auto ldF = [&](int64_t o) { return OV ? OV_LD(F + o) : F[o]; };
    auto fetch = [&](int k0, int kw) {
        dk = (ck < kw) ? ldF((k0 + ck) + (int64_t)(k0 + ck) * f) : 0.0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            av[q] = (ck < kw && crr + q < nr) ? ldF((nc + r0 + crr + q) + (int64_t)(k0 + ck) * f) : 0.0;
            bv[q] = (ti != tj && ck < kw && crr + q < nq) ? ldF((nc + q0 + crr + q) + (int64_t)(k0 + ck) * f) : 0.0;
        }
    };
    // The chunk loop is software-pipelined through registers: chunk k + 1's operands are on their way from memory while
    // chunk k's products run (a tile's 6 chunks used to be 6 x (load latency + two barriers)).
    // (Overlap mode only: without it -- the wide middle levels of cfg2 -- the 18 registers the operands in flight cost the
    //  kernel were measured as +10 us per factorisation, 1.744 -> 1.755 ms; there the loads stay in front of the barrier.)
    if (OV && nc > 0) {
        ensure(0, min(KC, nc));
        __syncthreads();
        if (!*sh_ok) return;
    }
    if (PIPE && nc > 0) fetch(0, min(KC, nc));
    for (int k0 = 0; k0 < nc; k0 += KC) {
        const bool more = k0 + KC < nc;
        const int kwn = more ? min(KC, nc - k0 - KC) : 0;
        if (!PIPE) fetch(k0, min(KC, nc - k0));
        __syncthreads();           // previous chunk fully consumed
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            As[ck][crr + q] = av[q];
            Bs[ck][crr + q] = ((ti == tj) ? av[q] : bv[q]) * dk;          // (diagonal tile: both strips are the same rows)
        }
        if (OV && more) ensure(k0 + KC, kwn);
        __syncthreads();
        if (OV && !*sh_ok) return;
        if (PIPE && more) fetch(k0 + KC, kwn);
        if (active) {
#pragma unroll
            for (int kk = 0; kk < KC; kk += 4) {
                double a[2], b[2];
#pragma unroll
                for (int i = 0; i < 2; ++i) a[i] = As[kk + mk][32 * wr + 16 * i + ml];
#pragma unroll
                for (int j = 0; j < 2; ++j) b[j] = Bs[kk + mk][32 * wc + 16 * j + ml];
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
            }
        }
    }
    __syncthreads();               // every wave is done reading the last operand chunk (Ct overlays it)
    // quadrants that took no part hold zeros in acc: the whole tile buffer gets defined here
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (OV) Ct[32 * wc + 16 * j + ml][32 * wr + 16 * i + mk + 4 * r] -= acc[i][j][r];      // (each entry has one owner)
                else Ct[32 * wc + 16 * j + ml][32 * wr + 16 * i + mk + 4 * r] = -acc[i][j][r];
    __syncthreads();
    if (!OV) pass_through();
    __syncthreads();
    // store the lower part of the tile
    const double* __restrict__ Ud = OV ? nullptr : dense_ptr();
    if (!OV && Ud) {
        auto dense_at = [&](int a, int b) -> double {
            return (a < nr && b < nq && (r0 + a) >= (q0 + b)) ? Ud[a + (int64_t)b * f] : 0.0;
        };
        // (non-overlap launches add the dense child's entries here, four loads per thread in flight: fetched earlier
        // and held across the product or the lists they cost the kernel a quarter of its occupancy)
#pragma unroll 1
        for (int base = 0; base < TS * TS; base += 4 * 256) {
            double dv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { const int idx = base + q * 256 + tid; dv[q] = dense_at(idx & 63, idx >> 6); }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int idx = base + q * 256 + tid;
                const int b = idx >> 6, a = idx & 63;
                if (a < nr && b < nq && (r0 + a) >= (q0 + b)) U[(r0 + a) + (int64_t)(q0 + b) * nb] = Ct[b][a] + dv[q];
            }
        }
    } else {
        for (int idx = tid; idx < TS * TS; idx += 256) {
            const int b = idx >> 6, a = idx & 63;
            if (a < nr && b < nq && (r0 + a) >= (q0 + b)) {
                if (OV) OV_ST(U + (r0 + a) + (int64_t)(q0 + b) * nb, Ct[b][a]);
                else U[(r0 + a) + (int64_t)(q0 + b) * nb] = Ct[b][a];
            }
        }
    }
    if (OV) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // every storing wave drains
        __syncthreads();
        if (tid == 0) __hip_atomic_fetch_add(A.ov_done + s, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    }   // tiles of this workgroup
}

// Overlap mode's gate: one wave, no LDS, on the tile stream in front of a launch's tile kernel.  It ends when every panel
// workgroup of the launch has started, i.e. IS RESIDENT (a workgroup keeps its CU until it ends).  From then on the
// launch's tiles may fill every other CU: each of them waits only for a panel workgroup that is running, and a running
// panel workgroup waits only for the previous launch's tiles, which were released the same way -- so every wait ends,
// whatever the hardware's dispatch order (forward progress by induction over the launches, not by submission order).
__global__ __launch_bounds__(64) void k_ov_gate(const int* __restrict__ started, int target, int* abort_word, long long limit)
{
    if (threadIdx.x == 0) (void)ov_wait_ge(started, target, abort_word, wall_clock64(), limit, 3, target);
}
void launch_ov_gate(const int* started, int target, int* abort_word, long long limit, hipStream_t st)
{
    hipLaunchKernelGGL(k_ov_gate, dim3(1), dim3(64), 0, st, started, target, abort_word, limit);
}

// Do two streams really run side by side?  HIP multiplexes a process's streams onto a few hardware queues; two streams
// that share one execute their kernels in submission order.  k_probe_wait (first stream) spins until k_probe_set (second
// stream, submitted AFTER it) has stored the word, or a bound of ~2 ms passes: word[1] = 1 if it saw the store.
__global__ __launch_bounds__(64) void k_probe_wait(int* word)
{
    if (threadIdx.x != 0) return;
    const long long t0 = wall_clock64();
    int seen = 0;
    while (wall_clock64() - t0 < 200000) {               // 2 ms at 100 MHz
        if (__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) { seen = 1; break; }
        __builtin_amdgcn_s_sleep(8);
    }
    word[1] = seen;
}
__global__ __launch_bounds__(64) void k_probe_set(int* word)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
void launch_concurrency_probe(int* word2, hipStream_t first, hipStream_t second)
{
    hipLaunchKernelGGL(k_probe_wait, dim3(1), dim3(64), 0, first, word2);
    hipLaunchKernelGGL(k_probe_set, dim3(1), dim3(64), 0, second, word2);
}

size_t panel_lds_bytes(int fmax, int panel_max)
{
    (void)fmax;
    return ((size_t)8 * NB + 2 * NB * NB + (kBdCols + NB) + 2 + (size_t)panel_max + 256) * sizeof(double);
}

static void init_factor_lds()
{
    static PerDeviceOnce once;
    once.run([]() {
        hipError_t e = hipSuccess;
        auto set = [&](auto k) { if (e == hipSuccess) e = set_max_lds(k, 160 * 1024); };
        set(k_panel<256, false, false>);
        set(k_panel<512, false, false>);
        set(k_panel<1024, false, false>);
        set(k_panel<256, false, true>);
        set(k_panel<512, false, true>);
        set(k_panel<1024, false, true>);
        set(k_panel<1024, true, false>);
        set(k_panel<1024, true, true>);
        set(k_front_wave);
        return e;
    });
}

void launch_front_tiny(const FactorArgs& a, int begin, int count, hipStream_t st)
{
    if (count <= 0) return;
    hipLaunchKernelGGL(k_front_tiny, dim3((count + 31) / 32), dim3(256), 0, st, a, begin, count);
}
void launch_front_wave(const FactorArgs& a, int begin, int count, int slice_doubles, hipStream_t st)
{
    if (count <= 0) return;
    init_factor_lds();
    hipLaunchKernelGGL(k_front_wave, dim3((count + 3) / 4), dim3(256), (size_t)4 * slice_doubles * sizeof(double), st,
                       a, begin, count, slice_doubles);
}
void launch_panel(const FactorArgs& a, int begin, int count, int bs, size_t lds, hipStream_t st)
{
    if (count <= 0) return;
    init_factor_lds();
    if (a.ov) {
        if (bs == 1024) hipLaunchKernelGGL((k_panel<1024, false, true>), dim3(count), dim3(1024), lds, st, a, begin);
        else if (bs == 512) hipLaunchKernelGGL((k_panel<512, false, true>), dim3(count), dim3(512), lds, st, a, begin);
        else hipLaunchKernelGGL((k_panel<256, false, true>), dim3(count), dim3(256), lds, st, a, begin);
        return;
    }
    if (bs == 1024) hipLaunchKernelGGL((k_panel<1024, false, false>), dim3(count), dim3(1024), lds, st, a, begin);
    else if (bs == 512) hipLaunchKernelGGL((k_panel<512, false, false>), dim3(count), dim3(512), lds, st, a, begin);
    else hipLaunchKernelGGL((k_panel<256, false, false>), dim3(count), dim3(256), lds, st, a, begin);
}
void launch_panel_sliced(const FactorArgs& a, int begin, int count, size_t lds, hipStream_t st)
{
    if (count <= 0) return;
    init_factor_lds();
    if (a.ov) hipLaunchKernelGGL((k_panel<1024, true, true>), dim3(count), dim3(1024), lds, st, a, begin);
    else hipLaunchKernelGGL((k_panel<1024, true, false>), dim3(count), dim3(1024), lds, st, a, begin);
}
void launch_schur(const FactorArgs& a, const int2* tiles, int tile_begin, int ntiles, hipStream_t st, int ov_grid, int tile_nc)
{
    if (ntiles <= 0) return;
    // (the pipelined chunk loop outside the overlap mode: launches of thousands of tiles whose products are four chunks deep
    //  or more -- the trailing blocks of very large fronts, cfg5's wide levels; on cfg2's wide middle levels, one to three
    //  chunks per tile, its registers cost more than the latency it hides, see k_schur)
    const int pipe_tiles = knobs().schur_pipe_tiles;
    const int pipe_nc = knobs().schur_pipe_nc;
    if (a.ov) hipLaunchKernelGGL((k_schur<true, true>), dim3(std::max(1, std::min(ntiles, ov_grid))), dim3(256), 0, st, a, tiles, tile_begin, ntiles);
    else if (ntiles > pipe_tiles && tile_nc >= pipe_nc) hipLaunchKernelGGL((k_schur<false, true>), dim3(ntiles), dim3(256), 0, st, a, tiles, tile_begin, ntiles);
    else hipLaunchKernelGGL((k_schur<false, false>), dim3(ntiles), dim3(256), 0, st, a, tiles, tile_begin, ntiles);
}

// ------------------------------------------------------------------------------------------------------------
//  Litmus test of the hand-over contract above (test infrastructure: hipkkt_selftest_handover; nothing in the product
//  path calls it).  Workgroup 2p produces, workgroup 2p + 1 consumes (consecutive workgroups run on different XCDs);
//  per round the producer stores `words` payload words (round * 1024 + index), signals, and waits for the consumer's
//  acknowledgement before the next round; the consumer waits for the signal, reads the payload and counts every word
//  that is not this round's.  VARIANT 0: the contract.  1: without P2 (no s_waitcnt before the signal).  2: without
//  P1 / C2 (plain payload stores and loads, the waits kept).  Every spin is bounded; a timeout is counted and ends
//  the pair.
// ------------------------------------------------------------------------------------------------------------
template <int VARIANT>
__global__ __launch_bounds__(256) void k_handover_litmus(double* payload, int* sig, int* ack, int words, int rounds,
                                                         unsigned long long* mismatches, unsigned long long* timeouts)
{
    __shared__ int sh_ok;
    const int pair = blockIdx.x >> 1, tid = threadIdx.x;
    const bool producer = (blockIdx.x & 1) == 0;
    double* pl = payload + (size_t)pair * words;
    const long long limit = 20000000;                 // 200 ms
    unsigned long long bad = 0;
    for (int r = 1; r <= rounds; ++r) {
        const long long t0 = wall_clock64();
        if (tid == 0) sh_ok = 1;
        __syncthreads();
        if (producer) {
            for (int w = tid; w < words; w += 256) {
                const double v = (double)r * 1024.0 + (double)w;
                if (VARIANT == 2) pl[w] = v; else OV_ST(pl + w, v);
            }
            if (VARIANT != 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) {
                OV_ST(sig + pair, r);
                // the consumer's acknowledgement before the payload is overwritten
                for (;;) {
                    if (OV_LD(ack + pair) >= r) break;
                    if (wall_clock64() - t0 > limit) { sh_ok = 0; atomicAdd(timeouts, 1ull); break; }
                    __builtin_amdgcn_s_sleep(1);
                }
            }
            __syncthreads();
            if (!sh_ok) return;
        } else {
            if (tid == 0) {
                for (;;) {
                    if (OV_LD(sig + pair) >= r) break;
                    if (wall_clock64() - t0 > limit) { sh_ok = 0; atomicAdd(timeouts, 1ull); break; }
                }
            }
            __syncthreads();
            if (!sh_ok) return;
            for (int w = tid; w < words; w += 256) {
                const double v = VARIANT == 2 ? pl[w] : OV_LD(pl + w);
                if (v != (double)r * 1024.0 + (double)w) ++bad;
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid == 0) OV_ST(ack + pair, r);
        }
    }
    if (bad) atomicAdd(mismatches, bad);
}
void launch_handover_litmus(int variant, double* payload, int* sig, int* ack, int pairs, int words, int rounds,
                            unsigned long long* mismatches, unsigned long long* timeouts, hipStream_t st)
{
    const dim3 grid(2 * pairs), block(256);
    if (variant == 0) hipLaunchKernelGGL(k_handover_litmus<0>, grid, block, 0, st, payload, sig, ack, words, rounds, mismatches, timeouts);
    else if (variant == 1) hipLaunchKernelGGL(k_handover_litmus<1>, grid, block, 0, st, payload, sig, ack, words, rounds, mismatches, timeouts);
    else hipLaunchKernelGGL(k_handover_litmus<2>, grid, block, 0, st, payload, sig, ack, words, rounds, mismatches, timeouts);
}

}  // namespace hipkkt
