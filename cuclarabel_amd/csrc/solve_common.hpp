// Device helpers shared by the sweep kernels (solve_kernels.hip, chain_kernels.hip).  gfx950 (wave64) only.
#pragma once
#include "kernels.hpp"

namespace hipkkt {

__device__ inline double readlane_f64(double v, int lane)
{
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Sum over the wave's 64 lanes, returned in every lane.  DPP row shifts inside the 16-lane rows, then the two row
// broadcasts: 6 steps of two 32-bit DPP moves and one add, no LDS crossbar (the __shfl_xor butterfly costs two
// ds_bpermute round trips per step: a 45-column slice of the backward sweep spent most of its time in them).
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_add_step(double v)
{
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xf, false);
    return v + __hiloint2double(hi, lo);           // (lanes without a source, or outside the row mask, add +0.0)
}
// ... and over aligned groups of eight lanes (the tiny fronts): neighbours, the quad's other pair, the other quad
__device__ inline double group8_sum(double v)
{
    v = dpp_add_step<0xB1, 0xf>(v);                // quad_perm [1,0,3,2]
    v = dpp_add_step<0x4E, 0xf>(v);                // quad_perm [2,3,0,1] -> every lane holds its quad's sum
    v = dpp_add_step<0x141, 0xf>(v);               // row_half_mirror: the other quad's sum
    return v;
}
__device__ inline double wave_reduce_sum(double v)
{
    v = dpp_add_step<0x111, 0xf>(v);               // row_shr:1
    v = dpp_add_step<0x112, 0xf>(v);               // row_shr:2
    v = dpp_add_step<0x114, 0xf>(v);               // row_shr:4
    v = dpp_add_step<0x118, 0xf>(v);               // row_shr:8   -> lane 15 of every row holds the row's sum
    v = dpp_add_step<0x142, 0xa>(v);               // row_bcast:15 into rows 1 and 3
    v = dpp_add_step<0x143, 0xc>(v);               // row_bcast:31 into rows 2 and 3 -> lane 63 holds the total
    return readlane_f64(v, 63);
}

// Every kernel below takes NR right-hand sides at once (NR = 1, 2 or 4; column c of a vector v lives at
// v + c * ld_v, SolveArgs::ld_*): the entries of L / W, the gather lists and the row indices are fetched ONCE and used
// for all NR columns.  These sweeps are bound by the latency of the tree's dependency chain, not by arithmetic, so two
// columns cost little more than one -- which is what lets an interior-point iteration's independent solves (constant
// and affine right-hand sides, /root/reference/src/kktsystem.jl:87-88 vs :170-171) share a sweep.
// ------------------------------------------------------------------ small fronts, one wave each
// The sweeps' internal vectors (xp, uvec) keep their NR columns interleaved: entry i of column c at i * NR + c.
// One 8 * NR-byte access fetches / stores an entry of every column (the stores are 256-byte aligned allocations).
template <int NR>
__device__ inline void ldv(const double* __restrict__ base, int64_t i, double (&out)[NR])
{
    if constexpr (NR == 1) out[0] = base[i];
    else if constexpr (NR == 2) {
        const double2 t = *reinterpret_cast<const double2*>(base + 2 * i);
        out[0] = t.x; out[1] = t.y;
    } else {
        static_assert(NR == 4, "1, 2 or 4 right-hand sides");
        const double4 t = *reinterpret_cast<const double4*>(base + 4 * i);
        out[0] = t.x; out[1] = t.y; out[2] = t.z; out[3] = t.w;
    }
}
template <int NR>
__device__ inline void stv(double* __restrict__ base, int64_t i, const double (&v)[NR])
{
    if constexpr (NR == 1) base[i] = v[0];
    else if constexpr (NR == 2) *reinterpret_cast<double2*>(base + 2 * i) = make_double2(v[0], v[1]);
    else *reinterpret_cast<double4*>(base + 4 * i) = make_double4(v[0], v[1], v[2], v[3]);
}

constexpr int kItemsInFlight = 2;     // matrix items (8 loads per lane each) a wave of the block solve kernels fetches at a time
// Sum of the n partials p[0], p[stride], p[2 stride], ... in index order (the order fixes the rounding), their LDS loads
// issued eight at a time: a plain loop compiles to read - wait - add per term, ~70 cycles each, which was a quarter
// of a hop of the persistent kernel (27 terms per entry in the backward sweep).
__device__ inline double lds_sum_strided(const double* p, int n, int stride)
{
    double v = 0.0;
    int k = 0;
    for (; k + 8 <= n; k += 8) {
        double t[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) t[q] = p[(k + q) * stride];
#pragma unroll
        for (int q = 0; q < 8; ++q) v += t[q];
    }
    double t[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) t[q] = (k + q < n) ? p[(k + q) * stride] : 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) if (k + q < n) v += t[q];
    return v;
}

struct ItemRegs { double m[8]; };
// part[ks*ldp + r] = sum_q R.m[q] * v[8 ks + q]  for the item (row block rb, column slice ks)
__device__ inline void item_apply(const ItemRegs& R, const double* v, int Rn, int Kn, double* part, int ldp, int it,
                                  int nrb, int lane)
{
    const int ks = it / nrb, rb = it - ks * nrb;
    const int r = rb * 64 + lane, k0 = 8 * ks;
    double acc = 0.0;
#pragma unroll
    for (int q = 0; q < 8; ++q) acc = fma(R.m[q], (k0 + q < Kn) ? v[k0 + q] : 0.0, acc);
    if (r < Rn) part[ks * ldp + r] = acc;
}

typedef __attribute__((address_space(1))) double gdouble;
typedef __attribute__((address_space(1))) int gint;
#define LD_AGENT_F64(p) __hip_atomic_load((p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define ST_AGENT_F64(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)

// The tail of a row's gather list, GPB sources per pair of load rounds (indices, then values) instead of two dependent
// loads per source; sums in list order.
template <int GPB>
__device__ inline double gather_rest(const TreeDev& T, const double* uvec, int64_t g, int64_t g1, double v)
{
    for (; g < g1; g += GPB) {
        int src[GPB];
        double u[GPB];
#pragma unroll
        for (int q = 0; q < GPB; ++q) src[q] = (g + q < g1) ? T.gl_src[g + q] : -1;
#pragma unroll
        for (int q = 0; q < GPB; ++q) u[q] = src[q] >= 0 ? LD_AGENT_F64(uvec + src[q]) : 0.0;
#pragma unroll
        for (int q = 0; q < GPB; ++q) if (src[q] >= 0) v += u[q];
    }
    return v;
}

__device__ inline bool wait_flag(int* flag, int epoch, int* abort_word, long long t0, long long limit)
{
    for (;;) {
        if (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == epoch) return true;
        if (__hip_atomic_load(abort_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) return false;
        if (wall_clock64() - t0 > limit) {
            __hip_atomic_store(abort_word, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// ------------------------------------------------------------------ packed sweep records (kernels.hpp: SolveHdr, RecSeg)
// What a forward step knows about one of its rows before any value is loaded: the first six sources of the row's gather
// list (indices into uvec) and where the list goes on in gl_src.
struct RowGather { int cnt; int src[6]; int ov; };

// the record of the launch's item-th front of class k (0 block-class, 1 one-wave, 2 tiny), or null: legacy layout
__device__ __forceinline__ const char* rec_of(const SolveArgs& A, const RecSeg& R, int k, int item)
{
    return A.recs ? A.recs + R.off[k] + (int64_t)item * R.stride[k] : nullptr;
}
__device__ __forceinline__ int rec_idx(const char* rec, int i) { return reinterpret_cast<const int*>(rec + sizeof(SolveHdr))[i]; }
__device__ __forceinline__ RowGather rec_gather(const char* rec, int fmax, int i)
{
    const int4* g = reinterpret_cast<const int4*>(rec + sizeof(SolveHdr) + 4 * (size_t)fmax) + 2 * i;
    const int4 a = g[0], b = g[1];
    return RowGather{a.x, {a.y, a.z, a.w, b.x, b.y, b.z}, b.w};
}
// the same from the legacy lists (local row lc = c0 + rp + i)
__device__ __forceinline__ RowGather row_gather_lists(const TreeDev& T, int64_t lc)
{
    const int64_t g0 = T.gl_ptr[lc], g1 = T.gl_ptr[lc + 1];
    RowGather G;
    G.cnt = (int)(g1 - g0);
    G.ov = (int)(g0 + 6);
#pragma unroll
    for (int q = 0; q < 6; ++q) G.src[q] = (g0 + q < g1) ? T.gl_src[g0 + q] : -1;
    return G;
}
// v += the row's contributions, in list order (the order fixes the rounding): the six known sources in one round of loads
// for all columns, the rest of the list six at a time (indices, then values).  AGENT: the values are handed over inside
// the running kernel (relaxed agent-scope loads: persistent / chained kernels); otherwise plain (vector) loads.
template <int NR, bool AGENT>
__device__ __forceinline__ void gather_add(const SolveArgs& A, const RowGather& G, double (&v)[NR])
{
    int src[6];
#pragma unroll
    for (int q = 0; q < 6; ++q) src[q] = G.src[q];
    for (int g = G.ov - 6, e = G.ov - 6 + G.cnt; g < e; g += 6) {
        if (g != G.ov - 6) {
#pragma unroll
            for (int q = 0; q < 6; ++q) src[q] = (g + q < e) ? A.T.gl_src[g + q] : -1;
        }
        double u[NR][6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            double t[NR];
#pragma unroll
            for (int c = 0; c < NR; ++c) t[c] = 0.0;
            if (src[q] >= 0) {
                if constexpr (AGENT) {
#pragma unroll
                    for (int c = 0; c < NR; ++c) t[c] = LD_AGENT_F64(A.uvec + (int64_t)src[q] * NR + c);
                } else {
                    ldv<NR>(A.uvec, src[q], t);
                }
            }
#pragma unroll
            for (int c = 0; c < NR; ++c) u[c][q] = t[c];
        }
#pragma unroll
        for (int c = 0; c < NR; ++c)
#pragma unroll
            for (int q = 0; q < 6; ++q) if (src[q] >= 0) v[c] += u[c][q];
    }
}

}  // namespace hipkkt
