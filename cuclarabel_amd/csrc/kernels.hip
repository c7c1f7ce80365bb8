// See kernels.hpp.  Hand-written HIP for gfx950 (CDNA4, wave64).
#include "kernels.hpp"
#include "knobs.hpp"
#include <algorithm>

#include <cmath>

namespace hipkkt {

// =====================================================================================
//  KKT value updates
// =====================================================================================
static inline int grid_for(int64_t n, int bs, int cap = 4096)
{
    int64_t g = (n + bs - 1) / bs;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

__global__ void k_scatter(double* __restrict__ K, const int* __restrict__ idx, const double* __restrict__ v,
                          int64_t n, double scale)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        K[idx[i]] = v[i] * scale;
}
void launch_scatter(double* K, const int* idx, const double* v, int64_t n, double scale, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scatter, dim3(grid_for(n, 256)), dim3(256), 0, st, K, idx, v, n, scale);
}
// The cone update of an iteration in ONE launch: -Hs into K (kktsolver_directldl.jl:225-228) and the sparse second-order
// cones' columns u, v and D (:235-241), each value also written THROUGH to its one or two slots of the residual's
// CSR-ordered copy (kpos: two slots per K entry, -1 = none) -- the copy used to be re-gathered whole (2.5 M entries,
// 13 us) after every update although only these entries change.
__device__ __forceinline__ void put_k(double* __restrict__ K, double* __restrict__ fval, const int* __restrict__ kpos, int e, double v)
{
    K[e] = v;
    if (fval) {
        const int p0 = kpos[2 * e], p1 = kpos[2 * e + 1];
        if (p0 >= 0) fval[p0] = v;
        if (p1 >= 0) fval[p1] = v;
    }
}
__global__ void k_update_values(double* __restrict__ K, const int* __restrict__ mapHs, const double* __restrict__ Hs, int nHs,
                                const int* __restrict__ mapU, const int* __restrict__ mapV, const int* __restrict__ mapD,
                                const double* __restrict__ u, const double* __restrict__ v, const double* __restrict__ eta2,
                                const int* __restrict__ soc_of_entry, int sparse_len, int nsparse,
                                double* __restrict__ fval, const int* __restrict__ kpos)
{
    const int total = nHs + sparse_len + nsparse;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        if (i < nHs) {
            put_k(K, fval, kpos, mapHs[i], -Hs[i]);
        } else if (i < nHs + sparse_len) {
            const int j = i - nHs;
            const double e2 = eta2[soc_of_entry[j]];
            put_k(K, fval, kpos, mapU[j], u[j] * (-e2));
            put_k(K, fval, kpos, mapV[j], v[j] * (-e2));
        } else {
            const int j = i - nHs - sparse_len;
            put_k(K, fval, kpos, mapD[2 * j], -eta2[j]);
            put_k(K, fval, kpos, mapD[2 * j + 1], eta2[j]);
        }
    }
}
void launch_update_values(double* K, const int* mapHs, const double* Hs, int nHs, const int* mapU, const int* mapV,
                          const int* mapD, const double* u, const double* v, const double* eta2, const int* soc_of_entry,
                          int sparse_len, int nsparse, double* fval, const int* kpos, hipStream_t st)
{
    const int total = nHs + sparse_len + nsparse;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_update_values, dim3(grid_for(total, 256)), dim3(256), 0, st, K, mapHs, Hs, nHs, mapU, mapV, mapD, u, v,
                       eta2, soc_of_entry, sparse_len, nsparse, fval, kpos);
}
__global__ void k_scale(double* __restrict__ K, const int* __restrict__ idx, double scale, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        K[idx[i]] *= scale;
}
void launch_scale(double* K, const int* idx, double scale, int64_t n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_scale, dim3(grid_for(n, 256)), dim3(256), 0, st, K, idx, scale, n);
}

// sparse second-order cones: K[u] = u*(-eta2), K[v] = v*(-eta2), D = (-eta2, +eta2)
// (directldl_datamaps.jl:61-79: update then scale, which is the single product written here)
__global__ void k_soc_columns(double* __restrict__ K, const int* __restrict__ mapU, const int* __restrict__ mapV,
                              const int* __restrict__ mapD, const double* __restrict__ u,
                              const double* __restrict__ v, const double* __restrict__ eta2,
                              const int* __restrict__ soc_of_entry, int sparse_len, int nsparse)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < sparse_len) {
        const double e2 = eta2[soc_of_entry[i]];
        K[mapU[i]] = u[i] * (-e2);
        K[mapV[i]] = v[i] * (-e2);
    }
    if (i < nsparse) {
        K[mapD[2 * i]] = -eta2[i];
        K[mapD[2 * i + 1]] = eta2[i];
    }
}
void launch_soc_columns(double* K, const int* mapU, const int* mapV, const int* mapD, const double* u,
                        const double* v, const double* eta2, const int* soc_of_entry, int sparse_len,
                        int nsparse, hipStream_t st)
{
    int n = sparse_len > nsparse ? sparse_len : nsparse;
    if (n <= 0) return;
    hipLaunchKernelGGL(k_soc_columns, dim3((n + 255) / 256), dim3(256), 0, st, K, mapU, mapV, mapD, u, v, eta2,
                       soc_of_entry, sparse_len, nsparse);
}

// ---- reductions: block max -> partial[], then one small block finishes
__device__ inline double block_max_256(double v, double* sh)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_down(v, o, 64));
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) sh[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < (int)(blockDim.x >> 6); ++w) v = fmax(v, sh[w]);
    }
    return v;   // valid in thread 0
}

constexpr int kRedBlocks = 256;

__global__ void k_diag_absmax(const double* __restrict__ K, const int* __restrict__ diag, int N,
                              double* __restrict__ partial)
{
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        v = fmax(v, fabs(K[diag[i]]));
    v = block_max_256(v, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = v;
}
__global__ void k_finish_regularizer(const double* __restrict__ partial, int nparts, double c0, double c1,
                                     double* __restrict__ eps_out)
{
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) v = fmax(v, partial[i]);
    v = block_max_256(v, sh);
    if (threadIdx.x == 0) eps_out[0] = c0 + c1 * v;
}
void launch_regularizer(const double* K, const int* diag, int N, double c0, double c1, double* partial,
                        double* eps_out, hipStream_t st)
{
    int g = grid_for(N, 256, kRedBlocks);
    hipLaunchKernelGGL(k_diag_absmax, dim3(g), dim3(256), 0, st, K, diag, N, partial);
    hipLaunchKernelGGL(k_finish_regularizer, dim3(1), dim3(256), 0, st, partial, g, c0, c1, eps_out);
}

// =====================================================================================
//  Residual e = b - K_sym x (kktsolver_directldl.jl:455-466) on the full symmetric CSR image of
//  the un-regularised K; G lanes cooperate on one row (fixed in-row summation order).
// =====================================================================================
// NC columns per workgroup (1, or 2: the 2-column solves' residual reads the matrix ONCE for both columns -- cfg3's
// 25 M entries: 180 -> see DESIGN.md us per 2-column residual); column c of b, x, e at c * ld, its partials as before
template <int G, int NC>
__global__ __launch_bounds__(256) void k_residual(SpmvDev A, const double* __restrict__ K,
                                                  const double* __restrict__ b, const double* __restrict__ x,
                                                  double* __restrict__ e, double* __restrict__ partial, int64_t ld,
                                                  double* __restrict__ bpartial)
{
    __shared__ double sh[4];
    const int col0 = blockIdx.y * NC;
    b += col0 * ld;
    x += col0 * ld;
    e += col0 * ld;
    partial += col0 * (gridDim.x + 1);
    const int sub = threadIdx.x % G;
    const int rows_per_block = 256 / G;
    if (bpartial) bpartial += col0 * gridDim.x;
    double vmax[NC], bmax[NC];               // bmax: ||b||_inf rides along when bpartial is given
    bool bad[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { vmax[c] = 0.0; bmax[c] = 0.0; bad[c] = false; }
    for (int row = blockIdx.x * rows_per_block + threadIdx.x / G; row < A.N; row += gridDim.x * rows_per_block) {
        const int64_t q0 = A.ptr[row], q1 = A.ptr[row + 1];
        if (q1 - q0 > kLongRow) continue;               // handled by k_residual_long_*
        double acc[NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] = 0.0;
        for (int64_t q = q0 + sub; q < q1; q += G) {
            const double a = A.val ? A.val[q] : K[A.vmap[q]];
            const int j = A.col[q];
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = fma(a, x[c * ld + j], acc[c]);
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
#pragma unroll
            for (int o = G / 2; o > 0; o >>= 1) acc[c] += __shfl_down(acc[c], o, G);
        }
        if (sub == 0) {
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double bv = b[c * ld + row];
                const double r = bv - acc[c];
                e[c * ld + row] = r;
                if (!isfinite(r)) bad[c] = true;
                vmax[c] = fmax(vmax[c], fabs(r));
                bmax[c] = isfinite(bv) ? fmax(bmax[c], fabs(bv)) : INFINITY;
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        if (c) __syncthreads();
        double v = bad[c] ? INFINITY : vmax[c];        // marks non-finite; finished as NaN below
        v = block_max_256(v, sh);
        if (threadIdx.x == 0) {
            partial[c * (gridDim.x + 1) + blockIdx.x] = v;
            if (blockIdx.x == 0 && A.nlong == 0) partial[c * (gridDim.x + 1) + gridDim.x] = 0.0;     // the long rows' slot
        }
        if (bpartial) {
            __syncthreads();
            const double w = block_max_256(bmax[c], sh);
            if (threadIdx.x == 0) bpartial[c * gridDim.x + blockIdx.x] = w;
        }
    }
}
// one workgroup per chunk of a long row: fixed assignment of entries to threads, fixed reduction tree
__global__ __launch_bounds__(256) void k_residual_long_chunks(SpmvDev A, const double* __restrict__ K,
                                                              const double* __restrict__ x, int64_t ld)
{
    __shared__ double sh[256];
    x += blockIdx.y * ld;
    const int64_t q0 = A.chunk_q[2 * blockIdx.x], q1 = A.chunk_q[2 * blockIdx.x + 1];
    double acc = 0.0;
    for (int64_t q = q0 + threadIdx.x; q < q1; q += 256)
        acc = fma(A.val ? A.val[q] : K[A.vmap[q]], x[A.col[q]], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) A.long_partial[(int64_t)blockIdx.y * A.nchunks + blockIdx.x] = sh[0];
}
__global__ __launch_bounds__(256) void k_residual_long_finish(SpmvDev A, const double* __restrict__ b,
                                                              double* __restrict__ e, double* __restrict__ partial,
                                                              int64_t ld, int gmain)
{
    __shared__ double sh[4];
    b += blockIdx.y * ld;
    e += blockIdx.y * ld;
    const double* lp = A.long_partial + (int64_t)blockIdx.y * A.nchunks;
    double vmax = 0.0;
    bool bad = false;
    for (int t = threadIdx.x; t < A.nlong; t += 256) {
        const int row = A.long_rows[t];
        double acc = 0.0;
        for (int64_t c = A.long_chunk_ptr[t]; c < A.long_chunk_ptr[t + 1]; ++c) acc += lp[c];
        const double r = b[row] - acc;
        e[row] = r;
        if (!isfinite(r)) bad = true;
        vmax = fmax(vmax, fabs(r));
    }
    if (bad) vmax = INFINITY;
    vmax = block_max_256(vmax, sh);
    if (threadIdx.x == 0) partial[blockIdx.y * (gmain + 1) + gmain] = vmax;
}
__global__ void k_finish_norm(const double* __restrict__ partial, int nparts, double* __restrict__ out,
                              const int* __restrict__ flag_in = nullptr, double* __restrict__ flag_out = nullptr,
                              const double* __restrict__ bpartial = nullptr, int nbparts = 0,
                              double* __restrict__ bout = nullptr)
{
    if (flag_out && blockIdx.x == 0 && threadIdx.x == 0) flag_out[0] = (flag_in && flag_in[0]) ? 1.0 : 0.0;
    __shared__ double sh[4];
    double v = 0.0;
    partial += blockIdx.x * nparts;
    out += blockIdx.x;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) v = fmax(v, partial[i]);
    v = block_max_256(v, sh);
    // norm(e, Inf) of a vector holding Inf or NaN is not finite either way; the caller only
    // tests isfinite() (kktsolver_directldl.jl:411,429)
    if (threadIdx.x == 0) out[0] = v;
    if (bpartial) {                          // (column blockIdx.x's ||b||: its partials sit nbparts apart)
        __syncthreads();
        double w = 0.0;
        for (int i = threadIdx.x; i < nbparts; i += blockDim.x) w = fmax(w, bpartial[blockIdx.x * nbparts + i]);
        w = block_max_256(w, sh);
        if (threadIdx.x == 0) bout[blockIdx.x] = w;
    }
}
int residual_grid(const SpmvDev& A)
{
    const int rows_per_block = 256 / A.lanes_per_row;
    int g = (A.N + rows_per_block - 1) / rows_per_block;
    if (g > kNormParts) g = kNormParts;
    return g < 1 ? 1 : g;
}
void launch_residual(const SpmvDev& A, const double* K, const double* b, const double* x, double* e,
                     double* partial, double* norm_out, hipStream_t st, int nrhs, int64_t ld, const int* flag_in,
                     double* flag_out, double* normb_out)
{
    const int g = residual_grid(A);
    // ||b||_inf in the same pass (no long rows: every row's b is read here anyway); partial then holds nrhs * (g + 1)
    // residual partials followed by nrhs * g partials of b, at most kResidualPartial(nrhs) doubles
    double* bpartial = (normb_out && nrhs <= kMaxNormbCols && A.nlong == 0) ? partial + (size_t)nrhs * (g + 1) : nullptr;
    if (nrhs % 2 == 0) {
        if (A.lanes_per_row == 8)
            hipLaunchKernelGGL((k_residual<8, 2>), dim3(g, nrhs / 2), dim3(256), 0, st, A, K, b, x, e, partial, ld, bpartial);
        else
            hipLaunchKernelGGL((k_residual<64, 2>), dim3(g, nrhs / 2), dim3(256), 0, st, A, K, b, x, e, partial, ld, bpartial);
    } else if (A.lanes_per_row == 8)
        hipLaunchKernelGGL((k_residual<8, 1>), dim3(g, nrhs), dim3(256), 0, st, A, K, b, x, e, partial, ld, bpartial);
    else
        hipLaunchKernelGGL((k_residual<64, 1>), dim3(g, nrhs), dim3(256), 0, st, A, K, b, x, e, partial, ld, bpartial);
    if (A.nlong > 0) {
        hipLaunchKernelGGL(k_residual_long_chunks, dim3(A.nchunks, nrhs), dim3(256), 0, st, A, K, x, ld);
        hipLaunchKernelGGL(k_residual_long_finish, dim3(1, nrhs), dim3(256), 0, st, A, b, e, partial, ld, g);
    }
    // norm_out == nullptr: the caller's next kernel reduces the partials itself (residual_partials_ok; k_ir_round)
    if (!norm_out) return;
    hipLaunchKernelGGL(k_finish_norm, dim3(nrhs), dim3(256), 0, st, partial, g + 1, norm_out, flag_in, flag_out,
                       (const double*)bpartial, g, normb_out);
    if (normb_out && !bpartial) launch_norm_inf(b, A.N, partial, normb_out, st, nrhs, ld);
}
__global__ void k_absmax(const double* __restrict__ v, int n, double* __restrict__ partial, int64_t ld)
{
    __shared__ double sh[4];
    v += blockIdx.y * ld;
    partial += blockIdx.y * gridDim.x;
    double m = 0.0;
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const double a = fabs(v[i]);
        if (!isfinite(a)) bad = true;
        m = fmax(m, a);
    }
    if (bad) m = INFINITY;
    m = block_max_256(m, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = m;
}
void launch_norm_inf(const double* v, int n, double* partial, double* out, hipStream_t st, int nrhs, int64_t ld)
{
    int g = grid_for(n, 256, kRedBlocks);
    hipLaunchKernelGGL(k_absmax, dim3(g, nrhs), dim3(256), 0, st, v, n, partial, ld);
    hipLaunchKernelGGL(k_finish_norm, dim3(nrhs), dim3(256), 0, st, partial, g, out, (const int*)nullptr, (double*)nullptr,
                       (const double*)nullptr, 0, (double*)nullptr);
}
__global__ void k_gather_values(double* __restrict__ val, const double* __restrict__ K, const int* __restrict__ vmap,
                                int64_t nnz)
{
    for (int64_t q = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; q < nnz; q += (int64_t)gridDim.x * blockDim.x)
        val[q] = K[vmap[q]];
}
void launch_gather_values(double* val, const double* Kval, const int* vmap, int64_t nnz, hipStream_t st)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_gather_values, dim3(grid_for(nnz, 256, 8192)), dim3(256), 0, st, val, Kval, vmap, nnz);
}
__global__ void k_sum2(double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b, int64_t n)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        y[i] = a[i] + b[i];
}
void launch_axpby_sum(double* y, const double* a, const double* b, int64_t n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sum2, dim3(grid_for(n, 256)), dim3(256), 0, st, y, a, b, n);
}
__global__ void k_pack_rhs(double* __restrict__ b, const double* __restrict__ rx, const double* __restrict__ rz,
                           int n, int m, int p)
{
    const int N = n + m + p;
    b += (int64_t)blockIdx.y * N;
    rx += (int64_t)blockIdx.y * n;
    rz += (int64_t)blockIdx.y * m;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        b[i] = i < n ? rx[i] : (i < n + m ? rz[i - n] : 0.0);      // kktsolver_directldl.jl:313-327
}
void launch_pack_rhs(double* b, const double* rx, const double* rz, int n, int m, int p, hipStream_t st, int nrhs)
{
    hipLaunchKernelGGL(k_pack_rhs, dim3(grid_for(n + m + p, 256), nrhs), dim3(256), 0, st, b, rx, rz, n, m, p);
}
// level C, affine kkt_solve! with the constant right-hand side riding along (kktsystem.jl:87-88, :157-173): column 0 =
// (-q, b), column 1 = (rhs.x, s - rhs.z) -- the affine step's Delta_s constant term is variables.s itself, so the
// right-hand side needs no separate offset pass; ncol = 1: column 1's content alone, in column 0
__global__ void k_pack_rhs_affine(double* __restrict__ b, const double* __restrict__ negq, const double* __restrict__ bb,
                                  const double* __restrict__ rhs_x, const double* __restrict__ s,
                                  const double* __restrict__ rhs_z, int n, int m, int p, int ncol)
{
    const int N = n + m + p;
    const bool aff = ncol == 1 || blockIdx.y == 1;
    b += (int64_t)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        double v = 0.0;
        if (i < n) v = aff ? rhs_x[i] : negq[i];
        else if (i < n + m) v = aff ? s[i - n] - rhs_z[i - n] : bb[i - n];
        b[i] = v;
    }
}
void launch_pack_rhs_affine(double* b, const double* negq, const double* bb, const double* rhs_x, const double* s,
                            const double* rhs_z, int n, int m, int p, int ncol, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_rhs_affine, dim3(grid_for(n + m + p, 256), ncol), dim3(256), 0, st, b, negq, bb, rhs_x, s, rhs_z,
                       n, m, p, ncol);
}
__global__ void k_accept_columns(double* __restrict__ x, const double* __restrict__ cand, double* __restrict__ e,
                                 const double* __restrict__ e2, const int* __restrict__ mask, int N)
{
    if (!mask[blockIdx.y]) return;
    const int64_t o = (int64_t)blockIdx.y * N;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x) {
        x[o + i] = cand[o + i];
        e[o + i] = e2[o + i];
    }
}
void launch_accept_columns(double* x, const double* cand, double* e, const double* e2, const int* mask, int N,
                           int nrhs, hipStream_t st)
{
    if (N <= 0 || nrhs <= 0) return;
    hipLaunchKernelGGL(k_accept_columns, dim3(grid_for(N, 256, 1024), nrhs), dim3(256), 0, st, x, cand, e, e2, mask, N);
}
__global__ void k_unpack_lhs(double* __restrict__ lhsx, double* __restrict__ lhsz, const double* __restrict__ x, int n, int m)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x) {
        if (i < n) { if (lhsx) lhsx[i] = x[i]; }
        else if (lhsz) lhsz[i - n] = x[i];
    }
}
void launch_unpack_lhs(double* lhsx, double* lhsz, const double* x, int n, int m, hipStream_t st)
{
    if (n + m <= 0 || (!lhsx && !lhsz)) return;
    hipLaunchKernelGGL(k_unpack_lhs, dim3(grid_for(n + m, 256)), dim3(256), 0, st, lhsx, lhsz, x, n, m);
}
__global__ void k_zero_ints(int* __restrict__ p, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
// several arrays in ONE launch (grid.y = array): a factorisation zeroes up to six small counters' arrays, and at the head
// of a step every launch costs the stream ~5 us whatever it does
__global__ void k_zero_ints_multi(ZeroList Z)
{
    int* __restrict__ p = Z.p[blockIdx.y];
    const int n = Z.n[blockIdx.y];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = 0;
}
void launch_zero_ints_multi(const ZeroList& Z, hipStream_t st)
{
    if (Z.count <= 0) return;
    int nmax = 1;
    for (int k = 0; k < Z.count; ++k) nmax = Z.n[k] > nmax ? Z.n[k] : nmax;
    hipLaunchKernelGGL(k_zero_ints_multi, dim3(grid_for(nmax, 256, 256), Z.count), dim3(256), 0, st, Z);
}
// the value update's status words in one place, so that ONE small copy brings them to the host
__global__ void k_collect_status(double* __restrict__ dst, const double* __restrict__ eps, const int* __restrict__ conefail,
                                 const int* __restrict__ flags, double* __restrict__ sticky)
{
    if (threadIdx.x == 0) {
        const double e = eps ? eps[0] : 0.0, cf = conefail ? (double)conefail[0] : 0.0;
        const double f0 = (double)flags[0], f1 = (double)flags[1], f2 = (double)flags[2];
        dst[0] = e;
        dst[1] = cf;
        dst[2] = f0;
        dst[3] = f1;
        dst[4] = f2;                     // an overlap-mode wait of the factorisation gave up
        if (sticky) {                    // deferred status: folded into the sticky record here (k_fold_update_status's rule)
            if (cf != 0.0 || f1 != 0.0) sticky[0] = 1.0;
            if (f2 != 0.0) sticky[7] = 1.0;
            sticky[4] += f0;
            sticky[5] = e;
        }
    }
}
void launch_collect_status(double* dst, const double* eps, const int* conefail, const int* flags, hipStream_t st, double* sticky)
{
    hipLaunchKernelGGL(k_collect_status, dim3(1), dim3(64), 0, st, dst, eps, conefail, flags, sticky);
}
// ---- the refinement loop's decisions on the device (kktsolver_directldl.jl:389-449).  State after round r
// (slot r of `state`, 4 doubles): {active: the reference's loop would go on, rounds done, bad: a residual norm was not
// finite, norme: residual norm of the accepted solution}.  Round r >= 1 has just produced the candidate dx = x + K^-1 e
// and its residual norm scal[4]; this kernel applies the reference's accept / stop rule to it and, if the candidate is
// accepted, makes it the solution (x <- dx; e already holds its residual).  Every thread evaluates the rule from the
// same read-only words; thread 0 of workgroup 0 publishes the new state.  r == 0: only the initial state.
__device__ inline void ir_initial(double norme0, double normb, double abstol, double reltol, int max_iter,
                                  double& active, double& rounds, double& bad, double& norme)
{
    norme = norme0;
    bad = isfinite(norme) ? 0.0 : 1.0;
    rounds = 0.0;
    active = (bad == 0.0 && max_iter > 0 && !(norme <= abstol + reltol * normb)) ? 1.0 : 0.0;
}
// grid.y = right-hand side column c: its state at state + c * state_stride, norms norme0[c], normb[c], cand[c], vectors
// x + c * ld, dx + c * ld, read-back record at readback + 5 c.  The sticky record is folded by column 0 only when
// there is a single column (several columns: k_ir_fold).
// The residual kernels' partial maxima may come un-finished (IrPartials; launch_residual with norm_out = nullptr): every
// workgroup of this kernel then takes the maximum over them itself -- a maximum does not depend on the order, so all
// workgroups decide alike -- instead of a one-workgroup kernel behind every residual (~5-8 us each on the solve's chain
// of dependent launches).  The grid is small (kIrBlocks) so that the partials are read a hundred times, not a thousand.
// the maxima of up to three arrays of partials at once (null: 0), in every thread: one round of loads, one pair of barriers
__device__ inline void block_max3_of(const double* __restrict__ p0, int n0, const double* __restrict__ p1, int n1,
                                     const double* __restrict__ p2, int n2, double (&out)[3], double (*sh)[4])
{
    double v[3] = {0.0, 0.0, 0.0};
    // (every load of the three arrays issued before the first maximum: one memory round trip, not one per array and trip)
    constexpr int kTrips = (kNormParts + 1 + 255) / 256;
    double t[3][kTrips];
#pragma unroll
    for (int k = 0; k < kTrips; ++k) {
        const int i = (int)threadIdx.x + 256 * k;
        t[0][k] = (p0 && i < n0) ? p0[i] : 0.0;
        t[1][k] = (p1 && i < n1) ? p1[i] : 0.0;
        t[2][k] = (p2 && i < n2) ? p2[i] : 0.0;
    }
#pragma unroll
    for (int k = 0; k < kTrips; ++k) { v[0] = fmax(v[0], t[0][k]); v[1] = fmax(v[1], t[1][k]); v[2] = fmax(v[2], t[2][k]); }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) v[k] = fmax(v[k], __shfl_down(v[k], o, 64));
        if (lane == 0) sh[k][wave] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 3; ++k) out[k] = fmax(fmax(sh[k][0], sh[k][1]), fmax(sh[k][2], sh[k][3]));
}
__global__ __launch_bounds__(256) void k_ir_round(double* __restrict__ state, int state_stride, int r, int first, double* __restrict__ norme0,
                           double* __restrict__ normb, const double* __restrict__ cand_norm,
                           const double* __restrict__ abort_word, double* __restrict__ x, const double* __restrict__ dx,
                           int n, int64_t ld, double abstol, double reltol, double stop_ratio, int max_iter,
                           double* __restrict__ readback, double* __restrict__ sticky, IrPartials Q, int several)
{
    __shared__ double sh[3][4];
    const int c = blockIdx.y;
    state += (int64_t)c * state_stride;
    x += c * ld;
    dx += c * ld;
    double active, rounds, bad, norme;
    double nb = 0.0, cand_v = 0.0;
    const bool initial = r == 0 || first;
    double mx[3] = {0.0, 0.0, 0.0};
    if (Q.e0 || Q.cand)
        block_max3_of((initial && Q.e0) ? Q.e0 + (int64_t)c * Q.np : nullptr, Q.np, (initial && Q.e0) ? Q.b0 + (int64_t)c * (Q.np - 1) : nullptr,
                      Q.np - 1, (r > 0 && Q.cand) ? Q.cand + (int64_t)c * Q.np : nullptr, Q.np, mx, sh);
    if (initial) {
        double ne0;
        if (Q.e0) {
            ne0 = mx[0]; nb = mx[1];
            if (blockIdx.x == 0 && threadIdx.x == 0) { norme0[c] = ne0; normb[c] = nb; }       // (later rounds read them)
        } else { ne0 = norme0[c]; nb = normb[c]; }
        ir_initial(ne0, nb, abstol, reltol, max_iter, active, rounds, bad, norme);
    } else {
        active = state[4 * (r - 1)]; rounds = state[4 * (r - 1) + 1]; bad = state[4 * (r - 1) + 2]; norme = state[4 * (r - 1) + 3];
        nb = normb[c];
    }
    if (r > 0) cand_v = Q.cand ? mx[2] : cand_norm[c];
    bool accept = false;
    if (r > 0 && active != 0.0) {
        const double cand = cand_v;
        rounds += 1.0;
        if (!isfinite(cand)) {                      // :429: return is_success = false
            bad = 1.0; active = 0.0;
        } else {
            const double ratio = norme / cand;      // :437-446
            if (ratio < stop_ratio) { accept = ratio > 1.0; active = 0.0; }
            else accept = true;
            if (accept) norme = cand;
            // the head of the next iteration (:407-417): tolerance met, or max_iter rounds done
            if (active != 0.0 && (r >= max_iter || norme <= abstol + reltol * nb)) active = 0.0;
        }
    }
    if (accept) {
        const int stride = gridDim.x * blockDim.x;
        for (int i0 = blockIdx.x * blockDim.x + threadIdx.x; i0 < n; i0 += 4 * stride) {       // (four loads in flight)
            double v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = (i0 + u * stride < n) ? dx[i0 + u * stride] : 0.0;
#pragma unroll
            for (int u = 0; u < 4; ++u) if (i0 + u * stride < n) x[i0 + u * stride] = v[u];
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        state[4 * r] = active; state[4 * r + 1] = rounds; state[4 * r + 2] = bad; state[4 * r + 3] = norme;
        const double aborted = Q.flag_in ? (Q.flag_in[0] ? 1.0 : 0.0) : (abort_word ? abort_word[0] : 0.0);
        if (readback) {
            double* rb = readback + 5 * c;
            rb[0] = active; rb[1] = rounds; rb[2] = bad; rb[3] = norme; rb[4] = aborted;
        }
        if (sticky) {                               // deferred status: the worst over all calls since the last query
            if (bad != 0.0) sticky[0] = 1.0;
            if (active != 0.0) sticky[1] = 1.0;
            if (aborted != 0.0) sticky[2] = 1.0;
            if (several) {                          // (every column's workgroup 0 adds its own: whole numbers, any order)
                (void)__hip_atomic_fetch_add(sticky + 3, rounds, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                (void)__hip_atomic_fetch_add(sticky + 6, 1.0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                sticky[3] += rounds;
                sticky[6] += 1.0;
            }
        }
    }
}
void launch_ir_round(double* state, int state_stride, int r, bool first, double* norme0, double* normb,
                     const double* cand, const double* abort_word, double* x, const double* dx, int n, int nr, double abstol,
                     double reltol, double stop_ratio, int max_iter, double* readback, double* sticky, hipStream_t st,
                     const IrPartials& Q)
{
    constexpr int kIrBlocks = 304;
    const int g = r == 0 ? 1 : grid_for(n, 256, Q.e0 || Q.cand ? kIrBlocks : 4096);
    hipLaunchKernelGGL(k_ir_round, dim3(g, nr), dim3(256), 0, st, state, state_stride, r, first ? 1 : 0, norme0, normb, cand,
                       abort_word, x, dx, n, (int64_t)n, abstol, reltol, stop_ratio, max_iter, readback, sticky, Q, nr > 1 ? 1 : 0);
}
// several columns: their final states (slot r of each) joined into the sticky record by one thread
__global__ void k_ir_fold(const double* __restrict__ state, int state_stride, int r, int nr, const double* __restrict__ abort_word,
                          double* __restrict__ sticky)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    for (int c = 0; c < nr; ++c) {
        const double* s = state + (int64_t)c * state_stride + 4 * r;
        if (s[2] != 0.0) sticky[0] = 1.0;
        if (s[0] != 0.0) sticky[1] = 1.0;
        sticky[3] += s[1];
        sticky[6] += 1.0;
    }
    if (abort_word && abort_word[0] != 0.0) sticky[2] = 1.0;
}
void launch_ir_fold(const double* state, int state_stride, int r, int nr, const double* abort_word, double* sticky, hipStream_t st)
{
    hipLaunchKernelGGL(k_ir_fold, dim3(1), dim3(64), 0, st, state, state_stride, r, nr, abort_word, sticky);
}
// deferred status of a value update: {eps, cone failure, #dynamic regularisations, non-finite pivot} (k_collect_status's
// words) folded into the sticky record
__global__ void k_fold_update_status(double* __restrict__ sticky, const double* __restrict__ st4)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        if (st4[1] != 0.0 || st4[3] != 0.0) sticky[0] = 1.0;
        if (st4[4] != 0.0) sticky[7] = 1.0;
        sticky[4] += st4[2];
        sticky[5] = st4[0];
    }
}
void launch_fold_update_status(double* sticky, const double* st4, hipStream_t st)
{
    hipLaunchKernelGGL(k_fold_update_status, dim3(1), dim3(64), 0, st, sticky, st4);
}
__global__ void k_fold_flag(double* __restrict__ sticky, const int* __restrict__ flag)
{
    if (threadIdx.x == 0 && blockIdx.x == 0 && flag[0] != 0) sticky[0] = 1.0;
}
void launch_fold_flag(double* sticky, const int* flag, hipStream_t st)
{
    hipLaunchKernelGGL(k_fold_flag, dim3(1), dim3(64), 0, st, sticky, flag);
}
void launch_zero_ints(int* p, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_zero_ints, dim3(grid_for(n, 64, 256)), dim3(64), 0, st, p, n);
}
__global__ void k_check_finite(const double* __restrict__ v, int n, int* __restrict__ flag)
{
    bool bad = false;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
        if (!isfinite(v[i])) bad = true;
    if (bad) *flag = 1;
}
void launch_check_finite(const double* v, int n, int* flag, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_check_finite, dim3(grid_for(n, 256, 1024)), dim3(256), 0, st, v, n, flag);
}

// =====================================================================================
//  Cone scalings on the device.
//    zero cone        Hs = 0                          coneops_zerocone.jl:78-102
//    nonnegative      w = sqrt(s/z), Hs = w^2         coneops_nncone.jl:77-101
//    second-order     eta, w (NT point), sparse (d, u, v) or dense 2ww' - J
//                                                     coneops_socone.jl:75-192
//    PSD (side <= 48)  A = R R' via Cholesky + Jacobi eigen, Hs = A (x)_s A   coneops_psdtrianglecone.jl:78-161
// =====================================================================================
// An elementwise kernel and its one-wave-per-second-order-cone companion touch disjoint rows, so they go out as ONE
// launch (k_cone_scaling, k_mul_Hs, k_sys_offset below): workgroups [0, ge) run the elementwise body over a grid of ge,
// the ones behind them take four cones each, a wave per cone.  On the step's chain of dependent launches a kernel
// boundary is ~5 us whatever the work.
__device__ inline void cone_elementwise_body(const ConeDev& C, const ConeState& S, const double* __restrict__ s,
                                             const double* __restrict__ z, int m, int bx, int nb)
{
    for (int i = bx * 256 + threadIdx.x; i < m; i += nb * 256) {
        const int c = C.elem_cone[i];
        const int kind = C.kind[c];
        if (kind == 0) {
            S.w[i] = 0.0;
            if (S.lam) S.lam[i] = 0.0;
            S.Hs[C.boff[c] + (i - C.off[c])] = 0.0;
        } else if (kind == 1) {
            const double w = sqrt(s[i] / z[i]);
            S.w[i] = w;
            if (S.lam) S.lam[i] = sqrt(s[i] * z[i]);
            S.Hs[C.boff[c] + (i - C.off[c])] = w * w;
        }
    }
}

__device__ inline double wave_sum(double v)
{
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// one wave per second-order cone
__device__ inline void cone_soc_body(const ConeDev& C, const ConeState& S, const double* __restrict__ s,
                                     const double* __restrict__ z, int ci, int lane)
{
    const int c = C.soc_list[ci];
    const int off = C.off[c], n = C.numel[c];
    const double* sc = s + off;
    const double* zc = z + off;
    double* w = S.w + off;
    // residuals (z0 - ||z1||)(z0 + ||z1||), coneops_socone.jl:415-425
    double ss = 0.0, zz = 0.0;
    for (int i = 1 + lane; i < n; i += 64) { ss += sc[i] * sc[i]; zz += zc[i] * zc[i]; }
    ss = sqrt(wave_sum(ss));
    zz = sqrt(wave_sum(zz));
    const double s0 = sc[0], z0 = zc[0];
    double sres = (s0 - ss) * (s0 + ss), zres = (z0 - zz) * (z0 + zz);
    const double sscale = sres > 0.0 ? sqrt(sres) : 0.0, zscale = zres > 0.0 ? sqrt(zres) : 0.0;
    if (sscale == 0.0 || zscale == 0.0) { if (lane == 0) *S.fail = 1; return; }
    const double eta = sqrt(sscale / zscale);
    // w = s/sscale + J z/zscale, normalised
    double w1sq = 0.0;
    for (int i = 1 + lane; i < n; i += 64) {
        const double wi = sc[i] / sscale - zc[i] / zscale;
        w[i] = wi;
        w1sq += wi * wi;
    }
    w1sq = wave_sum(w1sq);
    const double w0 = s0 / sscale + z0 / zscale;
    const double w1n = sqrt(w1sq);
    const double wres = (w0 - w1n) * (w0 + w1n);
    const double wscale = wres > 0.0 ? sqrt(wres) : 0.0;
    if (wscale == 0.0) { if (lane == 0) *S.fail = 1; return; }
    double w1sqn = 0.0;
    for (int i = 1 + lane; i < n; i += 64) {
        const double wi = w[i] / wscale;
        w[i] = wi;
        w1sqn += wi * wi;
    }
    w1sqn = wave_sum(w1sqn);
    const double w0n = sqrt(1.0 + w1sqn);
    if (lane == 0) { w[0] = w0n; S.eta[c] = eta; }
    if (S.lam) {
        // scaling point lambda = W z = W^{-T} s (coneops_socone.jl:113-123)
        double* lam = S.lam + off;
        const double gamma = 0.5 * wscale;
        const double a = (gamma + z0 / zscale) / sscale, b = (gamma + s0 / sscale) / zscale;
        const double inv = 1.0 / (s0 / sscale + z0 / zscale + 2.0 * gamma);
        const double sz = sqrt(sscale * zscale);
        for (int i = 1 + lane; i < n; i += 64) lam[i] = (a * sc[i] + b * zc[i]) * inv * sz;
        if (lane == 0) lam[0] = gamma * sz;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "workgroup");      // w[] written above is re-read below by other lanes of this wave
    const double eta2 = eta * eta;
    double* Hs = S.Hs + C.boff[c];
    const int sidx = C.sidx[c];
    if (sidx >= 0) {
        // sparse form: D = eta^2 [d, 1, ..., 1]; u, v for the two extension columns (:125-151)
        const double alpha = 2.0 * w0n;
        const double wsq = w0n * w0n + w1sqn, wsqinv = 1.0 / wsq;
        const double d = wsqinv / 2.0;
        const double u0 = sqrt(wsq - d), u1 = alpha / u0;
        const double v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
        double* u = S.u + C.soff[c];
        double* v = S.v + C.soff[c];
        for (int i = lane; i < n; i += 64) {
            if (i == 0) { u[0] = u0; v[0] = 0.0; Hs[0] = eta2 * d; }
            else { const double wi = w[i]; u[i] = u1 * wi; v[i] = v1 * wi; Hs[i] = eta2; }
        }
        if (lane == 0) S.eta2[sidx] = eta2;
    } else {
        // dense form (dim <= 4): packed triu of eta^2 (2 w w' - J)   (:168-186)
        if (lane == 0) {
            Hs[0] = (sqrt(2.0) * w0n - 1.0) * (sqrt(2.0) * w0n + 1.0) * eta2;
            int h = 1;
            for (int col = 1; col < n; ++col) {
                const double wc = w[col];
                for (int row = 0; row <= col; ++row) {
                    const double wr = row == 0 ? w0n : w[row];
                    double val = 2.0 * wr * wc;
                    if (row == col) val += 1.0;
                    Hs[h++] = val * eta2;
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_cone_scaling(ConeDev C, ConeState S, const double* __restrict__ s,
                                                      const double* __restrict__ z, int m, int ge)
{
    if ((int)blockIdx.x < ge) { cone_elementwise_body(C, S, s, z, m, blockIdx.x, ge); return; }
    const int ci = ((int)blockIdx.x - ge) * 4 + (int)(threadIdx.x >> 6);
    if (ci < C.nsoc) cone_soc_body(C, S, s, z, ci, threadIdx.x & 63);
}

// -------------------------------------------------------------------------------------
//  PSD cones (coneops_psdtrianglecone.jl:78-161): one workgroup per cone, everything in LDS.
//    S, Z from svec;  L1 = chol(S), L2 = chol(Z)  (failure => not interior);
//    SVD of M = L2' L1 = U Lam V'  -- as the reference does (:118-121), NOT an eigen-decomposition of
//    L1' Z L1 = M'M, whose eigenvalues are the squares of the singular values: forming it squares the condition
//    number, and on late interior-point iterates (S, Z nearly complementary) the small singular values drop below
//    eps ||M'M||.  The SVD is a one-sided (Hestenes) Jacobi on the columns of M, which computes every singular
//    value to high RELATIVE accuracy; rotations in the round-robin order (a sweep is m - 1 steps of m/2 rotations
//    on disjoint column pairs, eight lanes per pair, one barrier per step).
//    lam = singular values sorted descending (LAPACK's order, dense_algebra.jl:219);
//    R = L1 V Lam^{-1/2}, Rinv = Lam^{-1/2} U' L2' (:127-132); A = R R' (:135-141);
//    Hs = A (x)_s A (skron!, :502-540) written straight into its packed upper triangle by a flat map over the
//    t(t+1)/2 entries, t = k(k+1)/2.
// -------------------------------------------------------------------------------------
__device__ inline void svec_index(int idx, int& row, int& col)     // idx = col(col+1)/2 + row, row <= col
{
    int c = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while (c * (c + 1) / 2 > idx) --c;
    while ((c + 1) * (c + 2) / 2 <= idx) ++c;
    col = c;
    row = idx - c * (c + 1) / 2;
}

__global__ __launch_bounds__(256) void k_cone_psd(ConeDev C, ConeState S, const double* __restrict__ s,
                                                  const double* __restrict__ z)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    __shared__ int sh_fail;
    __shared__ double sh_off[32], sh_sv[kPsdMaxDim];
    __shared__ int sh_rank[kPsdMaxDim];
    const int tid = threadIdx.x;
    const int c = C.psd_list[blockIdx.x];
    const int k = C.psd_dim[c], off = C.off[c];
    const int kk = k * k, t = k * (k + 1) / 2;
    double* Sm = smem;            // S -> L1 (lower)
    double* Zm = smem + kk;       // Z -> L2 (lower)
    double* M = smem + 2 * kk;    // L2' L1 -> U diag(sv) -> U
    double* V = smem + 3 * kk;    // right singular vectors
    double* Tm = smem + 4 * kk;   // R
    double* Am = smem + 5 * kk;   // A = R R'
    const double is2 = 0.70710678118654752440;
    if (tid == 0) sh_fail = 0;
    for (int idx = tid; idx < t; idx += 256) {
        int r, cl;
        svec_index(idx, r, cl);
        const double sv = s[off + idx], zv = z[off + idx];
        if (r == cl) { Sm[r + cl * k] = sv; Zm[r + cl * k] = zv; }
        else {
            Sm[r + cl * k] = Sm[cl + r * k] = sv * is2;
            Zm[r + cl * k] = Zm[cl + r * k] = zv * is2;
        }
    }
    __syncthreads();
    // Cholesky of S and of Z, in place, right-looking (:97-104)
    for (int pass = 0; pass < 2; ++pass) {
        double* Mx = pass == 0 ? Sm : Zm;
        for (int j = 0; j < k; ++j) {
            const double d = Mx[j + j * k];
            if (!(d > 0.0)) { if (tid == 0) sh_fail = 1; }
            __syncthreads();
            if (sh_fail) break;
            const double sd = sqrt(d);
            for (int i = j + 1 + tid; i < k; i += 256) Mx[i + j * k] /= sd;
            if (tid == 0) Mx[j + j * k] = sd;
            __syncthreads();
            for (int idx = tid; idx < (k - j - 1) * (k - j - 1); idx += 256) {
                const int a = j + 1 + idx / (k - j - 1), b = j + 1 + idx % (k - j - 1);
                if (a >= b) Mx[a + b * k] -= Mx[a + j * k] * Mx[b + j * k];
            }
            __syncthreads();
        }
        if (sh_fail) break;
    }
    if (sh_fail) { if (tid == 0) *S.fail = 1; return; }
    for (int idx = tid; idx < kk; idx += 256) {
        const int r = idx % k, cl = idx / k;
        if (r < cl) { Sm[idx] = 0.0; Zm[idx] = 0.0; }
    }
    __syncthreads();
    // M = L2' L1 (:113-114), V = I
    for (int idx = tid; idx < kk; idx += 256) {
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        for (int q = (r > cl ? r : cl); q < k; ++q) acc = fma(Zm[q + r * k], Sm[q + cl * k], acc);
        M[idx] = acc;
        V[idx] = (r == cl) ? 1.0 : 0.0;
    }
    __syncthreads();
    // one-sided Jacobi: rotate column pairs (p, q) of M (and of V) until all columns are mutually orthogonal
    const int m = (k + 1) & ~1, npair = m / 2;          // k <= 48: at most 24 pairs, eight lanes each
    const int grp = tid >> 3, sub = tid & 7;
    for (int sweep = 0; sweep < 60; ++sweep) {
        double myoff = 0.0;
        for (int step = 0; step < m - 1; ++step) {
            if (grp < npair) {
                int p, q;
                if (grp == 0) { p = m - 1; q = step; }
                else { p = (step + grp) % (m - 1); q = (step - grp + (m - 1)) % (m - 1); }
                if (p > q) { const int tmp = p; p = q; q = tmp; }
                if (q < k) {                                           // (q == k: the dummy index of an odd k)
                    double a = 0.0, b = 0.0, cc = 0.0;
                    for (int i = sub; i < k; i += 8) {
                        const double mp = M[i + p * k], mq = M[i + q * k];
                        a = fma(mp, mp, a); b = fma(mq, mq, b); cc = fma(mp, mq, cc);
                    }
#pragma unroll
                    for (int o = 4; o > 0; o >>= 1) {
                        a += __shfl_xor(a, o, 8); b += __shfl_xor(b, o, 8); cc += __shfl_xor(cc, o, 8);
                    }
                    const double ab = sqrt(a * b);
                    if (!(fabs(cc) <= 1e-300 || fabs(cc) <= 1e-17 * ab)) {
                        myoff = fmax(myoff, fabs(cc) / ab);
                        const double zeta = (b - a) / (2.0 * cc);
                        const double tt = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                        const double cs = 1.0 / sqrt(1.0 + tt * tt), sn = cs * tt;
                        for (int i = sub; i < k; i += 8) {
                            const double mp = M[i + p * k], mq = M[i + q * k];
                            M[i + p * k] = cs * mp - sn * mq;
                            M[i + q * k] = sn * mp + cs * mq;
                            const double vp = V[i + p * k], vq = V[i + q * k];
                            V[i + p * k] = cs * vp - sn * vq;
                            V[i + q * k] = sn * vp + cs * vq;
                        }
                    }
                }
            }
            __syncthreads();               // the next step pairs the columns differently
        }
        if (sub == 0 && grp < 32) sh_off[grp] = grp < npair ? myoff : 0.0;
        __syncthreads();
        double offn = 0.0;
        for (int i = 0; i < npair; ++i) offn = fmax(offn, sh_off[i]);   // uniform: every thread reads the same words
        __syncthreads();
        if (offn < 1e-15) break;
    }
    // singular values = column norms; U = M with unit columns; descending order like LAPACK's
    if (tid < k) {
        double nrm = 0.0;
        for (int i = 0; i < k; ++i) nrm = fma(M[i + tid * k], M[i + tid * k], nrm);
        sh_sv[tid] = sqrt(nrm);
    }
    __syncthreads();
    if (tid < k) {
        const double mine = sh_sv[tid];
        int rank = 0;
        for (int i = 0; i < k; ++i) rank += (sh_sv[i] > mine || (sh_sv[i] == mine && i < tid)) ? 1 : 0;
        sh_rank[tid] = rank;
        if (S.lam) S.lam[off + rank] = mine;                   // lam occupies the first k of the cone's t slots
    }
    if (S.lam) for (int i = k + tid; i < t; i += 256) S.lam[off + i] = 0.0;
    __syncthreads();
    // R = L1 V Lam^{-1/2} (columns in sorted order), Rinv = Lam^{-1/2} U' L2' (rows in sorted order)
    double* Rout = S.psdR + C.psd_aoff[c];
    double* Riout = S.psdRinv + C.psd_aoff[c];
    for (int idx = tid; idx < kk; idx += 256) {
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        for (int q = 0; q <= r; ++q) acc = fma(Sm[r + q * k], V[q + cl * k], acc);
        const double sc = 1.0 / sqrt(sh_sv[cl]);
        const double v = acc * sc;
        Tm[r + sh_rank[cl] * k] = v;
        Rout[r + sh_rank[cl] * k] = v;
        // Rinv(row cl of the unsorted order, column r): (1/sqrt(s_cl)) (1/s_cl) sum_q M(q, cl) L2(r, q)
        double acc2 = 0.0;
        for (int q = 0; q <= r; ++q) acc2 = fma(M[q + cl * k], Zm[r + q * k], acc2);
        Riout[sh_rank[cl] + r * k] = acc2 * sc / sh_sv[cl];
    }
    __syncthreads();
    double* Aout = S.psdA + C.psd_aoff[c];
    for (int idx = tid; idx < kk; idx += 256) {
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        if (r <= cl) {                                           // upper triangle (syrk 'U'), mirrored: exactly symmetric
            for (int q = 0; q < k; ++q) acc = fma(Tm[r + q * k], Tm[cl + q * k], acc);
            Am[r + cl * k] = acc; Am[cl + r * k] = acc;
            Aout[r + cl * k] = acc; Aout[cl + r * k] = acc;
        }
    }
    __syncthreads();
    // Hs = A (x)_s A, packed upper triangle (column-major over (row, col) with row <= col)
    double* Hs = S.Hs + C.boff[c];
    const double s2 = 1.41421356237309504880;
    const int nh = t * (t + 1) / 2;
    for (int e = tid; e < nh; e += 256) {
        int row, col;
        svec_index(e, row, col);
        int i, j, kq, l;
        svec_index(row, i, j);          // row <-> (i, j), i <= j
        svec_index(col, kq, l);         // col <-> (k, l), k <= l
        double v;
        const bool ij = (i == j), kl = (kq == l);
        if (!ij && !kl) v = Am[i + kq * k] * Am[j + l * k] + Am[i + l * k] * Am[j + kq * k];
        else if (ij && !kl) v = s2 * Am[j + l * k] * Am[j + kq * k];
        else if (!ij && kl) v = s2 * Am[i + l * k] * Am[j + kq * k];
        else v = Am[j + l * k] * Am[j + l * k];
        Hs[e] = v;
    }
}

// y = (A (x)_s A) x = svec(A X A) for PSD cones
__global__ __launch_bounds__(256) void k_mul_Hs_psd(ConeDev C, ConeState S, double* __restrict__ y,
                                                    const double* __restrict__ x, const double* __restrict__ addend)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int c = C.psd_list[blockIdx.x];
    const int k = C.psd_dim[c], off = C.off[c], kk = k * k, t = k * (k + 1) / 2;
    double* X = smem;
    double* Am = smem + kk;
    double* Tm = smem + 2 * kk;
    const double is2 = 0.70710678118654752440;
    const double* A = S.psdA + C.psd_aoff[c];
    for (int idx = tid; idx < kk; idx += 256) Am[idx] = A[idx];
    for (int idx = tid; idx < t; idx += 256) {
        int r, cl;
        svec_index(idx, r, cl);
        const double v = x[off + idx];
        if (r == cl) X[r + cl * k] = v;
        else X[r + cl * k] = X[cl + r * k] = v * is2;
    }
    __syncthreads();
    for (int idx = tid; idx < kk; idx += 256) {
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        for (int q = 0; q < k; ++q) acc = fma(X[r + q * k], Am[q + cl * k], acc);
        Tm[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < t; idx += 256) {
        int r, cl;
        svec_index(idx, r, cl);
        double a1 = 0.0, a2 = 0.0;
        for (int q = 0; q < k; ++q) { a1 = fma(Am[r + q * k], Tm[q + cl * k], a1); a2 = fma(Am[cl + q * k], Tm[q + r * k], a2); }
        const double v = (r == cl) ? a1 : (a1 + a2) * is2;
        y[off + idx] = addend ? -(v + addend[off + idx]) : v;
    }
}

void launch_cone_scaling(const ConeDev& C, const ConeState& S, const double* s, const double* z, int m,
                         hipStream_t st)
{
    const int ge = m > 0 ? grid_for(m, 256) : 0, gs = (C.nsoc + 3) / 4;
    if (ge + gs > 0) hipLaunchKernelGGL(k_cone_scaling, dim3(ge + gs), dim3(256), 0, st, C, S, s, z, m, ge);
    if (C.npsd > 0) {
        static PerDeviceOnce once;
        once.run([]() { return set_max_lds(k_cone_psd, 150 * 1024); });
        const size_t lds = (size_t)6 * C.psd_kmax * C.psd_kmax * sizeof(double);
        hipLaunchKernelGGL(k_cone_psd, dim3(C.npsd), dim3(256), lds, st, C, S, s, z);
    }
}

// y = W'W x : zero -> 0, NN -> w*(w*x), SOC -> eta^2 (2 w (w'x) - J x)   (mul_Hs!); with an addend: y = -(W'W x + addend),
// the Delta_s recovery of kkt_solve! (kktsystem.jl:206-212) in the same pass
__device__ inline void mul_Hs_elementwise_body(const ConeDev& C, const ConeState& S, double* __restrict__ y,
                                               const double* __restrict__ x, int m, const double* __restrict__ addend, int bx, int nb)
{
    for (int i = bx * 256 + threadIdx.x; i < m; i += nb * 256) {
        const int kind = C.kind[C.elem_cone[i]];
        double v;
        if (kind == 0) v = 0.0;
        else if (kind == 1) v = S.w[i] * (S.w[i] * x[i]);
        else continue;
        y[i] = addend ? -(v + addend[i]) : v;
    }
}
__device__ inline void mul_Hs_soc_body(const ConeDev& C, const ConeState& S, double* __restrict__ y,
                                       const double* __restrict__ x, const double* __restrict__ addend, int ci, int lane)
{
    const int c = C.soc_list[ci];
    const int off = C.off[c], n = C.numel[c];
    const double* w = S.w + off;
    double dot = 0.0;
    for (int i = lane; i < n; i += 64) dot += w[i] * x[off + i];
    dot = 2.0 * wave_sum(dot);
    const double e2 = S.eta[c] * S.eta[c];
    for (int i = lane; i < n; i += 64) {
        const double xi = x[off + i];
        const double v = ((i == 0 ? -xi : xi) + dot * w[i]) * e2;
        y[off + i] = addend ? -(v + addend[off + i]) : v;
    }
}
__device__ inline void publish_record(const Publish& P)
{
    for (int i = 0; i < P.n; ++i) P.dst[i] = P.rec[i];
    P.dst[P.n] = P.seq;
    for (int i = 0; i < P.nzero; ++i) P.rec[i] = 0.0;
}
__global__ void k_publish(Publish P) { if (blockIdx.x == 0 && threadIdx.x == 0) publish_record(P); }
void launch_publish(const Publish& p, hipStream_t st)
{
    if (p.dst) hipLaunchKernelGGL(k_publish, dim3(1), dim3(64), 0, st, p);
}
__global__ __launch_bounds__(256) void k_mul_Hs(ConeDev C, ConeState S, double* __restrict__ y, const double* __restrict__ x,
                                                int m, const double* __restrict__ addend, int ge, Publish P)
{
    // (the record was completed by earlier kernels of the stream and is not touched by this one)
    if (P.dst && blockIdx.x == gridDim.x - 1 && threadIdx.x == 255) publish_record(P);
    if ((int)blockIdx.x < ge) { mul_Hs_elementwise_body(C, S, y, x, m, addend, blockIdx.x, ge); return; }
    const int ci = ((int)blockIdx.x - ge) * 4 + (int)(threadIdx.x >> 6);
    if (ci < C.nsoc) mul_Hs_soc_body(C, S, y, x, addend, ci, threadIdx.x & 63);
}
// A = R R' per PSD cone, from a caller-supplied R (hipkkt_kkt_system_update_cones); one workgroup per cone
__global__ __launch_bounds__(256) void k_psd_A_from_R(ConeDev C, ConeState S)
{
    const int c = C.psd_list[blockIdx.x];
    const int k = C.psd_dim[c];
    const double* R = S.psdR + C.psd_aoff[c];
    double* A = S.psdA + C.psd_aoff[c];
    for (int idx = threadIdx.x; idx < k * k; idx += 256) {
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        for (int q = 0; q < k; ++q) acc = fma(R[r + q * k], R[cl + q * k], acc);
        A[idx] = acc;
    }
}
void launch_psd_A_from_R(const ConeDev& C, const ConeState& S, hipStream_t st)
{
    if (C.npsd > 0) hipLaunchKernelGGL(k_psd_A_from_R, dim3(C.npsd), dim3(256), 0, st, C, S);
}

// ---- get_Hs! and the sparse second-order-cone vectors from a GIVEN scaling (w, eta; psdA = R R'): what a caller that
// holds the reference's cone objects need not send over PCIe -- Hsblocks, u, v, eta^2 are functions of (w, eta, R)
// (coneops_nncone.jl:91-101, coneops_socone.jl:125-192, coneops_psdtrianglecone.jl:135-161).  The arithmetic is the
// tail of k_cone_elementwise / k_cone_soc / k_cone_psd, term for term and in the same order, so that a scaling computed
// on the device and fed back gives bit-identical K values for the elementwise and second-order cones.
__global__ void k_hs_from_w_elementwise(ConeDev C, ConeState S, int m)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int c = C.elem_cone[i];
        const int kind = C.kind[c];
        if (kind == 0) S.Hs[C.boff[c] + (i - C.off[c])] = 0.0;
        else if (kind == 1) { const double w = S.w[i]; S.Hs[C.boff[c] + (i - C.off[c])] = w * w; }
    }
}
__global__ __launch_bounds__(64) void k_soc_from_w(ConeDev C, ConeState S)
{
    const int c = C.soc_list[blockIdx.x];
    const int off = C.off[c], n = C.numel[c];
    const int lane = threadIdx.x;
    const double* w = S.w + off;
    double w1sqn = 0.0;
    for (int i = 1 + lane; i < n; i += 64) { const double wi = w[i]; w1sqn += wi * wi; }
    w1sqn = wave_sum(w1sqn);
    const double w0n = w[0];
    const double eta = S.eta[c];
    const double eta2 = eta * eta;
    double* Hs = S.Hs + C.boff[c];
    const int sidx = C.sidx[c];
    if (sidx >= 0) {
        const double alpha = 2.0 * w0n;
        const double wsq = w0n * w0n + w1sqn, wsqinv = 1.0 / wsq;
        const double d = wsqinv / 2.0;
        const double u0 = sqrt(wsq - d), u1 = alpha / u0;
        const double v1 = sqrt(2.0 * (2.0 + wsqinv) / (2.0 * wsq - wsqinv));
        double* u = S.u + C.soff[c];
        double* v = S.v + C.soff[c];
        for (int i = lane; i < n; i += 64) {
            if (i == 0) { u[0] = u0; v[0] = 0.0; Hs[0] = eta2 * d; }
            else { const double wi = w[i]; u[i] = u1 * wi; v[i] = v1 * wi; Hs[i] = eta2; }
        }
        if (lane == 0) S.eta2[sidx] = eta2;
    } else if (lane == 0) {
        Hs[0] = (sqrt(2.0) * w0n - 1.0) * (sqrt(2.0) * w0n + 1.0) * eta2;
        int h = 1;
        for (int col = 1; col < n; ++col) {
            const double wc = w[col];
            for (int row = 0; row <= col; ++row) {
                const double wr = row == 0 ? w0n : w[row];
                double val = 2.0 * wr * wc;
                if (row == col) val += 1.0;
                Hs[h++] = val * eta2;
            }
        }
    }
}
__global__ __launch_bounds__(256) void k_psd_hs_from_A(ConeDev C, ConeState S)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int c = C.psd_list[blockIdx.x];
    const int k = C.psd_dim[c], t = k * (k + 1) / 2;
    double* Am = smem;
    const double* A = S.psdA + C.psd_aoff[c];
    for (int idx = tid; idx < k * k; idx += 256) Am[idx] = A[idx];
    __syncthreads();
    double* Hs = S.Hs + C.boff[c];
    const double s2 = 1.41421356237309504880;
    const int nh = t * (t + 1) / 2;
    for (int e = tid; e < nh; e += 256) {
        int row, col;
        svec_index(e, row, col);
        int i, j, kq, l;
        svec_index(row, i, j);
        svec_index(col, kq, l);
        double v;
        const bool ij = (i == j), kl = (kq == l);
        if (!ij && !kl) v = Am[i + kq * k] * Am[j + l * k] + Am[i + l * k] * Am[j + kq * k];
        else if (ij && !kl) v = s2 * Am[j + l * k] * Am[j + kq * k];
        else if (!ij && kl) v = s2 * Am[i + l * k] * Am[j + kq * k];
        else v = Am[j + l * k] * Am[j + l * k];
        Hs[e] = v;
    }
}
void launch_cone_from_scaling(const ConeDev& C, const ConeState& S, int m, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(k_hs_from_w_elementwise, dim3(grid_for(m, 256)), dim3(256), 0, st, C, S, m);
    if (C.nsoc > 0) hipLaunchKernelGGL(k_soc_from_w, dim3(C.nsoc), dim3(64), 0, st, C, S);
    if (C.npsd > 0) {
        hipLaunchKernelGGL(k_psd_A_from_R, dim3(C.npsd), dim3(256), 0, st, C, S);
        const size_t lds = (size_t)C.psd_kmax * C.psd_kmax * sizeof(double);
        hipLaunchKernelGGL(k_psd_hs_from_A, dim3(C.npsd), dim3(256), lds, st, C, S);
    }
}
void launch_mul_Hs(const ConeDev& C, const ConeState& S, double* y, const double* x, int m, hipStream_t st, const double* addend,
                   const Publish& pub)
{
    const int ge = m > 0 ? grid_for(m, 256) : 0, gs = (C.nsoc + 3) / 4;
    const bool ride = ge + gs > 0 && C.npsd == 0;          // (the publication rides with the LAST kernel of the call)
    if (ge + gs > 0) hipLaunchKernelGGL(k_mul_Hs, dim3(ge + gs), dim3(256), 0, st, C, S, y, x, m, addend, ge, ride ? pub : Publish{});
    if (C.npsd > 0) {
        const size_t lds = (size_t)3 * C.psd_kmax * C.psd_kmax * sizeof(double);
        hipLaunchKernelGGL(k_mul_Hs_psd, dim3(C.npsd), dim3(256), lds, st, C, S, y, x, addend);
    }
    if (!ride) launch_publish(pub, st);
}

}  // namespace hipkkt


// =====================================================================================
//  Reduced-system algebra of kkt_solve! on the device (kktsystem.jl:135-215)
// =====================================================================================
namespace hipkkt {

__device__ inline void sys_offset_elementwise_body(const ConeDev& C, const ConeState& S, double* __restrict__ konst,
                                                   double* __restrict__ workz, const double* __restrict__ ds,
                                                   const double* __restrict__ z, const double* __restrict__ rhs_z, int m,
                                                   int affine, int bx, int nb)
{
    for (int i = bx * 256 + threadIdx.x; i < m; i += nb * 256) {
        double o;
        if (affine) {
            o = ds[i];                                  // Delta_s_const_term = variables.s (kktsystem.jl:157-158)
        } else {
            const int kind = C.kind[C.elem_cone[i]];
            if (kind == 0) o = 0.0;                     // coneops_zerocone.jl:137-150
            else if (kind == 1) o = ds[i] / z[i];       // coneops_nncone.jl:140-148
            else continue;                              // second-order / PSD cones: k_sys_offset_soc / k_sys_offset_psd
        }
        konst[i] = o;
        workz[i] = o - rhs_z[i];
    }
}

// out = W'(lambda \ ds) in the reference's more stable form (coneops_socone.jl:241-268); one wave per cone
__device__ inline void sys_offset_soc_body(const ConeDev& C, const ConeState& S, double* __restrict__ konst,
                                           double* __restrict__ workz, const double* __restrict__ ds,
                                           const double* __restrict__ z, const double* __restrict__ rhs_z, int ci, int lane)
{
    const int c = C.soc_list[ci];
    const int off = C.off[c], n = C.numel[c];
    const double* dsc = ds + off;
    const double* zc = z + off;
    const double* w = S.w + off;
    const double* lam = S.lam + off;
    double zz = 0.0, l1d1 = 0.0, w1d1 = 0.0;
    for (int i = 1 + lane; i < n; i += 64) {
        zz += zc[i] * zc[i];
        l1d1 += lam[i] * dsc[i];
        w1d1 += w[i] * dsc[i];
    }
    zz = sqrt(wave_sum(zz));
    l1d1 = wave_sum(l1d1);
    w1d1 = wave_sum(w1d1);
    const double z0 = zc[0];
    const double resz = (z0 - zz) * (z0 + zz);
    const double eta = S.eta[c];
    const double cc = (lam[0] * dsc[0] - l1d1) / resz;
    const double linv = 1.0 / lam[0];
    const double wfac = w1d1 / (1.0 + w[0]);
    for (int i = lane; i < n; i += 64) {
        double o;
        if (i == 0) o = z0 * cc + eta * w1d1;
        else o = -zc[i] * cc + eta * (dsc[i] + wfac * w[i]);
        o *= linv;
        konst[off + i] = o;
        workz[off + i] = o - rhs_z[off + i];
    }
}

// out = W'(lambda \ ds) for a PSD cone (_Delta_s_from_Delta_z_offset_symmetric!, coneops_symmetric_common.jl:39-52):
// X = mat(ds); X(i, j) <- 2 X(i, j) / (lam_i + lam_j) (lambda_inv_circ_op!, coneops_psdtrianglecone.jl:335-353);
// out = svec(R X R') (mul_W! with :T, :409-437).  One workgroup per cone.
__global__ __launch_bounds__(256) void k_sys_offset_psd(ConeDev C, ConeState S, double* __restrict__ konst,
                                                       double* __restrict__ workz, const double* __restrict__ ds,
                                                       const double* __restrict__ rhs_z)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int c = C.psd_list[blockIdx.x];
    const int k = C.psd_dim[c], off = C.off[c], kk = k * k, t = k * (k + 1) / 2;
    double* X = smem;
    double* Rm = smem + kk;
    double* Tm = smem + 2 * kk;
    const double is2 = 0.70710678118654752440;
    const double* R = S.psdR + C.psd_aoff[c];
    const double* lam = S.lam + off;
    for (int idx = tid; idx < kk; idx += 256) Rm[idx] = R[idx];
    for (int idx = tid; idx < t; idx += 256) {
        int r, cl;
        svec_index(idx, r, cl);
        const double v = ds[off + idx] * (r == cl ? 1.0 : is2) * 2.0 / (lam[r] + lam[cl]);
        X[r + cl * k] = v;
        X[cl + r * k] = v;
    }
    __syncthreads();
    for (int idx = tid; idx < kk; idx += 256) {          // Tm = X R'
        const int r = idx % k, cl = idx / k;
        double acc = 0.0;
        for (int q = 0; q < k; ++q) acc = fma(X[r + q * k], Rm[cl + q * k], acc);
        Tm[idx] = acc;
    }
    __syncthreads();
    for (int idx = tid; idx < t; idx += 256) {           // Y = R Tm, svec
        int r, cl;
        svec_index(idx, r, cl);
        double a1 = 0.0, a2 = 0.0;
        for (int q = 0; q < k; ++q) { a1 = fma(Rm[r + q * k], Tm[q + cl * k], a1); a2 = fma(Rm[cl + q * k], Tm[q + r * k], a2); }
        const double o = (r == cl) ? a1 : (a1 + a2) * is2;
        konst[off + idx] = o;
        workz[off + idx] = o - rhs_z[off + idx];
    }
}

__global__ __launch_bounds__(256) void k_sys_offset(ConeDev C, ConeState S, double* __restrict__ konst, double* __restrict__ workz,
                                                    const double* __restrict__ ds, const double* __restrict__ z,
                                                    const double* __restrict__ rhs_z, int m, int affine, int ge)
{
    if ((int)blockIdx.x < ge) { sys_offset_elementwise_body(C, S, konst, workz, ds, z, rhs_z, m, affine, blockIdx.x, ge); return; }
    const int ci = ((int)blockIdx.x - ge) * 4 + (int)(threadIdx.x >> 6);
    if (ci < C.nsoc) sys_offset_soc_body(C, S, konst, workz, ds, z, rhs_z, ci, threadIdx.x & 63);
}
bool launch_sys_offset(const ConeDev& C, const ConeState& S, double* konst, double* workz, const double* ds,
                       const double* z, const double* rhs_z, int m, bool affine, hipStream_t st)
{
    if (m <= 0) return true;
    if (!affine && C.npsd > 0) {
        static PerDeviceOnce once;
        once.run([]() { return set_max_lds(k_sys_offset_psd, 150 * 1024); });
        const size_t lds = (size_t)3 * C.psd_kmax * C.psd_kmax * sizeof(double);
        hipLaunchKernelGGL(k_sys_offset_psd, dim3(C.npsd), dim3(256), lds, st, C, S, konst, workz, ds, rhs_z);
    }
    const int ge = grid_for(m, 256), gs = affine ? 0 : (C.nsoc + 3) / 4;
    hipLaunchKernelGGL(k_sys_offset, dim3(ge + gs), dim3(256), 0, st, C, S, konst, workz, ds, z, rhs_z, m, affine ? 1 : 0, ge);
    return true;
}

__global__ __launch_bounds__(256) void k_P_spmv(SpmvDev A, const double* __restrict__ K, const double* __restrict__ x,
                                                double* __restrict__ y, int n)
{
    const int sub = threadIdx.x & 7;
    for (int row = blockIdx.x * 32 + (threadIdx.x >> 3); row < n; row += gridDim.x * 32) {
        double acc = 0.0;
        const int64_t qend = A.pend ? A.pend[row] : A.ptr[row + 1];
        for (int64_t q = A.ptr[row] + sub; q < qend; q += 8) {
            const int c = A.col[q];
            if (c < n) acc = fma(A.val ? A.val[q] : K[A.vmap[q]], x[c], acc);
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) acc += __shfl_down(acc, o, 8);
        if (sub == 0) y[row] = acc;
    }
}
void launch_P_spmv(const SpmvDev& A, const double* Kval, const double* x, double* y, int n, hipStream_t st)
{
    if (n <= 0) return;
    int g = (n + 31) / 32;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_P_spmv, dim3(g), dim3(256), 0, st, A, Kval, x, y, n);
}

// pa = P x1 and pb = P xm with xm = x / tau - x2 formed on the fly (and stored to xm_out by the row that owns it):
// the two quad_form products of kkt_solve! (kktsystem.jl:185-196) in one pass over P's rows
__global__ __launch_bounds__(256) void k_P_spmv2(SpmvDev A, const double* __restrict__ K, const double* __restrict__ x1,
                                                 const double* __restrict__ x, const double* __restrict__ x2, double tau,
                                                 double* __restrict__ pa, double* __restrict__ pb,
                                                 double* __restrict__ xm_out, double* __restrict__ pc, int n)
{
    // pc (nullable): P x2 as well -- the x2-only terms of tau_den, needed once per kkt_update! (kktsystem.jl:194-196)
    const int sub = threadIdx.x & 7;
    for (int row = blockIdx.x * 32 + (threadIdx.x >> 3); row < n; row += gridDim.x * 32) {
        double a1 = 0.0, a2 = 0.0, a3 = 0.0;
        const int64_t qend = A.pend ? A.pend[row] : A.ptr[row + 1];       // (the row's P entries are a prefix: SpmvDev::pend)
        for (int64_t q = A.ptr[row] + sub; q < qend; q += 8) {
            const int c = A.col[q];
            if (c < n) {
                const double v = A.val ? A.val[q] : K[A.vmap[q]];
                const double x2c = x2[c];
                a1 = fma(v, x1[c], a1);
                a2 = fma(v, x[c] / tau - x2c, a2);
                a3 = fma(v, x2c, a3);
            }
        }
#pragma unroll
        for (int o = 4; o > 0; o >>= 1) { a1 += __shfl_down(a1, o, 8); a2 += __shfl_down(a2, o, 8); a3 += __shfl_down(a3, o, 8); }
        if (sub == 0) {
            pa[row] = a1;
            pb[row] = a2;
            if (pc) pc[row] = a3;
            xm_out[row] = x[row] / tau - x2[row];
        }
    }
}
void launch_P_spmv2(const SpmvDev& A, const double* Kval, const double* x1, const double* x, const double* x2, double tau,
                    double* pa, double* pb, double* xm_out, double* pc, int n, hipStream_t st)
{
    if (n <= 0) return;
    int g = (n + 31) / 32;
    if (g > 2048) g = 2048;
    hipLaunchKernelGGL(k_P_spmv2, dim3(g), dim3(256), 0, st, A, Kval, x1, x, x2, tau, pa, pb, xm_out, pc, n);
}

__global__ void k_sys_axpby(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ alpha,
                            const double* __restrict__ b, const double* __restrict__ beta, double beta_host, int n)
{
    const double be = beta ? beta[0] : beta_host;
    if (alpha) {
        const double al = alpha[0];
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            out[i] = a[i] / al + be * b[i];
    } else {
        for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x)
            out[i] = a[i] + be * b[i];
    }
}
void launch_sys_axpby(double* out, const double* a, const double* alpha_div, const double* b, const double* beta,
                      double beta_host, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sys_axpby, dim3(grid_for(n, 256)), dim3(256), 0, st, out, a, alpha_div, b, beta, beta_host, n);
}

constexpr int kDotBlocks = 64;
__global__ __launch_bounds__(256) void k_dots(DotPairs P, double* __restrict__ partial)
{
    __shared__ double sh[256];
    const int p = blockIdx.y;
    const double* __restrict__ a = P.a[p];
    const double* __restrict__ b = P.b[p];
    const int n = P.len[p];
    double acc = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += kDotBlocks * 256) acc = fma(a[i], b[i], acc);
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[p * kDotBlocks + blockIdx.x] = sh[0];
}
__global__ void k_dots_finish(const double* __restrict__ partial, double* __restrict__ out, int npairs)
{
    const int p = threadIdx.x;
    if (p >= npairs) return;
    double acc = 0.0;
    for (int i = 0; i < kDotBlocks; ++i) acc += partial[p * kDotBlocks + i];
    out[p] = acc;
}
void launch_dots(const DotPairs& P, double* partial, double* out, hipStream_t st)
{
    if (P.npairs <= 0) return;
    hipLaunchKernelGGL(k_dots, dim3(kDotBlocks, P.npairs), dim3(256), 0, st, P, partial);
    hipLaunchKernelGGL(k_dots_finish, dim3(1), dim3(64), 0, st, partial, out, P.npairs);
}

__global__ void k_sys_scalars(const double* __restrict__ dots, const double* __restrict__ cached,
                              const double* __restrict__ in, double* __restrict__ out)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const double rhs_tau = in[0], rhs_kappa = in[1], tau = in[2], kappa = in[3];
    // kktsystem.jl:185-196; xi = x / tau, so xi.(P x1) = dots[2] / tau
    const double tau_num = rhs_tau - rhs_kappa / tau + dots[0] + dots[1] + 2.0 * (dots[2] / tau);
    double tau_den = kappa / tau - cached[0] - cached[1];
    tau_den += dots[3] - cached[2];
    const double dtau = tau_num / tau_den;
    out[0] = dtau;
    out[1] = -(rhs_kappa + kappa * dtau) / tau;          // :206
    out[2] = tau_num;
    out[3] = tau_den;
}
void launch_sys_scalars(const double* dots, const double* cached, const double* scal_in, double* out, hipStream_t st)
{
    hipLaunchKernelGGL(k_sys_scalars, dim3(1), dim3(64), 0, st, dots, cached, scal_in, out);
}
// (dx, dz) = (x1, z1) + dtau (x2, z2) in one launch (kktsystem.jl:200-203); dtau = scal[0] on the device
__global__ void k_sys_step(double* __restrict__ dx, double* __restrict__ dz, const double* __restrict__ x1,
                           const double* __restrict__ z1, const double* __restrict__ x2, const double* __restrict__ z2,
                           const double* __restrict__ scal, int n, int m)
{
    const double dtau = scal[0];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x) {
        if (i < n) dx[i] = x1[i] + dtau * x2[i];
        else dz[i - n] = z1[i - n] + dtau * z2[i - n];
    }
}
void launch_sys_step(double* dx, double* dz, const double* x1, const double* z1, const double* x2, const double* z2,
                     const double* scal, int n, int m, hipStream_t st)
{
    if (n + m <= 0) return;
    hipLaunchKernelGGL(k_sys_step, dim3(grid_for(n + m, 256)), dim3(256), 0, st, dx, dz, x1, z1, x2, z2, scal, n, m);
}
// The second stage of the dot products (k_dots' partial sums), the scalars of kkt_solve! and the step in ONE launch:
// every workgroup adds the partial sums up itself -- npairs x 64 doubles from L2, the same order everywhere, so every
// workgroup holds the same dtau -- instead of a one-workgroup kernel in between (~5 us on the step's chain of dependent
// launches).  Workgroup 0 stores {dtau, dkappa, tau_num, tau_den} and, with npairs = 7, the x2-only terms.
__global__ __launch_bounds__(256) void k_sys_step_scalars(double* __restrict__ dx, double* __restrict__ dz, const double* __restrict__ x1,
                                                          const double* __restrict__ z1, const double* __restrict__ x2,
                                                          const double* __restrict__ z2, const double* __restrict__ partial,
                                                          double* __restrict__ cached, double rhs_tau, double rhs_kappa, double tau,
                                                          double kappa, double* __restrict__ out, int npairs, int n, int m,
                                                          double* __restrict__ keep_x2, double* __restrict__ keep_z2)
{
    // keep_x2 / keep_z2 (nullable): (x2, z2) came out of this call's own 2-column solve and outlives it: copied on the way
    __shared__ double d[8];
    __shared__ double sh_dtau;
    const int p = threadIdx.x;
    if (p < npairs) {
        double acc = 0.0;
        for (int i = 0; i < kDotBlocks; ++i) acc += partial[p * kDotBlocks + i];      // (the order of k_dots_finish)
        d[p] = acc;
    }
    __syncthreads();
    if (p == 0) {
        // npairs = 7: pairs 4..6 are the x2-only terms {q.x2, b.z2, x2.(P x2)}, kept in `cached` for the iteration's later
        // solves; otherwise they come from there (kktsystem.jl:185-196; xi = x / tau, so xi.(P x1) = d[2] / tau)
        const double c0 = npairs == 7 ? d[4] : cached[0], c1 = npairs == 7 ? d[5] : cached[1], c2 = npairs == 7 ? d[6] : cached[2];
        const double tau_num = rhs_tau - rhs_kappa / tau + d[0] + d[1] + 2.0 * (d[2] / tau);
        double tau_den = kappa / tau - c0 - c1;
        tau_den += d[3] - c2;
        const double dtau = tau_num / tau_den;
        sh_dtau = dtau;
        if (blockIdx.x == 0) {
            if (npairs == 7) { cached[0] = c0; cached[1] = c1; cached[2] = c2; }
            out[0] = dtau;
            out[1] = -(rhs_kappa + kappa * dtau) / tau;          // :206
            out[2] = tau_num;
            out[3] = tau_den;
        }
    }
    __syncthreads();
    const double dtau = sh_dtau;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n + m; i += gridDim.x * blockDim.x) {
        if (i < n) {
            const double v = x2[i];
            dx[i] = x1[i] + dtau * v;
            if (keep_x2) keep_x2[i] = v;
        } else {
            const double v = z2[i - n];
            dz[i - n] = z1[i - n] + dtau * v;
            if (keep_z2) keep_z2[i - n] = v;
        }
    }
}
void launch_dots_sys_step(const DotPairs& P, double* partial, double* cached, double rhs_tau, double rhs_kappa, double tau,
                          double kappa, double* out, double* dx, double* dz, const double* x1, const double* z1, const double* x2,
                          const double* z2, int n, int m, hipStream_t st, double* keep_x2, double* keep_z2)
{
    hipLaunchKernelGGL(k_dots, dim3(kDotBlocks, P.npairs), dim3(256), 0, st, P, partial);
    hipLaunchKernelGGL(k_sys_step_scalars, dim3(grid_for(std::max(n + m, 1), 256)), dim3(256), 0, st, dx, dz, x1, z1, x2, z2,
                       (const double*)partial, cached, rhs_tau, rhs_kappa, tau, kappa, out, P.npairs, n, m, keep_x2, keep_z2);
}

__global__ void k_neg_sum(double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = -(a[i] + b[i]);
}
void launch_neg_sum(double* y, const double* a, const double* b, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_neg_sum, dim3(grid_for(n, 256)), dim3(256), 0, st, y, a, b, n);
}
__global__ void k_neg_copy(double* __restrict__ y, const double* __restrict__ a, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = -a[i];
}
void launch_neg_copy(double* y, const double* a, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_neg_copy, dim3(grid_for(n, 256)), dim3(256), 0, st, y, a, n);
}

}  // namespace hipkkt


// =====================================================================================
//  Ruiz equilibration (problemdata.jl:133-221).  Infinity norms are order-independent, so the
//  column / row maxima use atomicMax on the bit pattern of the non-negative doubles (exact and
//  deterministic); the one sum (mean column norm of P) is reduced in a fixed two-stage order.
// =====================================================================================
namespace hipkkt {

__device__ inline void atomic_max_nonneg(double* addr, double v)
{
    atomicMax(reinterpret_cast<unsigned long long*>(addr), (unsigned long long)__double_as_longlong(v));
}

__global__ void k_equil_norms(EquilDev E)
{
    // kkt_col_norms! (mathutils.jl:129-141): dwork = column norms of [P A'] (P symmetric from its triangle),
    // ework = row norms of A; both zeroed by the caller
    const int64_t tot = E.nnzP + E.nnzA;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < tot; j += (int64_t)gridDim.x * blockDim.x) {
        if (j < E.nnzP) {
            const double v = fabs(E.Pval[j]);
            atomic_max_nonneg(E.dwork + E.Pcol[j], v);
            atomic_max_nonneg(E.dwork + E.Prow[j], v);
        } else {
            const int64_t k = j - E.nnzP;
            const double v = fabs(E.Aval[k]);
            atomic_max_nonneg(E.dwork + E.Acol[k], v);
            atomic_max_nonneg(E.ework + E.Arow[k], v);
        }
    }
}
__global__ void k_equil_scalings(EquilDev E, double smin, double smax)
{
    // problemdata.jl:167-177
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E.n + E.m; i += gridDim.x * blockDim.x) {
        double* w = i < E.n ? E.dwork + i : E.ework + (i - E.n);
        double* cum = i < E.n ? E.d + i : E.e + (i - E.n);
        double v = *w;
        if (v == 0.0) v = 1.0;
        v = 1.0 / sqrt(v);
        const double lo = smin / *cum, hi = smax / *cum;
        v = v < lo ? lo : (v > hi ? hi : v);
        *w = v;
        *cum *= v;                                  // :182-183
    }
}
__global__ void k_equil_scale_data(EquilDev E, int with_d)
{
    // scale_data! (problemdata.jl:223-242)
    const int64_t tot = (with_d ? E.nnzP : 0) + E.nnzA;
    const int64_t pn = with_d ? E.nnzP : 0;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < tot; j += (int64_t)gridDim.x * blockDim.x) {
        if (j < pn) {
            E.Pval[j] *= E.dwork[E.Prow[j]] * E.dwork[E.Pcol[j]];
        } else {
            const int64_t k = j - pn;
            if (with_d) E.Aval[k] *= E.ework[E.Arow[k]] * E.dwork[E.Acol[k]];
            else E.Aval[k] *= E.ework[E.Arow[k]];
        }
    }
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E.n + E.m; i += gridDim.x * blockDim.x) {
        if (i < E.n) { if (with_d) E.q[i] *= E.dwork[i]; }
        else E.b[i - E.n] *= E.ework[i - E.n];
    }
}
__global__ void k_equil_colnorm_P(EquilDev E)
{
    // col_norms!(dwork, P) on the stored triangle only (problemdata.jl:188; mathutils.jl:143-165); dwork zeroed
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < E.nnzP; j += (int64_t)gridDim.x * blockDim.x)
        atomic_max_nonneg(E.dwork + E.Pcol[j], fabs(E.Pval[j]));
}
__global__ __launch_bounds__(256) void k_equil_cost_partials(EquilDev E)
{
    // partial[0..255] = block sums of dwork, partial[256..511] = block maxima of |q|
    __shared__ double sh[256];
    double acc = 0.0, mx = 0.0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < E.n; i += gridDim.x * 256) {
        acc += E.dwork[i];
        mx = fmax(mx, fabs(E.q[i]));
    }
    sh[threadIdx.x] = acc;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] += sh[threadIdx.x + o]; __syncthreads(); }
    if (threadIdx.x == 0) E.partial[blockIdx.x] = sh[0];
    __syncthreads();
    sh[threadIdx.x] = mx;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) { if ((int)threadIdx.x < o) sh[threadIdx.x] = fmax(sh[threadIdx.x], sh[threadIdx.x + o]); __syncthreads(); }
    if (threadIdx.x == 0) E.partial[256 + blockIdx.x] = sh[0];
}
__global__ void k_equil_cost_scalar(EquilDev E, int nblocks, double smin, double smax)
{
    // problemdata.jl:189-203: ctmp = clip(1 / max(||q||_inf, mean col norm of P), ...), 1 if either is zero
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double sum = 0.0, mx = 0.0;
    for (int i = 0; i < nblocks; ++i) { sum += E.partial[i]; mx = fmax(mx, E.partial[256 + i]); }
    const double mean = E.n > 0 ? sum / (double)E.n : 0.0;
    double ctmp = 1.0;
    if (mean != 0.0 && mx != 0.0) {
        const double c = E.scal[0];
        ctmp = 1.0 / fmax(mx, mean);
        const double lo = smin / c, hi = smax / c;
        ctmp = ctmp < lo ? lo : (ctmp > hi ? hi : ctmp);
        E.scal[0] = c * ctmp;
    }
    E.scal[1] = ctmp;
}
__global__ void k_equil_apply_cost(EquilDev E)
{
    const double ctmp = E.scal[1];
    if (ctmp == 1.0) return;
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < E.nnzP + E.n; j += (int64_t)gridDim.x * blockDim.x) {
        if (j < E.nnzP) E.Pval[j] *= ctmp;
        else E.q[j - E.nnzP] *= ctmp;
    }
}
void launch_equil_round(const EquilDev& E, double smin, double smax, hipStream_t st)
{
    const int64_t tot = E.nnzP + E.nnzA;
    (void)hipMemsetAsync(E.dwork, 0, (size_t)std::max(E.n, 1) * sizeof(double), st);
    (void)hipMemsetAsync(E.ework, 0, (size_t)std::max(E.m, 1) * sizeof(double), st);
    if (tot > 0) hipLaunchKernelGGL(k_equil_norms, dim3(grid_for(tot, 256)), dim3(256), 0, st, E);
    if (E.n + E.m > 0) hipLaunchKernelGGL(k_equil_scalings, dim3(grid_for(E.n + E.m, 256)), dim3(256), 0, st, E, smin, smax);
    hipLaunchKernelGGL(k_equil_scale_data, dim3(grid_for(std::max<int64_t>(tot, E.n + E.m), 256)), dim3(256), 0, st, E, 1);
    (void)hipMemsetAsync(E.dwork, 0, (size_t)std::max(E.n, 1) * sizeof(double), st);
    if (E.nnzP > 0) hipLaunchKernelGGL(k_equil_colnorm_P, dim3(grid_for(E.nnzP, 256)), dim3(256), 0, st, E);
    const int nb = std::max(1, std::min(256, (E.n + 255) / 256));
    hipLaunchKernelGGL(k_equil_cost_partials, dim3(nb), dim3(256), 0, st, E);
    hipLaunchKernelGGL(k_equil_cost_scalar, dim3(1), dim3(64), 0, st, E, nb, smin, smax);
    hipLaunchKernelGGL(k_equil_apply_cost, dim3(grid_for(E.nnzP + E.n, 256)), dim3(256), 0, st, E);
}

// one wave per cone that needs a scalar scaling: delta = mean(e) / e over the cone
__global__ __launch_bounds__(64) void k_equil_rectify(EquilDev E, const int* __restrict__ kind, const int* __restrict__ off,
                                                      const int* __restrict__ numel, int ncones)
{
    const int c = blockIdx.x;
    if (c >= ncones) return;
    const int k = kind[c];
    if (k == 0 || k == 1) return;                      // zero / nonnegative: elementwise scaling allowed, delta = 1
    const int o = off[c], n = numel[c];
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 64) acc += E.e[o + i];
    acc = wave_sum(acc);
    const double mean = acc / (double)n;
    for (int i = threadIdx.x; i < n; i += 64) E.ework[o + i] = mean / E.e[o + i];
}
__global__ void k_equil_apply_e(EquilDev E)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < E.m; i += gridDim.x * blockDim.x) E.e[i] *= E.ework[i];
}
__global__ void k_fill(double* __restrict__ p, double v, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) p[i] = v;
}
void launch_equil_rectify(const EquilDev& E, const int* cone_kind, const int* cone_off, const int* cone_numel,
                          const int* elem_cone, int ncones, hipStream_t st)
{
    (void)elem_cone;
    if (E.m <= 0 || ncones <= 0) return;
    hipLaunchKernelGGL(k_fill, dim3(grid_for(E.m, 256)), dim3(256), 0, st, E.ework, 1.0, E.m);
    hipLaunchKernelGGL(k_equil_rectify, dim3(ncones), dim3(64), 0, st, E, cone_kind, cone_off, cone_numel, ncones);
    hipLaunchKernelGGL(k_equil_scale_data, dim3(grid_for(std::max<int64_t>(E.nnzA, E.n + E.m), 256)), dim3(256), 0, st, E, 0);
    hipLaunchKernelGGL(k_equil_apply_e, dim3(grid_for(E.m, 256)), dim3(256), 0, st, E);
}

__global__ void k_lrscale(double* __restrict__ v, const int* __restrict__ row, const int* __restrict__ col, int64_t nnz,
                          const double* __restrict__ L, const double* __restrict__ R, double cscale)
{
    for (int64_t j = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; j < nnz; j += (int64_t)gridDim.x * blockDim.x)
        v[j] = v[j] * (L[row[j]] * R[col[j]]) * cscale;
}
void launch_lrscale(double* values, const int* row, const int* col, int64_t nnz, const double* L, const double* R,
                    double cscale, hipStream_t st)
{
    if (nnz <= 0) return;
    hipLaunchKernelGGL(k_lrscale, dim3(grid_for(nnz, 256)), dim3(256), 0, st, values, row, col, nnz, L, R, cscale);
}

}  // namespace hipkkt


// =====================================================================================
//  Row-major multi-column helpers (N x KP): setrhs!/getlhs!, residual with norms, accept
// =====================================================================================
namespace hipkkt {

__global__ void k_pack_rhs_rm(double* __restrict__ B, const double* __restrict__ rx, const double* __restrict__ rz, int n,
                              int m, int p, int nrhs, int KP)
{
    // thread = (row i, group of 8 columns): column reads coalesced over i, one 64-byte write per thread
    const int N = n + m + p, groups = KP >> 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)N * groups;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % N), g = (int)(idx / N);
        double v[8];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = 8 * g + c;
            v[c] = (col < nrhs) ? (i < n ? rx[(int64_t)col * n + i] : (i < n + m ? rz[(int64_t)col * m + (i - n)] : 0.0)) : 0.0;
        }
        double2* dst = reinterpret_cast<double2*>(B + (int64_t)i * KP + 8 * g);
#pragma unroll
        for (int c = 0; c < 4; ++c) dst[c] = make_double2(v[2 * c], v[2 * c + 1]);
    }
}
void launch_pack_rhs_rm(double* B, const double* rx, const double* rz, int n, int m, int p, int nrhs, int KP, hipStream_t st)
{
    const int64_t work = (int64_t)(n + m + p) * (KP >> 3);
    hipLaunchKernelGGL(k_pack_rhs_rm, dim3(grid_for(work, 256, 8192)), dim3(256), 0, st, B, rx, rz, n, m, p, nrhs, KP);
}
__global__ void k_unpack_lhs_rm(double* __restrict__ lhsx, double* __restrict__ lhsz, const double* __restrict__ X, int n,
                                int m, int nrhs, int KP)
{
    const int rows = n + m, groups = KP >> 3;
    for (int64_t idx = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; idx < (int64_t)rows * groups;
         idx += (int64_t)gridDim.x * blockDim.x) {
        const int i = (int)(idx % rows), g = (int)(idx / rows);
        const double2* src = reinterpret_cast<const double2*>(X + (int64_t)i * KP + 8 * g);
        double v[8];
#pragma unroll
        for (int c = 0; c < 4; ++c) { const double2 t = src[c]; v[2 * c] = t.x; v[2 * c + 1] = t.y; }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int col = 8 * g + c;
            if (col >= nrhs) continue;
            if (i < n) { if (lhsx) lhsx[(int64_t)col * n + i] = v[c]; }
            else if (lhsz) lhsz[(int64_t)col * m + (i - n)] = v[c];
        }
    }
}
void launch_unpack_lhs_rm(double* lhsx, double* lhsz, const double* X, int n, int m, int nrhs, int KP, hipStream_t st)
{
    if (n + m <= 0 || (!lhsx && !lhsz)) return;
    const int64_t work = (int64_t)(n + m) * (KP >> 3);
    hipLaunchKernelGGL(k_unpack_lhs_rm, dim3(grid_for(work, 256, 8192)), dim3(256), 0, st, lhsx, lhsz, X, n, m, nrhs, KP);
}

// workgroup = 16 rows x 16 columns per step; the row's matrix entries are the same for its 16 lanes, the gathered
// x rows are 128 contiguous bytes
constexpr int kRmBlocks = 512;
template <int V>                 // columns per lane: 1, or 2 (16-byte accesses, KP a multiple of 32; r03, 512 columns: 2.2 -> see DESIGN.md)
__global__ __launch_bounds__(256) void k_residual_rm(SpmvDev A, const double* __restrict__ B, const double* __restrict__ X,
                                                     double* __restrict__ E, double* __restrict__ partial,
                                                     double* __restrict__ bpartial, int KP)
{
    typedef double vec_t __attribute__((ext_vector_type(V)));
    __shared__ double sh[2][16][16 * V + 1];
    const int c = threadIdx.x & 15, r = threadIdx.x >> 4;
    const int cb0 = blockIdx.y * 16 * V + c * V;
    double vmax[V], bmax[V];
#pragma unroll
    for (int v = 0; v < V; ++v) vmax[v] = bmax[v] = 0.0;
    for (int row = blockIdx.x * 16 + r; row < A.N; row += gridDim.x * 16) {
        const int64_t q0 = A.ptr[row], q1 = A.ptr[row + 1];
        const vec_t bv = *reinterpret_cast<const vec_t*>(B + (int64_t)row * KP + cb0);
        vec_t acc = 0.0;
        // eight entries in flight per round (indices and values, then the eight rows of X), summed in row order:
        // one entry per round left the gather latency-bound (r03, 512 columns: 2.9 ms per residual; four per round 2.25)
        constexpr int RU = 8;
        for (int64_t q = q0; q < q1; q += RU) {
            int cj[RU];
            double vj[RU];
            vec_t xj[RU];
#pragma unroll
            for (int u = 0; u < RU; ++u) {
                const bool ok = q + u < q1;
                cj[u] = ok ? A.col[q + u] : -1;
                vj[u] = ok ? A.val[q + u] : 0.0;
            }
#pragma unroll
            for (int u = 0; u < RU; ++u) xj[u] = cj[u] >= 0 ? *reinterpret_cast<const vec_t*>(X + (int64_t)cj[u] * KP + cb0) : (vec_t)0.0;
#pragma unroll
            for (int u = 0; u < RU; ++u)
                if (cj[u] >= 0) {
#pragma unroll
                    for (int v = 0; v < V; ++v) acc[v] = fma(vj[u], xj[u][v], acc[v]);
                }
        }
        const vec_t rr = bv - acc;
        *reinterpret_cast<vec_t*>(E + (int64_t)row * KP + cb0) = rr;
#pragma unroll
        for (int v = 0; v < V; ++v) {
            vmax[v] = isfinite(rr[v]) ? fmax(vmax[v], fabs(rr[v])) : INFINITY;
            bmax[v] = isfinite(bv[v]) ? fmax(bmax[v], fabs(bv[v])) : INFINITY;
        }
    }
#pragma unroll
    for (int v = 0; v < V; ++v) {
        sh[0][r][c * V + v] = vmax[v];
        sh[1][r][c * V + v] = bmax[v];
    }
    __syncthreads();
    if (r == 0) {
#pragma unroll
        for (int v = 0; v < V; ++v) {
            double m = 0.0, w = 0.0;
#pragma unroll
            for (int k = 0; k < 16; ++k) { m = fmax(m, sh[0][k][c * V + v]); w = fmax(w, sh[1][k][c * V + v]); }
            partial[(int64_t)blockIdx.x * KP + cb0 + v] = m;
            if (bpartial) bpartial[(int64_t)blockIdx.x * KP + cb0 + v] = w;
        }
    }
}
// one workgroup per 64 columns, 16 threads per column over the partial rows (the maximum is order-independent); a single
// thread per column walking all 512 partial rows took 140-230 us of pure latency per residual
__global__ __launch_bounds__(1024) void k_finish_norm_rm(const double* __restrict__ partial, const double* __restrict__ bpartial,
                                                         int nblocks, int KP, double* __restrict__ out, double* __restrict__ bout)
{
    __shared__ double sh[2][16][64];
    const int cl = threadIdx.x & 63, part = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    double v = 0.0, w = 0.0;
    if (c < KP)
        for (int i = part; i < nblocks; i += 16) {
            v = fmax(v, partial[(int64_t)i * KP + c]);
            if (bpartial) w = fmax(w, bpartial[(int64_t)i * KP + c]);
        }
    sh[0][part][cl] = v;
    sh[1][part][cl] = w;
    __syncthreads();
    if (part == 0 && c < KP) {
        for (int k = 1; k < 16; ++k) { v = fmax(v, sh[0][k][cl]); w = fmax(w, sh[1][k][cl]); }
        out[c] = v;
        if (bpartial) bout[c] = w;
    }
}
void launch_residual_rm(const SpmvDev& A, const double* B, const double* X, double* E, double* partial, double* norm_out,
                        double* normb_out, int KP, hipStream_t st)
{
    int g = (A.N + 15) / 16;
    if (g > kRmBlocks) g = kRmBlocks;
    if (g < 1) g = 1;
    double* bpartial = normb_out ? partial + (size_t)kRmBlocks * KP : nullptr;
    const int vmax = knobs().multi_vec;
    if (KP % 64 == 0 && vmax >= 4) hipLaunchKernelGGL(k_residual_rm<4>, dim3(g, KP / 64), dim3(256), 0, st, A, B, X, E, partial, bpartial, KP);
    else if (KP % 32 == 0) hipLaunchKernelGGL(k_residual_rm<2>, dim3(g, KP / 32), dim3(256), 0, st, A, B, X, E, partial, bpartial, KP);
    else hipLaunchKernelGGL(k_residual_rm<1>, dim3(g, KP / 16), dim3(256), 0, st, A, B, X, E, partial, bpartial, KP);
    hipLaunchKernelGGL(k_finish_norm_rm, dim3((KP + 63) / 64), dim3(1024), 0, st, (const double*)partial, (const double*)bpartial, g,
                       KP, norm_out, normb_out);
}
__global__ void k_accept_columns_rm(double* __restrict__ X, const double* __restrict__ cand, double* __restrict__ E,
                                    const double* __restrict__ E2, const int* __restrict__ mask, int64_t total, int KP)
{
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        if (mask[i % KP]) { X[i] = cand[i]; E[i] = E2[i]; }
    }
}
void launch_accept_columns_rm(double* X, const double* cand, double* E, const double* E2, const int* mask, int N, int KP,
                              hipStream_t st)
{
    const int64_t total = (int64_t)N * KP;
    if (total <= 0) return;
    hipLaunchKernelGGL(k_accept_columns_rm, dim3(grid_for(total, 256, 8192)), dim3(256), 0, st, X, cand, E, E2, mask, total, KP);
}

}  // namespace hipkkt
