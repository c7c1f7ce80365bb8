// See kernels.hpp.  Hand-written HIP for gfx950 (CDNA4, wave64).
#include "kernels.hpp"

#include <cmath>

namespace hipkkt {

// =====================================================================================
//  Numeric LDL^T: one workgroup per supernode, levels of the assembly tree launched in order.
//  Replaces QDLDL.refactor! (call site /root/reference/src/kktsolvers/direct-ldl/
//  directldl_qdldl.jl:72-81): same pivot rule -- D_k*sign_k < eps  =>  D_k = sign_k*delta --
//  applied at pivot time inside the dense panel, no pivoting, static structure.
//
//  Per front:  zero panel -> scatter K (+ static eps*sign on the diagonal) -> extend-add the
//  children's update blocks that land in the panel -> right-looking blocked LDL^T with the
//  current block column staged in LDS -> update block U = -L21 D L21^T -> extend-add the
//  children's pass-through part into U.  Children are added one after another in a fixed order,
//  each by an injective map, so the sums are reproducible run to run (no atomics).
// =====================================================================================
template <int BS>
__global__ __launch_bounds__(BS) void k_factor(FactorArgs A, int begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const TreeDev& T = A.T;
    const int s = T.sched[begin + blockIdx.x];
    const int c0 = T.sn_start[s];
    const int nc = T.sn_start[s + 1] - c0;
    const int nb = (int)(T.rowptr[s + 1] - T.rowptr[s]);
    const int f = nc + nb;
    double* __restrict__ F = A.fronts + T.front_off[s];
    double* __restrict__ U = A.upd + T.upd_off[s];

    double* sh_d = smem;                 // kMaxNbk
    double* sh_dinv = smem + kMaxNbk;    // kMaxNbk
    double* Bl = smem + 2 * kMaxNbk;     // ldB x nbk block column

    // ---- 1. zero the panel
    for (int i = tid; i < f * nc; i += BS) F[i] = 0.0;
    __syncthreads();
    // ---- 2. scatter the K entries of these columns, then the static regulariser
    for (int64_t e = T.kptr[s] + tid; e < T.kptr[s + 1]; e += BS) F[T.kdst[e]] = A.Kval[T.ksrc[e]];
    __syncthreads();
    if (A.eps) {
        const double eps = *A.eps;
        for (int k = tid; k < nc; k += BS) F[k + (int64_t)k * f] += eps * (double)T.psign[c0 + k];
    }
    // ---- 3. children: entries whose column lands in this supernode's columns
    for (int ce = T.child_ptr[s]; ce < T.child_ptr[s + 1]; ++ce) {
        const int c = T.child_idx[ce];
        const int64_t crp = T.rowptr[c];
        const int nbc = (int)(T.rowptr[c + 1] - crp);
        const int kc = T.ncolpar[c];
        const double* __restrict__ Uc = A.upd + T.upd_off[c];
        const int* __restrict__ relc = T.rel + crp;
        __syncthreads();
        for (int idx = tid; idx < kc * nbc; idx += BS) {
            const int b = idx / nbc, a = idx - b * nbc;
            if (a >= b) F[relc[a] + (int64_t)relc[b] * f] += Uc[a + (int64_t)b * nbc];
        }
    }
    __syncthreads();

    // ---- 4. blocked right-looking factorisation, trailing update over panel and U
    const int nbk = A.nbk;
    for (int kb = 0; kb < nc; kb += nbk) {
        const int w = min(nbk, nc - kb);
        const int R = f - kb;
        const int ldB = (R + 7) & ~3;        // >= R + 4: tiles may read up to 3 rows past R
        for (int idx = tid; idx < ldB * w; idx += BS) {
            const int j = idx / ldB, i = idx - j * ldB;
            Bl[idx] = (i < R && i >= j) ? F[(kb + i) + (int64_t)(kb + j) * f] : 0.0;
        }
        for (int k = 0; k < w; ++k) {
            __syncthreads();
            double d = Bl[k + k * ldB];
            const double sg = (double)T.psign[c0 + kb + k];
            const bool reg = (d * sg < A.dyn_eps);
            if (reg) d = sg * A.dyn_delta;
            const double dinv = 1.0 / d;
            if (tid == 0) {
                sh_d[k] = d;
                sh_dinv[k] = dinv;
                if (reg) atomicAdd(&A.flags[0], 1);
                if (!isfinite(dinv)) A.flags[1] = 1;
            }
            const double* __restrict__ colk = Bl + k * ldB;
            for (int i = k + 1 + tid; i < R; i += BS) {
                const double lik = colk[i] * dinv;
                const int jmax = min(i, w - 1);
                for (int j = k + 1; j <= jmax; ++j) Bl[i + j * ldB] -= lik * colk[j];
            }
        }
        __syncthreads();
        // scale to L, write L and D back, keep L in LDS for the trailing update
        for (int idx = tid; idx < ldB * w; idx += BS) {
            const int j = idx / ldB, i = idx - j * ldB;
            if (i >= R) continue;
            if (i > j) {
                const double l = Bl[idx] * sh_dinv[j];
                Bl[idx] = l;
                F[(kb + i) + (int64_t)(kb + j) * f] = l;
            } else if (i == j) {
                F[(kb + i) + (int64_t)(kb + j) * f] = sh_d[j];
                A.Dinv[c0 + kb + j] = sh_dinv[j];
            }
        }
        __syncthreads();
        // trailing update: C(i,j) -= sum_k L(i,k) d_k L(j,k), i >= j, over columns kb+w .. f
        const int Tn = f - kb - w;
        if (Tn > 0) {
            const int ntile = (Tn + 3) >> 2;
            const int nt = ntile * (ntile + 1) / 2;
            const int g0 = kb + w;           // global front index of local 0
            for (int t = tid; t < nt; t += BS) {
                int tr = (int)((sqrt(8.0 * (double)t + 1.0) - 1.0) * 0.5);
                while (tr * (tr + 1) / 2 > t) --tr;
                while ((tr + 1) * (tr + 2) / 2 <= t) ++tr;
                const int tc = t - tr * (tr + 1) / 2;
                const double* __restrict__ Ar = Bl + w + 4 * tr;
                const double* __restrict__ Bc = Bl + w + 4 * tc;
                double acc[4][4];
#pragma unroll
                for (int a = 0; a < 4; ++a)
#pragma unroll
                    for (int b = 0; b < 4; ++b) acc[a][b] = 0.0;
                for (int k = 0; k < w; ++k) {
                    const double dk = sh_d[k];
                    double av[4], bv[4];
#pragma unroll
                    for (int a = 0; a < 4; ++a) av[a] = Ar[k * ldB + a];
#pragma unroll
                    for (int b = 0; b < 4; ++b) bv[b] = Bc[k * ldB + b] * dk;
#pragma unroll
                    for (int a = 0; a < 4; ++a)
#pragma unroll
                        for (int b = 0; b < 4; ++b) acc[a][b] = fma(av[a], bv[b], acc[a][b]);
                }
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const int j = 4 * tc + b;
                    if (j >= Tn) continue;
                    const int gj = g0 + j;
#pragma unroll
                    for (int a = 0; a < 4; ++a) {
                        const int i = 4 * tr + a;
                        if (i >= Tn || i < j) continue;
                        const int gi = g0 + i;
                        if (gj < nc) {
                            F[gi + (int64_t)gj * f] -= acc[a][b];
                        } else {
                            double* u = U + (gi - nc) + (int64_t)(gj - nc) * nb;
                            *u = (kb == 0) ? -acc[a][b] : (*u - acc[a][b]);
                        }
                    }
                }
            }
        }
        __syncthreads();
    }
    // ---- 6. children: pass-through part into the update block
    for (int ce = T.child_ptr[s]; ce < T.child_ptr[s + 1]; ++ce) {
        const int c = T.child_idx[ce];
        const int64_t crp = T.rowptr[c];
        const int nbc = (int)(T.rowptr[c + 1] - crp);
        const int kc = T.ncolpar[c];
        const int wc = nbc - kc;
        const double* __restrict__ Uc = A.upd + T.upd_off[c];
        const int* __restrict__ relc = T.rel + crp;
        __syncthreads();
        for (int idx = tid; idx < wc * wc; idx += BS) {
            const int bb = idx / wc, aa = idx - bb * wc;
            if (aa < bb) continue;
            const int a = kc + aa, b = kc + bb;
            U[(relc[a] - nc) + (int64_t)(relc[b] - nc) * nb] += Uc[a + (int64_t)b * nbc];
        }
    }
}

// dynamic LDS above 64 KiB has to be asked for once per kernel
template <class K>
static void allow_big_lds(K kernel)
{
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              160 * 1024);
}
static void init_lds_limits()
{
    static bool done = false;
    if (done) return;
    done = true;
    allow_big_lds(k_factor<64>);
    allow_big_lds(k_factor<256>);
}

void launch_factor(const FactorArgs& a, int begin, int count, int bs, size_t lds, hipStream_t st)
{
    if (count <= 0) return;
    init_lds_limits();
    if (bs == 64) hipLaunchKernelGGL(k_factor<64>, dim3(count), dim3(64), lds, st, a, begin);
    else hipLaunchKernelGGL(k_factor<256>, dim3(count), dim3(256), lds, st, a, begin);
}

// =====================================================================================
//  Triangular solves on the supernodal tree.  Replaces QDLDL.solve! (directldl_qdldl.jl:85-96):
//  permute, L \, D^{-1}, L' \, inverse permute.
//  Forward: multifrontal style -- each front gathers b and its children's contribution vectors,
//  solves its unit-lower diagonal block, and leaves  -L21*y  (plus what passed through) for its
//  parent: no scatter conflicts, fixed summation order.  Backward: each front gathers the
//  ancestors' solution entries it needs.
// =====================================================================================
template <int BS>
__global__ __launch_bounds__(BS) void k_fwd(SolveArgs A, int begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const TreeDev& T = A.T;
    const int s = T.sched[begin + blockIdx.x];
    const int c0 = T.sn_start[s];
    const int nc = T.sn_start[s + 1] - c0;
    const int64_t rp = T.rowptr[s];
    const int nb = (int)(T.rowptr[s + 1] - rp);
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + T.front_off[s];
    double* y = smem;                          // f
    double* Ld = smem + ((f + 1) & ~1);        // kTriBlock x (kTriBlock+1)
    constexpr int ldd = kTriBlock + 1;

    for (int i = tid; i < f; i += BS) y[i] = (i < nc) ? A.b[T.perm[c0 + i]] : 0.0;
    for (int ce = T.child_ptr[s]; ce < T.child_ptr[s + 1]; ++ce) {
        const int c = T.child_idx[ce];
        const int64_t crp = T.rowptr[c];
        const int nbc = (int)(T.rowptr[c + 1] - crp);
        __syncthreads();
        for (int t = tid; t < nbc; t += BS) y[T.rel[crp + t]] += A.uvec[crp + t];
    }
    __syncthreads();
    for (int kb = 0; kb < nc; kb += kTriBlock) {
        const int w = min(kTriBlock, nc - kb);
        for (int idx = tid; idx < w * w; idx += BS) {
            const int j = idx / w, i = idx - j * w;
            Ld[i + j * ldd] = (i > j) ? F[(kb + i) + (int64_t)(kb + j) * f] : 0.0;
        }
        __syncthreads();
        for (int k = 0; k < w - 1; ++k) {
            const double yk = y[kb + k];
            for (int i = k + 1 + tid; i < w; i += BS) y[kb + i] -= Ld[i + k * ldd] * yk;
            __syncthreads();
        }
        // rows below the block: y_i -= sum_k L(i,k) y_k
        for (int i = kb + w + tid; i < f; i += BS) {
            double acc = 0.0;
            for (int k = 0; k < w; ++k) acc = fma(F[i + (int64_t)(kb + k) * f], y[kb + k], acc);
            y[i] -= acc;
        }
        __syncthreads();
    }
    for (int i = tid; i < nc; i += BS) A.xp[c0 + i] = y[i];
    for (int t = tid; t < nb; t += BS) A.uvec[rp + t] = y[nc + t];
}

template <int BS>
__global__ __launch_bounds__(BS) void k_bwd(SolveArgs A, int begin)
{
    extern __shared__ __attribute__((aligned(16))) double smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int NW = BS / 64;
    const TreeDev& T = A.T;
    const int s = T.sched[begin + blockIdx.x];
    const int c0 = T.sn_start[s];
    const int nc = T.sn_start[s + 1] - c0;
    const int64_t rp = T.rowptr[s];
    const int nb = (int)(T.rowptr[s + 1] - rp);
    const int f = nc + nb;
    const double* __restrict__ F = A.fronts + T.front_off[s];
    double* y = smem;
    double* Ld = smem + ((f + 1) & ~1);
    constexpr int ldd = kTriBlock + 1;

    for (int i = tid; i < f; i += BS)
        y[i] = (i < nc) ? A.xp[c0 + i] * A.Dinv[c0 + i] : A.xp[T.rows[rp + i - nc]];
    __syncthreads();
    const int nblk = (nc + kTriBlock - 1) / kTriBlock;
    for (int bi = nblk - 1; bi >= 0; --bi) {
        const int kb = bi * kTriBlock;
        const int w = min(kTriBlock, nc - kb);
        // y_k -= sum_{i >= kb+w} L(i,k) y_i : one wave per column, lanes over rows
        for (int k = wave; k < w; k += NW) {
            const double* __restrict__ col = F + (int64_t)(kb + k) * f;
            double acc = 0.0;
            for (int i = kb + w + lane; i < f; i += 64) acc = fma(col[i], y[i], acc);
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
            if (lane == 0) y[kb + k] -= acc;
        }
        for (int idx = tid; idx < w * w; idx += BS) {
            const int j = idx / w, i = idx - j * w;
            Ld[i + j * ldd] = (i > j) ? F[(kb + i) + (int64_t)(kb + j) * f] : 0.0;
        }
        __syncthreads();
        for (int k = w - 1; k > 0; --k) {
            const double yk = y[kb + k];
            for (int j = tid; j < k; j += BS) y[kb + j] -= Ld[k + j * ldd] * yk;
            __syncthreads();
        }
    }
    for (int i = tid; i < nc; i += BS) {
        const double v = y[i];
        A.xp[c0 + i] = v;
        A.out[T.perm[c0 + i]] = v;
    }
}

static void init_lds_limits_solve()
{
    static bool done = false;
    if (done) return;
    done = true;
    allow_big_lds(k_fwd<64>);
    allow_big_lds(k_fwd<256>);
    allow_big_lds(k_bwd<64>);
    allow_big_lds(k_bwd<256>);
}
void launch_fwd(const SolveArgs& a, int begin, int count, int bs, size_t lds, hipStream_t st)
{
    if (count <= 0) return;
    init_lds_limits_solve();
    if (bs == 64) hipLaunchKernelGGL(k_fwd<64>, dim3(count), dim3(64), lds, st, a, begin);
    else hipLaunchKernelGGL(k_fwd<256>, dim3(count), dim3(256), lds, st, a, begin);
}
void launch_bwd(const SolveArgs& a, int begin, int count, int bs, size_t lds, hipStream_t st)
{
    if (count <= 0) return;
    init_lds_limits_solve();
    if (bs == 64) hipLaunchKernelGGL(k_bwd<64>, dim3(count), dim3(64), lds, st, a, begin);
    else hipLaunchKernelGGL(k_bwd<256>, dim3(count), dim3(256), lds, st, a, begin);
}

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
template <int G>
__global__ __launch_bounds__(256) void k_residual(SpmvDev A, const double* __restrict__ K,
                                                  const double* __restrict__ b, const double* __restrict__ x,
                                                  double* __restrict__ e, double* __restrict__ partial)
{
    __shared__ double sh[4];
    const int sub = threadIdx.x % G;
    const int rows_per_block = 256 / G;
    double vmax = 0.0;
    bool bad = false;
    for (int row = blockIdx.x * rows_per_block + threadIdx.x / G; row < A.N; row += gridDim.x * rows_per_block) {
        double acc = 0.0;
        for (int64_t q = A.ptr[row] + sub; q < A.ptr[row + 1]; q += G) acc = fma(K[A.vmap[q]], x[A.col[q]], acc);
#pragma unroll
        for (int o = G / 2; o > 0; o >>= 1) acc += __shfl_down(acc, o, G);
        if (sub == 0) {
            const double r = b[row] - acc;
            e[row] = r;
            if (!isfinite(r)) bad = true;
            vmax = fmax(vmax, fabs(r));
        }
    }
    if (bad) vmax = INFINITY;        // marks non-finite; finished as NaN below
    vmax = block_max_256(vmax, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = vmax;
}
__global__ void k_finish_norm(const double* __restrict__ partial, int nparts, double* __restrict__ out)
{
    __shared__ double sh[4];
    double v = 0.0;
    for (int i = threadIdx.x; i < nparts; i += blockDim.x) v = fmax(v, partial[i]);
    v = block_max_256(v, sh);
    // norm(e, Inf) of a vector holding Inf or NaN is not finite either way; the caller only
    // tests isfinite() (kktsolver_directldl.jl:411,429)
    if (threadIdx.x == 0) out[0] = v;
}
void launch_residual(const SpmvDev& A, const double* K, const double* b, const double* x, double* e,
                     double* partial, double* norm_out, hipStream_t st)
{
    int rows_per_block = 256 / A.lanes_per_row;
    int g = (A.N + rows_per_block - 1) / rows_per_block;
    if (g > 2048) g = 2048;
    if (g < 1) g = 1;
    if (A.lanes_per_row == 8) hipLaunchKernelGGL(k_residual<8>, dim3(g), dim3(256), 0, st, A, K, b, x, e, partial);
    else hipLaunchKernelGGL(k_residual<64>, dim3(g), dim3(256), 0, st, A, K, b, x, e, partial);
    hipLaunchKernelGGL(k_finish_norm, dim3(1), dim3(256), 0, st, partial, g, norm_out);
}
__global__ void k_absmax(const double* __restrict__ v, int n, double* __restrict__ partial)
{
    __shared__ double sh[4];
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
void launch_norm_inf(const double* v, int n, double* partial, double* out, hipStream_t st)
{
    int g = grid_for(n, 256, kRedBlocks);
    hipLaunchKernelGGL(k_absmax, dim3(g), dim3(256), 0, st, v, n, partial);
    hipLaunchKernelGGL(k_finish_norm, dim3(1), dim3(256), 0, st, partial, g, out);
}
__global__ void k_sum2(double* __restrict__ y, const double* __restrict__ a, const double* __restrict__ b, int n)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) y[i] = a[i] + b[i];
}
void launch_axpby_sum(double* y, const double* a, const double* b, int n, hipStream_t st)
{
    if (n <= 0) return;
    hipLaunchKernelGGL(k_sum2, dim3(grid_for(n, 256)), dim3(256), 0, st, y, a, b, n);
}
__global__ void k_pack_rhs(double* __restrict__ b, const double* __restrict__ rx, const double* __restrict__ rz,
                           int n, int m, int p)
{
    const int N = n + m + p;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < N; i += gridDim.x * blockDim.x)
        b[i] = i < n ? rx[i] : (i < n + m ? rz[i - n] : 0.0);      // kktsolver_directldl.jl:313-327
}
void launch_pack_rhs(double* b, const double* rx, const double* rz, int n, int m, int p, hipStream_t st)
{
    hipLaunchKernelGGL(k_pack_rhs, dim3(grid_for(n + m + p, 256)), dim3(256), 0, st, b, rx, rz, n, m, p);
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
//  PSD cones are scaled by the caller (hipkkt_kkt_update_cones) in this version.
// =====================================================================================
__global__ void k_cone_elementwise(ConeDev C, ConeState S, const double* __restrict__ s,
                                   const double* __restrict__ z, int m)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int c = C.elem_cone[i];
        const int kind = C.kind[c];
        if (kind == 0) {
            S.w[i] = 0.0;
            S.Hs[C.boff[c] + (i - C.off[c])] = 0.0;
        } else if (kind == 1) {
            const double w = sqrt(s[i] / z[i]);
            S.w[i] = w;
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
__global__ __launch_bounds__(64) void k_cone_soc(ConeDev C, ConeState S, const double* __restrict__ s,
                                                 const double* __restrict__ z)
{
    const int c = C.soc_list[blockIdx.x];
    const int off = C.off[c], n = C.numel[c];
    const int lane = threadIdx.x;
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
    __syncthreads();      // w[] written above is re-read below by other lanes of this wave
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

void launch_cone_scaling(const ConeDev& C, const ConeState& S, const double* s, const double* z, int m,
                         hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(k_cone_elementwise, dim3(grid_for(m, 256)), dim3(256), 0, st, C, S, s, z, m);
    if (C.nsoc > 0) hipLaunchKernelGGL(k_cone_soc, dim3(C.nsoc), dim3(64), 0, st, C, S, s, z);
}

// y = W'W x : zero -> 0, NN -> w*(w*x), SOC -> eta^2 (2 w (w'x) - J x)   (mul_Hs!)
__global__ void k_mul_Hs_elementwise(ConeDev C, ConeState S, double* __restrict__ y, const double* __restrict__ x, int m)
{
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < m; i += gridDim.x * blockDim.x) {
        const int kind = C.kind[C.elem_cone[i]];
        if (kind == 0) y[i] = 0.0;
        else if (kind == 1) y[i] = S.w[i] * (S.w[i] * x[i]);
    }
}
__global__ __launch_bounds__(64) void k_mul_Hs_soc(ConeDev C, ConeState S, double* __restrict__ y,
                                                   const double* __restrict__ x)
{
    const int c = C.soc_list[blockIdx.x];
    const int off = C.off[c], n = C.numel[c], lane = threadIdx.x;
    const double* w = S.w + off;
    double dot = 0.0;
    for (int i = lane; i < n; i += 64) dot += w[i] * x[off + i];
    dot = 2.0 * wave_sum(dot);
    const double e2 = S.eta[c] * S.eta[c];
    for (int i = lane; i < n; i += 64) {
        const double xi = x[off + i];
        y[off + i] = ((i == 0 ? -xi : xi) + dot * w[i]) * e2;
    }
}
void launch_mul_Hs(const ConeDev& C, const ConeState& S, double* y, const double* x, int m, hipStream_t st)
{
    if (m > 0) hipLaunchKernelGGL(k_mul_Hs_elementwise, dim3(grid_for(m, 256)), dim3(256), 0, st, C, S, y, x, m);
    if (C.nsoc > 0) hipLaunchKernelGGL(k_mul_Hs_soc, dim3(C.nsoc), dim3(64), 0, st, C, S, y, x);
}

}  // namespace hipkkt
