// FP64 peak microbenchmark for the roofline denominators ("verify with a microbenchmark on the box").
//   hipcc --offload-arch=gfx950 -O3 scripts/fp64_mfma_peak.hip -o /tmp/fp64_peak && /tmp/fp64_peak
// Every wave issues back-to-back v_mfma_f64_16x16x4_f64 on 8 independent accumulators (2 * 16*16*4 flop each),
// then the same with plain v_fma_f64 (8 independent chains per lane); prints one JSON line.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k_mfma(double* out, int iters)
{
    d4 acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_fma(double* out, int iters)
{
    double acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = i;
    double a = 1.0 + threadIdx.x * 1e-9, b = 1e-9;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 8; ++i) acc[i] = __builtin_fma(acc[i], a, b);
    double s = 0.0;
    for (int i = 0; i < 8; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
int main()
{
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int blocks = p.multiProcessorCount * 8, iters = 20000;
    double* out;
    hipMalloc(&out, (size_t)blocks * 256 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2] = {0, 0};
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 3; ++rep) {          // last repetition counts (clocks settled)
            hipEventRecord(e0);
            if (which == 0) hipLaunchKernelGGL(k_mfma, dim3(blocks), dim3(256), 0, 0, out, iters);
            else hipLaunchKernelGGL(k_fma, dim3(blocks), dim3(256), 0, 0, out, iters);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[which], e0, e1);
        }
    const double waves = (double)blocks * 4;
    const double mfma_tf = waves * iters * 8 * (2.0 * 16 * 16 * 4) / (ms[0] * 1e-3) / 1e12;
    const double fma_tf = (double)blocks * 256 * iters * 8 * 2.0 / (ms[1] * 1e-3) / 1e12;
    std::printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"fp64_mfma_TFLOPs\": %.2f, \"fp64_vector_fma_TFLOPs\": %.2f}\n",
                p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000, mfma_tf, fma_tf);
    return 0;
}
