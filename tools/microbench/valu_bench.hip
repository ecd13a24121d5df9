// valu_bench.hip -- f32 VALU issue rate on gfx950: scalar v_fma_f32 vs packed v_pk_fma_f32, by occupancy.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));

__global__ __launch_bounds__(256) void k_scalar(float *out, int iters, float b) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_fmaf(a[i], b, 0.5f);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_packed(float *out, int iters, float b) {
    f2 a[8];
    for (int i = 0; i < 8; ++i) { a[i].x = threadIdx.x * 1e-3f + i; a[i].y = a[i].x + 0.5f; }
    f2 bb = {b, b}, cc = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = __builtin_elementwise_fma(a[i], bb, cc);
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUs = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 8192;
    for (int bpc = 1; bpc <= 8; bpc *= 2) {
        for (int pk = 0; pk < 2; ++pk) {
            auto launch = [&] { if (pk) hipLaunchKernelGGL(k_packed, dim3(CUs * bpc), dim3(256), 0, 0, out, iters, 0.999f);
                                else hipLaunchKernelGGL(k_scalar, dim3(CUs * bpc), dim3(256), 0, 0, out, iters, 0.999f); };
            launch(); CK(hipDeviceSynchronize());
            CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            double winstr = (double)CUs * bpc * 4 * iters * 64;
            printf("%s %d waves/SIMD: %.3f ms, %.2f SIMD-cycles per wave-instr (@2.4GHz), %.1f TFLOP/s\n", pk ? "v_pk_fma_f32" : "v_fma_f32   ",
                   bpc, ms, ms * 1e-3 * clk * CUs * 4 / winstr, winstr * 64 * 2 * (pk ? 2 : 1) / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
