// lds_atomic_bench.hip -- throughput of ds_add_f32 (LDS float atomics) on gfx950 by address pattern.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

template <int MODE>  // 0: ds_add_f32, 1: plain read+add+write (racy), 2: ds_write only, 3: ds_add_u32, 4: ds_add_u64, 5: ds_add_f64
__global__ __launch_bounds__(256) void k(float *out, int iters, int lane_stride, int it_stride) {
    __shared__ float box[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) box[i] = 0.f;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    int a = (lane * lane_stride + w * 2048) & 8191;
    float v = 1.0f + lane;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = (a + r * 37) & 8191;
            if (MODE == 0) atomicAdd(&box[idx], v);
            else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned int *>(&box[idx]), (unsigned int)lane + 1u);
            else if (MODE == 4) atomicAdd(reinterpret_cast<unsigned long long *>(&box[idx & ~1]), (unsigned long long)lane + 1ull);
            else if (MODE == 5) atomicAdd(reinterpret_cast<double *>(&box[idx & ~1]), (double)v);
            else if (MODE == 1) box[idx] += v;
            else box[idx] = v;
        }
        a = (a + it_stride) & 8191;
    }
    __syncthreads();
    float s = 0; for (int i = threadIdx.x; i < 8192; i += 256) s += box[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUs = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048, blocks = CUs * 4;
    struct { const char *name; int ls, is; } pats[] = {{"distinct banks (stride 1)", 1, 64}, {"same address (stride 0)", 0, 1},
        {"same bank, 64 addresses (stride 32)", 32, 1}, {"2-way (stride 16)", 16, 3}, {"stride 19", 19, 7}, {"stride 361", 361, 5}, {"pairs same addr (stride 1, lane/2)", -1, 64}};
    for (auto &p : pats) for (int mode = 0; mode < 6; ++mode) {
        int ls = p.ls;
        auto launch = [&] {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is);
            else if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is);
            else if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is);
            else if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is);
            else if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is);
            else hipLaunchKernelGGL(k<5>, dim3(blocks), dim3(256), 0, 0, out, iters, ls, p.is); };
        if (ls < 0) continue;
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double winstr = (double)blocks * 4 * iters * 8;
        printf("%-40s %-12s %8.3f ms  %7.1f cyc/CU per wave-instr\n", p.name, mode == 0 ? "ds_add_f32" : mode == 1 ? "read+write" : mode == 2 ? "ds_write" : mode == 3 ? "ds_add_u32" : mode == 4 ? "ds_add_u64" : "ds_add_f64", ms,
               ms * 1e-3 * clk * CUs / winstr);
    }
    return 0;
}
