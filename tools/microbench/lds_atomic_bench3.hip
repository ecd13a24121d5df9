// lds_atomic_bench3.hip -- ds_add_u64: do duplicate addresses cost the same when the lanes that share one sit in
// DIFFERENT 16-lane rows of the wave (pattern B) as when they are neighbours (pattern A)?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
__global__ __launch_bounds__(256) void k(float *out, int iters, int pattern, int mult) {
    __shared__ unsigned long long box[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) box[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    // mult lanes share an address: A = neighbours (lane / mult), B = lanes a whole row apart (lane % (64 / mult))
    const int grp = pattern == 0 ? lane / mult : lane % (64 / mult);
    int a = ((grp * 19) + w * 1024) & 4095;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) atomicAdd(&box[(a + r * 37) & 4095], (unsigned long long)lane + 1ull);
        a = (a + 5) & 4095;
    }
    __syncthreads();
    unsigned long long s = 0; for (int i = threadIdx.x; i < 4096; i += 256) s += box[i];
    out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}
int main() {
    float *out; CK(hipMalloc(&out, 256 * 8 * 256 * 4));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUs = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048, blocks = CUs * 4;
    for (int mult : {1, 2, 4}) for (int pattern : {0, 1}) {
        auto launch = [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, pattern, mult); };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%d lanes per address, %s: %7.1f cyc/CU per ds_add_u64\n", mult, pattern ? "a row or more apart (lane %% (64/m))" : "neighbouring lanes (lane / m)",
               ms * 1e-3 * clk * CUs / ((double)blocks * 4 * iters * 8));
    }
    return 0;
}
