// lds_atomic_bench2.hip -- ds_add_u64 cost vs number of active lanes and duplicate-address run length.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
// lanes: `active` of 64 participate (lane % (64/active) == 0); consecutive groups of `run` lanes share an address;
// distinct addresses are `stride` elements apart.
__global__ __launch_bounds__(256) void k(float *out, int iters, int active_mod, int run, int stride) {
    __shared__ unsigned long long box[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) box[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const bool act = (lane % active_mod) == 0;
    int a = (((lane / run) * stride) + w * 1024) & 4095;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            const int idx = (a + r * 37) & 4095;
            if (act) atomicAdd(&box[idx], (unsigned long long)lane + 1ull);
        }
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
    int cfgs[][3] = {{1,1,1},{1,1,19},{1,1,361},{2,1,19},{4,1,19},{8,1,19},{1,2,19},{1,4,19},{1,8,19},{1,16,19},{4,4,19},{1,4,361},{1,4,381}};
    for (auto &c : cfgs) {
        auto launch = [&] { hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters, c[0], c[1], c[2]); };
        launch(); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double winstr = (double)blocks * 4 * iters * 8;
        printf("active 1/%d lanes, run %2d, stride %3d: %7.1f cyc/CU per ds_add_u64\n", c[0], c[1], c[2], ms * 1e-3 * clk * CUs / winstr);
    }
    return 0;
}
