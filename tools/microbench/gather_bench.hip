// gather_bench.hip -- micro-benchmark of the memory primitives the march kernels can be built from
// (design input for DESIGN.md; not part of the product library).
//   hipcc --offload-arch=gfx950 -O3 gather_bench.hip -o gather_bench && ./gather_bench
// Measures, for ray-march-like address patterns over a 512^3 f32 volume, the cost of one wave-wide load
// instruction (cycles per CU) for global dword/dwordx2/dwordx4 gathers and LDS ds_read_b32 gathers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

struct Pattern {
    int lane_base[64];   // element offset of the lane's ray origin relative to the wave's region
    int lane_s[64];      // sample index offset of the lane
    int adv;             // samples advanced per iteration (1: lane = ray, 64: lane = sample)
    int qx, qy, qz;      // step per sample in 1/1024 voxel along x, y, z
    int SX, SY;          // element strides of x and y (z stride = 1)
};

template <int VEC, int LOADS>
__global__ __launch_bounds__(256) void gather_global(const float *__restrict__ vol, Pattern P, int iters,
                                                     long wave_region, long region_mask, float *out) {
    const int lane = threadIdx.x & 63;
    const long wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const long base = (wave * wave_region) & region_mask;
    const int lb = P.lane_base[lane], ls = P.lane_s[lane];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int k = (it * P.adv + ls) & 511;  // restart the ray every 512 samples (stays inside the volume)
        const long o = base + lb + (long)((k * P.qx) >> 10) * P.SX + (long)((k * P.qy) >> 10) * P.SY + ((k * P.qz) >> 10);
        const float *p = vol + o;
#pragma unroll
        for (int t = 0; t < LOADS; ++t) {
            // taps: +-1 in x / y planes around the cell (row gathers along z)
            const float *q = p + (t & 1) * P.SX + ((t >> 1) & 1) * P.SY + (t >> 2) * 2 * P.SX;
            if (VEC == 1) acc += q[0];
            else if (VEC == 2) { float2 v; __builtin_memcpy(&v, q, 8); acc += v.x + v.y; }
            else { float4 v; __builtin_memcpy(&v, q, 16); acc += (v.x + v.y) + (v.z + v.w); }
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int LOADS>
__global__ __launch_bounds__(256) void gather_lds(const float *__restrict__ vol, Pattern P, int iters, float *out) {
    __shared__ float box[4][4096];  // 16 KB per wave: 16x16x16 voxels
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int i = lane; i < 4096; i += 64) box[w][i] = vol[i + w * 4096];
    __syncthreads();
    const int lb = P.lane_base[lane], ls = P.lane_s[lane];
    float acc = 0.f;
    for (int it = 0; it < iters; ++it) {
        const int k = (it * P.adv + ls) & 31;  // stay inside the box
        const int o = lb + ((k * P.qx) >> 10) * 256 + ((k * P.qy) >> 10) * 16 + ((k * P.qz) >> 10);
#pragma unroll
        for (int t = 0; t < LOADS; ++t) {
            const int a = (o + (t & 1) * 256 + ((t >> 1) & 1) * 16 + (t >> 2)) & 4095;
            acc += box[w][a];
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void valu_rate(float *out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int k = 0; k < 16; ++k) { a = a * b + c; c = c * b + d; d = d * b + a; b = b * 0.99999f + 1e-6f; }
    }
    out[blockIdx.x * 256 + threadIdx.x] = a + b + c + d;
}

static Pattern lane_ray(double dx, double dy, double dz, int SX, int SY, int plane /*0: patch in yz, 1: xy, 2: xz*/) {
    Pattern P{}; P.adv = 1; P.SX = SX; P.SY = SY;
    P.qx = (int)lround(dx * 0.29 * 1024); P.qy = (int)lround(dy * 0.29 * 1024); P.qz = (int)lround(dz * 0.29 * 1024);
    for (int l = 0; l < 64; ++l) {
        int a = (int)floor((l >> 3) * 1.5), b = (int)floor((l & 7) * 1.5);
        P.lane_base[l] = plane == 0 ? a * SY + b : plane == 1 ? a * SX + b * SY : a * SX + b;
        P.lane_s[l] = 0;
    }
    return P;
}
static Pattern lane_sample(double dx, double dy, double dz, int SX, int SY) {
    Pattern P{}; P.adv = 64; P.SX = SX; P.SY = SY;
    P.qx = (int)lround(dx * 0.29 * 1024); P.qy = (int)lround(dy * 0.29 * 1024); P.qz = (int)lround(dz * 0.29 * 1024);
    for (int l = 0; l < 64; ++l) { P.lane_base[l] = 0; P.lane_s[l] = l; }
    return P;
}

int main() {
    const int N = 512; const long NE = (long)N * N * N;
    float *vol, *out; CK(hipMalloc(&vol, NE * 4 + 4096)); CK(hipMalloc(&out, 4096 * 256 * 4));
    CK(hipMemset(vol, 0, NE * 4 + 4096));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const double clk = prop.clockRate * 1e3; const int CUs = prop.multiProcessorCount;
    printf("device %s, %d CUs, %.0f MHz\n", prop.name, CUs, clk / 1e6);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int SX = N * N, SY = N;
    const double inv3 = 1 / sqrt(3.0);
    struct Case { const char *name; Pattern P; } cases[] = {
        {"lane=ray   patch yz, march x      ", lane_ray(1, 0.1, 0.1, SX, SY, 0)},
        {"lane=ray   patch xy, march z      ", lane_ray(0.1, 0.1, 1, SX, SY, 1)},
        {"lane=ray   patch xz, march y      ", lane_ray(0.1, 1, 0.1, SX, SY, 2)},
        {"lane=ray   patch yz, march diag   ", lane_ray(inv3, inv3, inv3, SX, SY, 0)},
        {"lane=sample march x               ", lane_sample(1, 0.1, 0.1, SX, SY)},
        {"lane=sample march y               ", lane_sample(0.1, 1, 0.1, SX, SY)},
        {"lane=sample march z               ", lane_sample(0.1, 0.1, 1, SX, SY)},
        {"lane=sample march diag            ", lane_sample(inv3, inv3, inv3, SX, SY)},
    };
    const int blocks = CUs * 4, iters_ray = 256, iters_smp = 16;  // 16 waves per CU
    // region per wave: spread waves over the volume but keep inside it
    const long wave_region = (long)SX * 3 + SY * 7 + 13; const long mask = (1L << 25) - 1;  // 32M-element window for wave bases
#define RUN(label, launch, nload_per_wave)                                                              \
    do {                                                                                                  \
        launch(); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); for (int r = 0; r < 5; ++r) { launch(); } \
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1)); \
        ms /= 5; double winstr = (double)blocks * 4 * (nload_per_wave);                                     \
        printf("%-58s %8.3f ms  %7.1f cyc/CU per wave-load\n", label, ms, ms * 1e-3 * clk * CUs / winstr);  \
    } while (0)
    char lab[128];
    for (auto &c : cases) {  // host-side bounds proof: max element offset any lane can form
        long mx = mask;
        int lbmax = 0; for (int l = 0; l < 64; ++l) lbmax = lbmax > c.P.lane_base[l] ? lbmax : c.P.lane_base[l];
        mx += lbmax + (long)((511L * c.P.qx) >> 10) * SX + (long)((511L * c.P.qy) >> 10) * SY + ((511L * c.P.qz) >> 10);
        mx += 3L * SX + SY + 4;
        if (mx >= NE) { printf("pattern %s would read out of bounds (%ld >= %ld)\n", c.name, mx, NE); return 2; }
    }
    for (auto &c : cases) {
        int iters = c.P.adv == 1 ? iters_ray : iters_smp * 16;
        snprintf(lab, sizeof lab, "global dword   x8  %s", c.name);
        auto l1 = [&] { hipLaunchKernelGGL((gather_global<1, 8>), dim3(blocks), dim3(256), 0, 0, vol, c.P, iters, wave_region, mask, out); };
        RUN(lab, l1, (double)iters * 8);
        snprintf(lab, sizeof lab, "global dwordx2 x8  %s", c.name);
        auto l2 = [&] { hipLaunchKernelGGL((gather_global<2, 8>), dim3(blocks), dim3(256), 0, 0, vol, c.P, iters, wave_region, mask, out); };
        RUN(lab, l2, (double)iters * 8);
        snprintf(lab, sizeof lab, "global dwordx4 x8  %s", c.name);
        auto l4 = [&] { hipLaunchKernelGGL((gather_global<4, 8>), dim3(blocks), dim3(256), 0, 0, vol, c.P, iters, wave_region, mask, out); };
        RUN(lab, l4, (double)iters * 8);
    }
    for (auto &c : cases) {
        Pattern P = c.P; P.SX = 256; P.SY = 16;
        for (int l = 0; l < 64; ++l) {  // re-express lane bases in the 16^3 box
            int a = (int)floor((l >> 3) * 1.5), b = (int)floor((l & 7) * 1.5);
            if (P.adv == 1) P.lane_base[l] = (c.name[17] == 'y' && c.name[18] == 'z') ? a * 16 + b : (c.name[17] == 'x' && c.name[18] == 'y') ? a * 256 + b * 16 : a * 256 + b;
        }
        int iters = 2048;
        snprintf(lab, sizeof lab, "LDS ds_read_b32 x8 %s", c.name);
        auto ll = [&] { hipLaunchKernelGGL((gather_lds<8>), dim3(blocks), dim3(256), 0, 0, vol, P, iters, out); };
        RUN(lab, ll, (double)iters * 8);
    }
    {
        const int iters = 4096;
        hipLaunchKernelGGL(valu_rate, dim3(CUs * 8), dim3(256), 0, 0, out, iters); CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0)); hipLaunchKernelGGL(valu_rate, dim3(CUs * 8), dim3(256), 0, 0, out, iters); CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1)); float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        double winstr = (double)CUs * 8 * 4 * iters * 64;  // wave-level FMA instructions
        printf("VALU v_fma_f32: %.3f ms, %.2f cycles/CU per wave-instr (%.1f TFLOP/s)\n", ms, ms * 1e-3 * clk * CUs / winstr,
               winstr * 64 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
