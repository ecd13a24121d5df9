// Exhaustive check of dr::sqrt_cr (differender_amd/csrc/dr_device.h) against the host's correctly rounded sqrtf:
// every non-negative float bit pattern 0 .. 0x7f800000. Its contract is x = 0 or x >= 2^-96 (the compiler's sqrtf scales
// smaller arguments up first: below ~2^-101 the fma residuals underflow); mismatches below that are counted separately.
//   hipcc -O2 --offload-arch=gfx950 -ffp-contract=off -I differender_amd/csrc -I include tools/microbench/sqrt_cr_check.hip -o /tmp/sqrt_cr_check && /tmp/sqrt_cr_check
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>
#include "dr_device.h"

__global__ void k(uint32_t base, uint32_t n, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = dr::sqrt_cr(__uint_as_float(base + i));
}

// pow_inv_sr(x, 2^-k), k = 1..4 (the opacity correction at sampling rates 2, 4, 8, 16), for EVERY 0 < x < 2^-96 -- the arguments
// sqrt_cr's contract excludes, which pow_inv_sr routes to the library's sqrtf under a wave-uniform __any -- with tiny and normal
// arguments MIXED in every wave (odd lanes tiny, even lanes a normal float), against nested host sqrtf.
__global__ void kpow(uint32_t base, uint32_t n_pairs, float inv_sr, float *out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= 2 * n_pairs) return;
    const uint32_t bits = (i & 1u) ? (base + (i >> 1)) : (0x30000000u + base + (i >> 1));
    out[i] = dr::pow_inv_sr(__uint_as_float(bits), inv_sr);
}
static float nested_sqrtf(float x, int k) {
    for (int j = 0; j < k; ++j) x = sqrtf(x);
    return x;
}
static int check_pow_inv_sr_tiny(float *d, std::vector<float> &h, uint32_t CH) {
    unsigned long long bad_tiny = 0, bad_normal = 0, total = 0;
    const uint32_t PAIRS = CH / 2;
    for (int k = 1; k <= 4; ++k) {
        const float inv_sr = 1.0f / (float)(1 << k);
        for (uint64_t base = 1; base < 0x0f800000ull; base += PAIRS) {
            const uint32_t n = (uint32_t)((0x0f800000ull - base) < PAIRS ? (0x0f800000ull - base) : PAIRS);
            hipLaunchKernelGGL(kpow, dim3((2 * n + 255) / 256), dim3(256), 0, 0, (uint32_t)base, n, inv_sr, d);
            if (hipMemcpy(h.data(), d, (size_t)2 * n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
#pragma omp parallel for reduction(+ : bad_tiny, bad_normal)
            for (long long i = 0; i < (long long)2 * n; ++i) {
                const uint32_t bits = (i & 1) ? ((uint32_t)base + (uint32_t)(i >> 1)) : (0x30000000u + (uint32_t)base + (uint32_t)(i >> 1));
                float x; memcpy(&x, &bits, 4);
                const float ref = nested_sqrtf(x, k);
                if (memcmp(&ref, &h[i], 4) != 0) { if (i & 1) ++bad_tiny; else ++bad_normal; }
            }
            total += 2ull * n;
        }
    }
    printf("pow_inv_sr(x, 1/2 .. 1/16) vs nested host sqrtf, tiny (0 < x < 2^-96) and normal lanes mixed in every wave: %llu evaluations; "
           "%llu mismatches on tiny lanes, %llu on normal lanes\n", total, bad_tiny, bad_normal);
    return (bad_tiny || bad_normal) ? 1 : 0;
}

int main() {
    const uint32_t CH = 1u << 26;
    float *d;
    if (hipMalloc(&d, (size_t)CH * 4) != hipSuccess) return 2;
    std::vector<float> h(CH);
    unsigned long long bad = 0, bad_denorm = 0, total = 0;
    uint32_t bad_max = 0;
    for (uint64_t base = 0; base <= 0x7f800000ull; base += CH) {
        const uint32_t n = (uint32_t)((0x7f800000ull + 1 - base) < CH ? (0x7f800000ull + 1 - base) : CH);
        hipLaunchKernelGGL(k, dim3((n + 255) / 256), dim3(256), 0, 0, (uint32_t)base, n, d);
        if (hipMemcpy(h.data(), d, (size_t)n * 4, hipMemcpyDeviceToHost) != hipSuccess) return 3;
#pragma omp parallel for reduction(+ : bad, bad_denorm) reduction(max : bad_max)
        for (long long i = 0; i < (long long)n; ++i) {
            const uint32_t bits = (uint32_t)base + (uint32_t)i;
            float x; memcpy(&x, &bits, 4);
            const float ref = sqrtf(x);
            if (memcmp(&ref, &h[i], 4) != 0 && !(ref != ref && h[i] != h[i])) { if (bits != 0 && bits < 0x0f800000u) { ++bad_denorm; if (bits > bad_max) bad_max = bits; } else ++bad; }
        }
        total += n;
    }
    float xm; memcpy(&xm, &bad_max, 4);
    printf("sqrt_cr vs host sqrtf: %llu arguments; %llu mismatches for x = 0 or x >= 2^-96 (the contract); %llu for 0 < x < 2^-96, the largest of them x = %g\n",
           total, bad, bad_denorm, xm);
    const int rc2 = check_pow_inv_sr_tiny(d, h, CH);
    return (bad || rc2) ? 1 : 0;
}
