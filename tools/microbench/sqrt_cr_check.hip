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
    return bad ? 1 : 0;
}
