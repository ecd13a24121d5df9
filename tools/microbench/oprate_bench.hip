// oprate_bench.hip -- issue rate (SIMD cycles per wave64 instruction) of the VALU opcodes the march kernels use.
// Each kernel runs 8 independent dependency chains of one opcode (inline asm, so the compiler cannot fold them).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)

#define KERNEL(NAME, ASM, TYPE, CONS) KERNELC(NAME, ASM, TYPE, CONS, "memory")
#define KERNELC(NAME, ASM, TYPE, CONS, ...)                                                      \
    __global__ __launch_bounds__(256) void NAME(TYPE *out, int iters, TYPE b, TYPE c) {       \
        TYPE a[8];                                                                            \
        for (int i = 0; i < 8; ++i) a[i] = (TYPE)(threadIdx.x + i + 1);                       \
        for (int it = 0; it < iters; ++it) {                                                  \
            _Pragma("unroll") for (int r = 0; r < 8; ++r)                                     \
            _Pragma("unroll") for (int i = 0; i < 8; ++i) asm volatile(ASM : "+v"(a[i]) : CONS(b), CONS(c) : __VA_ARGS__); \
        }                                                                                     \
        TYPE s = 0; for (int i = 0; i < 8; ++i) s += a[i];                                    \
        out[blockIdx.x * 256 + threadIdx.x] = s;                                              \
    }
#define V "v"
KERNEL(k_fma, "v_fma_f32 %0, %0, %1, %2", float, V)
KERNEL(k_add, "v_add_f32 %0, %0, %1", float, V)
KERNEL(k_mul, "v_mul_f32 %0, %0, %1", float, V)
KERNEL(k_max, "v_max_f32 %0, %0, %1", float, V)
KERNEL(k_floor, "v_floor_f32 %0, %0", float, V)
KERNEL(k_cvt_i32_f32, "v_cvt_i32_f32 %0, %0", float, V)
KERNEL(k_cvt_f32_i32, "v_cvt_f32_i32 %0, %0", float, V)
KERNEL(k_rcp, "v_rcp_f32 %0, %0", float, V)
KERNEL(k_rsq, "v_rsq_f32 %0, %0", float, V)
KERNEL(k_mov, "v_mov_b32 %0, %1", float, V)
KERNEL(k_mov_dpp, "v_mov_b32_dpp %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", float, V)
KERNEL(k_add_dpp, "v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", float, V)
KERNELC(k_cndmask, "v_cndmask_b32 %0, %0, %1, vcc", float, V, "vcc")
KERNELC(k_cndmask_e32, "v_cndmask_b32_e32 %0, %0, %1, vcc", float, V, "vcc")
KERNELC(k_cndmask_sgpr, "v_cndmask_b32_e64 %0, %0, %1, s[20:21]", float, V, "s20", "s21")
KERNELC(k_cmp_cndmask, "v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32_e32 %0, %0, %2, vcc", float, V, "vcc")
KERNELC(k_cmp_sgpr, "v_cmp_lt_f32_e64 s[20:21], %0, %1", float, V, "s20", "s21")
KERNELC(k_cmp, "v_cmp_lt_f32 vcc, %0, %1", float, V, "vcc")
KERNEL(k_add_u32, "v_add_u32 %0, %0, %1", int, V)
KERNEL(k_lshl_add, "v_lshl_add_u32 %0, %0, 2, %1", int, V)
KERNEL(k_add3, "v_add3_u32 %0, %0, %1, %2", int, V)
KERNEL(k_mul_lo, "v_mul_lo_u32 %0, %0, %1", int, V)
KERNEL(k_mul_hi, "v_mul_hi_i32 %0, %0, %1", int, V)
KERNEL(k_mul_u24, "v_mul_u32_u24 %0, %0, %1", int, V)
KERNEL(k_mad_u24, "v_mad_u32_u24 %0, %0, %1, %2", int, V)
KERNEL(k_min_i32, "v_min_i32 %0, %0, %1", int, V)
KERNEL(k_bfe, "v_bfe_u32 %0, %0, 3, 5", int, V)
KERNEL(k_cvt_rpi, "v_cvt_rpi_i32_f32 %0, %0", float, V)
KERNEL(k_cvt_flr, "v_cvt_flr_i32_f32 %0, %0", float, V)
KERNEL(k_fract, "v_fract_f32 %0, %0", float, V)
KERNEL(k_max_dpp, "v_max_f32_dpp %0, %0, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1", float, V)
KERNEL(k_frexp_exp, "v_frexp_exp_i32_f32 %0, %0", float, V)
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %1", float, V)
KERNEL(k_ashr, "v_ashrrev_i32 %0, 31, %0", int, V)
KERNEL(k_lshl, "v_lshlrev_b32 %0, 3, %0", int, V)
KERNEL(k_perm, "v_perm_b32 %0, %0, %1, %2", int, V)
KERNELC(k_readlane, "v_readlane_b32 s20, %0, 63", int, V, "s20")

__global__ __launch_bounds__(256) void k_mad64(int *out, int iters, int b, int c) {
    unsigned long long a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i + 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(a[i]) : "v"(b), "v"(c) : "vcc");
    }
    unsigned long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (int)s;
}

// 64-bit results: sign-extending multiply (q * 2^k as one instruction) and float -> double conversion
__global__ __launch_bounds__(256) void k_mad_i64_i32(int *out, int iters, int b, int c) {
    long long a[8]; int q[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0; q[i] = threadIdx.x + i + 1; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(q[i]), "v"(b) : "vcc");
    }
    long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (int)s;
}
__global__ __launch_bounds__(256) void k_mad_i64_i32_sgpr(int *out, int iters, int b, int c) {
    long long a[8]; int q[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0; q[i] = threadIdx.x + i + 1; }
    const int bs = __builtin_amdgcn_readfirstlane(b);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_mad_i64_i32 %0, vcc, %1, %2, 0" : "=v"(a[i]) : "v"(q[i]), "s"(bs) : "vcc");
    }
    long long s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (int)s;
}
__global__ __launch_bounds__(256) void k_cvt_f64_f32(int *out, int iters, int b, int c) {
    double a[8]; float q[8];
    for (int i = 0; i < 8; ++i) { a[i] = 0; q[i] = (float)(threadIdx.x + i + 1); }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(q[i]));
    }
    double s = 0; for (int i = 0; i < 8; ++i) s += a[i];
    out[blockIdx.x * 256 + threadIdx.x] = (int)s;
}

template <typename T, typename K>
static void run(const char *name, K kern, T *out, int CUs, double clk) {
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int iters = 2048, bpc = 4;
    hipLaunchKernelGGL(kern, dim3(CUs * bpc), dim3(256), 0, 0, out, iters, (T)3, (T)5); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(kern, dim3(CUs * bpc), dim3(256), 0, 0, out, iters, (T)3, (T)5);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double winstr = (double)CUs * bpc * 4 * iters * 64;
    printf("%-16s %.2f SIMD-cycles per wave-instr (4 waves/SIMD)\n", name, ms * 1e-3 * clk * CUs * 4 / winstr);
}
int main() {
    void *out; CK(hipMalloc(&out, 256 * 8 * 256 * 8));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int CUs = prop.multiProcessorCount; const double clk = prop.clockRate * 1e3;
    printf("CUs %d clock %.0f MHz\n", CUs, clk / 1e6);
#define RF(k) run<float>(#k, k, (float *)out, CUs, clk);
#define RI(k) run<int>(#k, k, (int *)out, CUs, clk);
    RF(k_fma) RF(k_add) RF(k_mul) RF(k_max) RF(k_floor) RF(k_cvt_i32_f32) RF(k_cvt_f32_i32) RF(k_rcp) RF(k_rsq) RF(k_mov)
    RF(k_mov_dpp) RF(k_add_dpp) RF(k_cndmask) RF(k_cndmask_e32) RF(k_cndmask_sgpr) RF(k_cmp_cndmask) RF(k_cmp_sgpr) RF(k_cmp)
    RI(k_add_u32) RI(k_lshl_add) RI(k_add3) RI(k_mul_lo) RI(k_mul_hi) RI(k_mul_u24) RI(k_mad_u24) RI(k_min_i32) RI(k_bfe) RI(k_readlane)
    RI(k_mad64) RI(k_mad_i64_i32) RI(k_mad_i64_i32_sgpr) RI(k_cvt_f64_f32)
    RF(k_cvt_rpi) RF(k_cvt_flr) RF(k_fract) RF(k_max_dpp) RF(k_frexp_exp) RF(k_ldexp) RI(k_ashr) RI(k_lshl) RI(k_perm)
    return 0;
}
