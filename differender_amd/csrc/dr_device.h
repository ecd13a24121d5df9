// dr_device.h -- device-side building blocks shared by the gfx950 march kernels.
// Semantics follow differender/volume_raycaster.py ("VR.py") of the reference; citations inline.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <stdint.h>
#include "dr_experiment.h"

namespace dr {

struct f3 { float x, y, z; };
__device__ __forceinline__ f3 make_f3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
// Arithmetic convention D2: a*b+c in dot/mix/position/compositing is ONE rounding, spelled as fmaf;
// every kernel translation unit is compiled with -ffp-contract=off so nothing else is contracted.
__device__ __forceinline__ float dot3(f3 a, f3 b) { return fmaf(a.z, b.z, fmaf(a.y, b.y, a.x * b.x)); }
// taichi Vector.normalized(): invlen = 1/(norm + 0); invlen * v
__device__ __forceinline__ f3 normalized3(f3 a) {
    float inv = 1.0f / (sqrtf(dot3(a, a)) + 0.0f);
    return make_f3(inv * a.x, inv * a.y, inv * a.z);
}
// taichi_glsl mix
__device__ __forceinline__ float mixf(float x, float y, float a) { return fmaf(y, a, x * (1.0f - a)); }

// jitter RNG (replaces ti.random, VR.py:255): counter-based integer hash of (seed, view, pixel)
__device__ __forceinline__ uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ float jitter_u(uint32_t seed, uint32_t view, uint32_t pix) {
    uint32_t h = hash_u32(seed ^ 0x9E3779B9U);
    h = hash_u32(h ^ (view * 0x85EBCA6BU + 0xC2B2AE35U));
    h = hash_u32(h ^ pix);
    return (float)(h >> 8) * (1.0f / 16777216.0f);
}

// pow(x, 32) for x >= 0 (VR.py:296, shininess 32) by five squarings: <= 16 ulp from powf, i.e. < 3e-7
// absolute on the specular term, and costs 5 multiplies instead of a log/exp pair.
__device__ __forceinline__ float pow32(float x) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return x16 * x16;
}
// x^31 = x^16 * x^8 * x^4 * x^2 * x (derivative of the specular term)
__device__ __forceinline__ float pow31(float x) {
    float x2 = x * x, x4 = x2 * x2, x8 = x4 * x4, x16 = x8 * x8;
    return ((x16 * x8) * (x4 * x2)) * x;
}

// VR.py:7-21 low_high_frac
__device__ __forceinline__ void low_high_frac(float x, int &lo, float &fr) {
    x = fmaxf(x, 0.0f);
    float low = floorf(x);
    fr = x - low;
    lo = (int)low;
}

__device__ __forceinline__ float ld_voxel(const float *p) { return *p; }
__device__ __forceinline__ float ld_voxel(const __half *p) { return __half2float(*p); }

// Volume view in the reference's field index space (VR.py:481), element strides.
template <typename VT>
struct VolView {
    const VT *p;
    int64_t sx, sy, sz;
    int VX, VY, VZ;
    float scx, scy, scz;  // shape - 1 - 1e-4, rounded from double (VR.py:165)
};

struct Cell {
    int x0, x1, y0, y1, z0, z1;
    float fx, fy, fz;
};

// VR.py:163-172: clamp to [0,1], scale, split, clamp the high index
template <typename VT>
__device__ __forceinline__ void tri_cell(const VolView<VT> &v, float px, float py, float pz, Cell &c) {
    float qx = fminf(1.0f, fmaxf(0.0f, fmaf(0.5f, px, 0.5f))) * v.scx;
    float qy = fminf(1.0f, fmaxf(0.0f, fmaf(0.5f, py, 0.5f))) * v.scy;
    float qz = fminf(1.0f, fmaxf(0.0f, fmaf(0.5f, pz, 0.5f))) * v.scz;
    low_high_frac(qx, c.x0, c.fx);
    low_high_frac(qy, c.y0, c.fy);
    low_high_frac(qz, c.z0, c.fz);
    c.x0 = min(c.x0, v.VX - 1); c.y0 = min(c.y0, v.VY - 1); c.z0 = min(c.z0, v.VZ - 1);
    c.x1 = min(c.x0 + 1, v.VX - 1); c.y1 = min(c.y0 + 1, v.VY - 1); c.z1 = min(c.z0 + 1, v.VZ - 1);
}

// two voxels that are neighbours in memory, fetched as ONE element-pair load (4-byte aligned: global_load_dwordx2 / dword)
struct __attribute__((packed, aligned(4))) VoxPairF { float a, b; };
struct __attribute__((packed, aligned(2))) VoxPairH { unsigned short a, b; };
__device__ __forceinline__ void ld_voxel_pair(const float *p, float &a, float &b) { const VoxPairF q = *reinterpret_cast<const VoxPairF *>(p); a = q.a; b = q.b; }
__device__ __forceinline__ void ld_voxel_pair(const __half *p, float &a, float &b) {
    const VoxPairH q = *reinterpret_cast<const VoxPairH *>(p);
    a = __half2float(__ushort_as_half(q.a)); b = __half2float(__ushort_as_half(q.b));
}

// VR.py:173-189: 8 gathers, lerp order x -> y -> z. When one axis of the volume is contiguous (stride 1: z for a field-order tensor,
// x for the reference's (1, D, H, W) layout seen through Raycaster) the cell's corners come in four neighbour PAIRS: four loads
// instead of eight -- a wave-wide gather costs the texture-address path ~25 cycles whatever its width, and the per-ray kernels
// (ray_cross_kernel: 72 % of its time there at sampling rate 8) are bound by exactly that. Same voxels, same lerps, same bits.
template <typename VT>
__device__ __forceinline__ float tri_sample(const VolView<VT> &v, float px, float py, float pz) {
    Cell c;
    tri_cell(v, px, py, pz, c);
    if (v.sx == 1 && c.x1 == c.x0 + 1) {   // (uniform stride test; the high index is only clamped for positions tri_cell never produces)
        const VT *b0 = v.p + c.x0 + c.y0 * v.sy, *b1 = v.p + c.x0 + c.y1 * v.sy;
        const int64_t o0 = c.z0 * v.sz, o1 = c.z1 * v.sz;
        float v000, v100, v010, v110, v001, v101, v011, v111;
        ld_voxel_pair(b0 + o0, v000, v100); ld_voxel_pair(b1 + o0, v010, v110);
        ld_voxel_pair(b0 + o1, v001, v101); ld_voxel_pair(b1 + o1, v011, v111);
        const float zl = mixf(mixf(v000, v100, c.fx), mixf(v010, v110, c.fx), c.fy);
        const float zh = mixf(mixf(v001, v101, c.fx), mixf(v011, v111, c.fx), c.fy);
        return mixf(zl, zh, c.fz);
    }
    if (v.sz == 1 && c.z1 == c.z0 + 1) {
        const VT *b00 = v.p + c.x0 * v.sx + c.y0 * v.sy + c.z0, *b10 = v.p + c.x1 * v.sx + c.y0 * v.sy + c.z0;
        const VT *b01 = v.p + c.x0 * v.sx + c.y1 * v.sy + c.z0, *b11 = v.p + c.x1 * v.sx + c.y1 * v.sy + c.z0;
        float v000, v001, v100, v101, v010, v011, v110, v111;
        ld_voxel_pair(b00, v000, v001); ld_voxel_pair(b10, v100, v101);
        ld_voxel_pair(b01, v010, v011); ld_voxel_pair(b11, v110, v111);
        const float zl = mixf(mixf(v000, v100, c.fx), mixf(v010, v110, c.fx), c.fy);
        const float zh = mixf(mixf(v001, v101, c.fx), mixf(v011, v111, c.fx), c.fy);
        return mixf(zl, zh, c.fz);
    }
    const VT *b00 = v.p + c.x0 * v.sx + c.y0 * v.sy, *b10 = v.p + c.x1 * v.sx + c.y0 * v.sy;
    const VT *b01 = v.p + c.x0 * v.sx + c.y1 * v.sy, *b11 = v.p + c.x1 * v.sx + c.y1 * v.sy;
    int64_t o0 = c.z0 * v.sz, o1 = c.z1 * v.sz;
    float a = mixf(ld_voxel(b00 + o0), ld_voxel(b10 + o0), c.fx);
    float b = mixf(ld_voxel(b01 + o0), ld_voxel(b11 + o0), c.fx);
    float zl = mixf(a, b, c.fy);
    a = mixf(ld_voxel(b00 + o1), ld_voxel(b10 + o1), c.fx);
    b = mixf(ld_voxel(b01 + o1), ld_voxel(b11 + o1), c.fx);
    float zh = mixf(a, b, c.fy);
    return mixf(zl, zh, c.fz);
}

// Float atomic add to a gradient in global memory that can never leave +-FLT_MAX (the "finite by construction" promise of
// the sanitising backward). Addends up to 1e30 -- everything the kernels produce from ordinary upstream
// gradients, and every per-sample contribution after its clamp -- go through the hardware atomic: 3e8 of them would have to
// meet in one element before the sum overflowed. Only the rare larger addend (a brick's or a workgroup's total under
// upstream gradients beyond ~1e30) takes a compare-and-swap loop that clamps the sum: what torch.nan_to_num would have made
// of the infinity (VR.py:463-475), without the two extra passes over the tensor. A later ordinary addend leaves +-FLT_MAX
// where it is (1e30 is less than half an ulp of it).
__device__ __forceinline__ void atomic_add_sat(float *p, float v) {
    if (fabsf(v) <= 1.0e30f) { unsafeAtomicAdd(p, v); return; }
    if (!(v == v)) return;   // (a NaN total cannot arise from sanitised addends; dropped all the same)
    unsigned int *u = reinterpret_cast<unsigned int *>(p);
    unsigned int old = __hip_atomic_load(u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), assumed;
    do {
        assumed = old;
        const float nv = fminf(fmaxf(__uint_as_float(assumed) + v, -3.4028234664e38f), 3.4028234664e38f);
        old = atomicCAS(u, assumed, __float_as_uint(nv));
    } while (old != assumed);
}

// adjoint of tri_sample w.r.t. the volume (global float atomics; baseline path)
struct GradView {
    float *p;
    int64_t sx, sy, sz;
};
template <typename VT>
__device__ __forceinline__ void tri_scatter_global(const VolView<VT> &v, const GradView &g, float px, float py,
                                                   float pz, float adj) {
    Cell c;
    tri_cell(v, px, py, pz, c);
    float gx = 1.0f - c.fx, gy = 1.0f - c.fy, gz = 1.0f - c.fz;
    float *b00 = g.p + c.x0 * g.sx + c.y0 * g.sy, *b10 = g.p + c.x1 * g.sx + c.y0 * g.sy;
    float *b01 = g.p + c.x0 * g.sx + c.y1 * g.sy, *b11 = g.p + c.x1 * g.sx + c.y1 * g.sy;
    int64_t o0 = c.z0 * g.sz, o1 = c.z1 * g.sz;
    unsafeAtomicAdd(b00 + o0, gx * gy * gz * adj);
    unsafeAtomicAdd(b10 + o0, c.fx * gy * gz * adj);
    unsafeAtomicAdd(b01 + o0, gx * c.fy * gz * adj);
    unsafeAtomicAdd(b11 + o0, c.fx * c.fy * gz * adj);
    unsafeAtomicAdd(b00 + o1, gx * gy * c.fz * adj);
    unsafeAtomicAdd(b10 + o1, c.fx * gy * c.fz * adj);
    unsafeAtomicAdd(b01 + o1, gx * c.fy * c.fz * adj);
    unsafeAtomicAdd(b11 + o1, c.fx * c.fy * c.fz * adj);
}

// Per-sample quantities of VR.py:270-299
struct Sample {
    float px, py, pz;
    float I, xtf, fr;
    int lo, hi;
    float r, g, b, a, op;
    f3 grad, nrm, ld, rf;
    float gnorm, ginv, m, ndl, q, rdv, spec, Lraw, L;  // ginv = 1/gnorm
    bool flat;
};

struct RayGeom {
    float entry, exit_, vx, vy, vz;
    int n;
    float t0;      // entry + 0.5*len/n (VR.py:273-275)
    float inv_nm1; // unused by the baseline (kept exact: division per sample)
};

// VR.py:277-280: pos = cam + mix(t0, exit, s/(n-1)) * vd
__device__ __forceinline__ void sample_pos(const RayGeom &rg, float cx, float cy, float cz, int s, float &px, float &py,
                                           float &pz) {
    float f = (float)s / (float)(rg.n - 1);
    float t = mixf(rg.t0, rg.exit_, f);
    px = fmaf(t, rg.vx, cx); py = fmaf(t, rg.vy, cy); pz = fmaf(t, rg.vz, cz);
}

// The same position with the quotient s/(n-1) from a per-ray reciprocal y = RN(1/(n-1)): q0 = RN(s*y),
// r = s - q0*(n-1) (exact in one fma), q = RN(q0 + r*y). By Markstein's theorem this IS the correctly rounded
// quotient (s, n-1 are integers below 2^23, so the divisor's significand is never all ones) -- checked
// exhaustively for n-1 <= 8192 and on 2e8 random pairs up to 8e6 (tests/test_host_logic.py). 3 full-rate VALU
// instead of the ~12 of the IEEE division sequence, bit-identical positions.
__device__ __forceinline__ void sample_pos_rcp(float t0, float exit_, float nm1, float inv_nm1, float vx, float vy,
                                               float vz, float cx, float cy, float cz, int s, float &px, float &py,
                                               float &pz) {
    const float sf = (float)s;
    float q = sf * inv_nm1;
    const float r = fmaf(-q, nm1, sf);
    q = fmaf(r, inv_nm1, q);
    const float t = mixf(t0, exit_, q);
    px = fmaf(t, vx, cx); py = fmaf(t, vy, cy); pz = fmaf(t, vz, cz);
}

// (1 - a)^(1/sr) of VR.py:284-285 -- a SPECIFIED function (the CPU checker of the tests restates it),
// because `ti.pow` is CUDA's approximate __powf in the reference (not a pinned function) and the result is ill-conditioned
// in f32: one ulp of the power is a relative 1e-4 of the opacity at sampling rate 16 and small alpha, and whether a ray
// crosses alpha 0.99 at sample s or s + 1 can hinge on it. Both sides evaluate, bit for bit:
//   * 1/sr = 2^-k (sampling rates 2, 4, 8, 16 -- with 1, all the rates of the reference's scripts: OPT.py:49,67, ND.py:27,
//     raycast_nondiff's default 4x): k correctly rounded square roots. Total error < 1 ulp, ~10 instructions each instead
//     of the ~100 of powf, which at sampling rate 8 was most of the arithmetic of the alpha pre-pass.
//   * any other exponent: exp2(y * log2(x)) in DOUBLE precision from IEEE +, -, *, / only (an atanh series for the
//     logarithm, a Taylor polynomial for the exponential; relative error < 1e-11), rounded once to float: the
//     correctly rounded power on every one of 4e5 sampled arguments (CPU test of the same formula), and the same float on
//     any IEEE platform always. Every translation unit is built with -ffp-contract=off: p * z2 + c is a multiply and an add.
__device__ __forceinline__ float pow_spec(float x, float y) {
    if (!(x > 0.0f)) return (x == 0.0f) ? 0.0f : __builtin_nanf("");
    const long long bits = __double_as_longlong((double)x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = __longlong_as_double((bits & 0x000fffffffffffffll) | 0x3ff0000000000000ll);  // [1, 2)
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }                                          // [0.707, 1.414]
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;                                        // |z| <= 0.172
    double p = 2.0 / 13.0;
    p = p * z2 + 2.0 / 11.0; p = p * z2 + 2.0 / 9.0; p = p * z2 + 2.0 / 7.0; p = p * z2 + 2.0 / 5.0;
    p = p * z2 + 2.0 / 3.0; p = p * z2 + 2.0;
    const double t = (double)y * ((double)e + (p * z) * 1.4426950408889634);                  // y * log2(x)
    if (!(t > -160.0)) return 0.0f;
    if (!(t < 128.0)) return __builtin_inff();
    const double n = __builtin_rint(t), r = (t - n) * 0.6931471805599453;                      // |r| <= 0.347
    double q = 1.0 / 39916800.0;
    q = q * r + 1.0 / 3628800.0; q = q * r + 1.0 / 362880.0; q = q * r + 1.0 / 40320.0; q = q * r + 1.0 / 5040.0;
    q = q * r + 1.0 / 720.0; q = q * r + 1.0 / 120.0; q = q * r + 1.0 / 24.0; q = q * r + 1.0 / 6.0;
    q = q * r + 0.5; q = q * r + 1.0; q = q * r + 1.0;
    return (float)(q * __longlong_as_double((long long)((int)n + 1023) << 52));
}
// Correctly rounded square root -- CONTRACT: x = 0, x >= 2^-96, NaN or x < 0 (-> NaN). v_sqrt_f32 (1 ulp) and the
// compiler's own one-ulp correction from the two exact fma residuals -- without the scaling that sqrtf() wraps around it
// for small arguments (9 instead of 17 instructions; three of them per sample at sampling rate 8). Below 2^-96 the
// residuals x - (y -+ ulp) y underflow and the correction can pick the wrong neighbour: such arguments are outside the
// contract (pow_inv_sr, the only caller, sends them to pow_spec). 1 - alpha never gets there: the difference is exact for
// alpha in [0.5, 1] and then 0 or >= 2^-24, and the nested roots only move towards 1.
// Checked against the host's sqrtf on every non-negative float (2 139 095 041 of them; the 1 879 048 194 of the contract must
// match, tools/microbench/sqrt_cr_check.hip), and pow_inv_sr's sqrtf branch on every 0 < x < 2^-96 with tiny and normal lanes mixed.
__device__ __forceinline__ float sqrt_cr(float x) {
    const float y = __builtin_amdgcn_sqrtf(x);
    const float ym = __int_as_float(__float_as_int(y) - 1), yp = __int_as_float(__float_as_int(y) + 1);
    const float rd = fmaf(-ym, y, x), ru = fmaf(-yp, y, x);  // x - (y -+ ulp) y
    float r = (rd <= 0.0f) ? ym : y;
    r = (ru > 0.0f) ? yp : r;
    return r;
}
__device__ __forceinline__ float pow_inv_sr(float base, float inv_sr) {
    if (inv_sr == 1.0f) return base;
    // (the first root of a base below sqrt_cr's contract -- not reachable from 1 - alpha, see above -- is the library's sqrtf,
    // scaling included: still the correctly rounded root the oracle takes, D6; one comparison, false for every lane in practice;
    // the roots that follow are of numbers >= 2^-75)
    if (inv_sr == 0.5f || inv_sr == 0.25f || inv_sr == 0.125f || inv_sr == 0.0625f) {
        const bool tiny = base != 0.0f && base < 1.2621774483536189e-29f;   // 2^-96
        float r;
        if (__any(tiny)) r = tiny ? sqrtf(base) : sqrt_cr(base);   // (wave-uniform branch: the 17-instruction sqrtf stays off the hot path)
        else r = sqrt_cr(base);
        if (inv_sr <= 0.25f) r = sqrt_cr(r);
        if (inv_sr <= 0.125f) r = sqrt_cr(r);
        if (inv_sr <= 0.0625f) r = sqrt_cr(r);
        return r;
    }
    return pow_spec(base, inv_sr);
}

// VR.py:205-219 + 284-285. tf is a [R][4] table (LDS or global). sm.I must be set.
// Two halves: the TF lookup, and the opacity of the sample's alpha at this sampling rate -- at rates other than 1 a
// power of 10 - 60 instructions that the non-differentiable march only needs where alpha > 1e-3 (VR.py:334).
// ... its alpha alone, from a table of alphas (the alpha pre-pass keeps nothing else of the TF in LDS): the same index, fraction
// and lerp as below, so sm.a is bit-identical
__device__ __forceinline__ void tf_alpha_from_I(const float *tfa, int R, float tf_len, Sample &sm) {
    sm.xtf = sm.I * tf_len;
    low_high_frac(sm.xtf, sm.lo, sm.fr);
    sm.lo = min(sm.lo, R - 1);
    sm.hi = min(sm.lo + 1, R - 1);
    sm.a = mixf(tfa[sm.lo], tfa[sm.hi], sm.fr);
}
__device__ __forceinline__ void tf_lookup_from_I(const float4 *tf, int R, float tf_len, Sample &sm) {
    sm.xtf = sm.I * tf_len;
    low_high_frac(sm.xtf, sm.lo, sm.fr);
    sm.lo = min(sm.lo, R - 1);  // defined-domain guard for I > 1 (reference reads out of bounds)
    sm.hi = min(sm.lo + 1, R - 1);
    float4 t0 = tf[sm.lo], t1 = tf[sm.hi];
    sm.r = mixf(t0.x, t1.x, sm.fr); sm.g = mixf(t0.y, t1.y, sm.fr);
    sm.b = mixf(t0.z, t1.z, sm.fr); sm.a = mixf(t0.w, t1.w, sm.fr);
}
__device__ __forceinline__ float opacity_of_alpha(float a, float inv_sr) { return 1.0f - pow_inv_sr(1.0f - a, inv_sr); }
__device__ __forceinline__ void classify_from_I(const float4 *tf, int R, float tf_len, float inv_sr, Sample &sm) {
    tf_lookup_from_I(tf, R, tf_len, sm);
    sm.op = opacity_of_alpha(sm.a, inv_sr);
}
template <typename VT>
__device__ __forceinline__ void classify(const VolView<VT> &v, const float4 *tf, int R, float tf_len, float inv_sr,
                                         Sample &sm) {
    sm.I = tri_sample(v, sm.px, sm.py, sm.pz);
    classify_from_I(tf, R, tf_len, inv_sr, sm);
}

// VR.py:191-203 normal, :287-299 Phong, given the central differences (dx,dy,dz) of the six normal taps.
// clampL: the differentiable path clamps lighting (VR.py:298).
// FAST: the two normalisations use v_rsq_f32 (1 ulp) instead of an IEEE sqrt + divide. These feed the
// lighting term directly (no cancellation downstream), so the change is ~1e-7 in L -- unlike the taps, whose
// bits decide the normal and are never touched. The baseline kernels keep FAST = false (exact twin of the oracle).
template <bool FAST = false>
__device__ __forceinline__ void shade_from_grad(float dx, float dy, float dz, f3 light_pos, f3 vd, bool clampL,
                                                Sample &sm) {
    sm.grad = make_f3(dx, dy, dz);
    const float gn2 = dot3(sm.grad, sm.grad);
    const f3 lv = make_f3(sm.px - light_pos.x, sm.py - light_pos.y, sm.pz - light_pos.z);
    if (FAST) {
        sm.flat = !(gn2 >= 1.17549435e-38f);  // zero (or denormal: |grad| < 1e-19) gradient
        sm.ginv = __builtin_amdgcn_rsqf(gn2);
        sm.gnorm = gn2 * sm.ginv;
        const float li = __builtin_amdgcn_rsqf(dot3(lv, lv));
        sm.ld = make_f3(li * lv.x, li * lv.y, li * lv.z);
    } else {
        sm.gnorm = sqrtf(gn2);
        sm.flat = !(sm.gnorm > 0.0f);
        sm.ginv = 1.0f / (sm.gnorm + 0.0f);
        sm.ld = normalized3(lv);
    }
    if (sm.flat) {
        // 0/0 normal in the reference; NaN-suppressing max() gives ndl = rdv = 0
        sm.nrm = make_f3(0.f, 0.f, 0.f);
        sm.m = 0.f; sm.ndl = 0.f; sm.rf = sm.ld; sm.q = 0.f; sm.rdv = 0.f;
    } else {
        sm.nrm = make_f3(sm.ginv * dx, sm.ginv * dy, sm.ginv * dz);
        sm.m = dot3(sm.nrm, sm.ld);
        sm.ndl = fmaxf(sm.m, 0.0f);
        float two_m = 2.0f * sm.m;
        sm.rf = make_f3(fmaf(-two_m, sm.nrm.x, sm.ld.x), fmaf(-two_m, sm.nrm.y, sm.ld.y), fmaf(-two_m, sm.nrm.z, sm.ld.z));
        sm.q = -dot3(sm.rf, vd);  // r.dot(-vd): negation is exact
        sm.rdv = fmaxf(sm.q, 0.0f);
    }
    sm.spec = 0.3f * pow32(sm.rdv);
    sm.Lraw = fmaf(0.8f, sm.ndl, sm.spec) + 0.4f;
    sm.L = clampL ? fminf(1.0f, sm.Lraw) : sm.Lraw;
}
// The adjoint of the lighting term is discontinuous where the forward has a kink: Lraw = 1 (min(1, .), VR.py:298),
// n.l = 0 and r.v = 0 (max(., 0), VR.py:291,294). The FAST normalisations move n.l by ~3e-7, r.v by ~1e-6 and Lraw by up
// to ~1e-5 (r.v^32); inside these bands the fast backward re-shades the sample exactly (shade_from_grad<false>).
__device__ __forceinline__ bool near_lighting_kink(const Sample &sm) {
    return !sm.flat && (fabsf(sm.Lraw - 1.0f) < 6e-5f || fabsf(sm.m) < 4e-6f || fabsf(sm.q) < 8e-6f);
}
template <typename VT>
__device__ __forceinline__ void shade(const VolView<VT> &v, f3 light_pos, f3 vd, bool clampL, Sample &sm) {
    const float delta = 1e-3f;
    float dx = tri_sample(v, sm.px + delta, sm.py, sm.pz) - tri_sample(v, sm.px - delta, sm.py, sm.pz);
    float dy = tri_sample(v, sm.px, sm.py + delta, sm.pz) - tri_sample(v, sm.px, sm.py - delta, sm.pz);
    float dz = tri_sample(v, sm.px, sm.py, sm.pz + delta) - tri_sample(v, sm.px, sm.py, sm.pz - delta);
    shade_from_grad<false>(dx, dy, dz, light_pos, vd, clampL, sm);
}

// Adjoint of one composited sample: everything downstream of the taps.
// Inputs: the shaded sample, T = 1 - A_before, suffix = gC.(C_final - C_after) + gA.(A_final - A_after)
// (0 for the last live sample), upstream gradient go. Outputs the adjoints of the TF sample (r,g,b,a),
// of the intensity tap (I_bar, needs tf slope) and of the central differences (gx,gy,gz).
struct SampleAdj {
    float r_bar, g_bar, b_bar, a_bar;  // d/d(tf colour, tf alpha) of this sample
    float gx, gy, gz;                   // d/d(dx,dy,dz); 0 when the normal is flat
    float Lop;                          // L*op*T: (r,g,b)_bar = Lop * upstream colour gradient
};
template <bool FAST = false>
__device__ __forceinline__ void sample_adjoint(const Sample &sm, f3 vd, float T, float suffix, bool last, float4 go,
                                               float inv_sr, SampleAdj &ad) {
    const float rgbdot = go.x * sm.r + go.y * sm.g + go.z * sm.b;  // gC . rgb
    const float qs = sm.L * rgbdot + go.w;
    const float sfx = FAST ? suffix * __builtin_amdgcn_rcpf(1.0f - sm.op) : suffix / (1.0f - sm.op);
    const float op_bar = T * qs - (last ? 0.0f : sfx);
    const float Lop = sm.L * sm.op * T;
    ad.Lop = Lop;
    ad.r_bar = Lop * go.x; ad.g_bar = Lop * go.y; ad.b_bar = Lop * go.z;
    const float L_bar = sm.op * T * rgbdot;
    const float Lraw_bar = (1.0f < sm.Lraw) ? 0.0f : L_bar;
    const float base = 1.0f - sm.a;
    ad.a_bar = op_bar * ((inv_sr == 1.0f) ? 1.0f : inv_sr * powf(base, inv_sr - 1.0f));
    if (sm.flat) {
        ad.gx = ad.gy = ad.gz = 0.0f;  // deviation D1: the reference injects NaN here
        return;
    }
    const float ndl_bar = 0.8f * Lraw_bar;
    const float rdv_bar = 0.3f * 32.0f * pow31(sm.rdv) * Lraw_bar;
    const float q_bar = (0.0f < sm.q) ? rdv_bar : 0.0f;
    const f3 rf_bar = make_f3(-vd.x * q_bar, -vd.y * q_bar, -vd.z * q_bar);
    const float m_bar = ((0.0f < sm.m) ? ndl_bar : 0.0f) - 2.0f * dot3(sm.nrm, rf_bar);
    const float m2 = -2.0f * sm.m;
    const f3 n_bar = make_f3(m2 * rf_bar.x + m_bar * sm.ld.x, m2 * rf_bar.y + m_bar * sm.ld.y,
                             m2 * rf_bar.z + m_bar * sm.ld.z);
    const float nn = dot3(sm.nrm, n_bar);
    const float inv = sm.ginv;
    ad.gx = inv * (n_bar.x - sm.nrm.x * nn);
    ad.gy = inv * (n_bar.y - sm.nrm.y * nn);
    ad.gz = inv * (n_bar.z - sm.nrm.z * nn);
}
// adjoint of the intensity tap given the TF texels around it
__device__ __forceinline__ float intensity_adjoint(const Sample &sm, float4 t0, float4 t1, const SampleAdj &ad,
                                                   float tf_len) {
    const float fr_bar = (t1.x - t0.x) * ad.r_bar + (t1.y - t0.y) * ad.g_bar + (t1.z - t0.z) * ad.b_bar +
                         (t1.w - t0.w) * ad.a_bar;
    return (0.0f < sm.xtf) ? fr_bar * tf_len : 0.0f;
}

}  // namespace dr
