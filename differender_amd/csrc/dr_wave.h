// dr_wave.h -- cross-lane building blocks of the brick kernels (gfx950 wave64): DPP moves, segmented wave scans of the
// front-to-back "over" operator, of sums and of products, whole-wave shifts. No LDS traffic anywhere in this file.
// Shared by march_flat.hip (per-brick passes) and tf_tape.hip (the TF-only backward over the per-sample tape).
#pragma once
#include <hip/hip_runtime.h>

namespace dr {

// inclusive sum of an int over the wave (DPP; lanes without a source add 0)
__device__ __forceinline__ int wave_incl_sum(int v) {
    v += __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x114, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x118, 0xf, 0xf, true);
    v += __builtin_amdgcn_update_dpp(0, v, 0x142, 0xa, 0xf, false);  // row_bcast:15 into rows 1 and 3
    v += __builtin_amdgcn_update_dpp(0, v, 0x143, 0xc, 0xf, false);  // row_bcast:31 into rows 2 and 3
    return v;
}


struct Over { float c0, c1, c2, a; };  // premultiplied composite element
__device__ __forceinline__ Over over(const Over &front, const Over &back) {  // front-to-back "over"
    const float T = 1.0f - front.a;
    Over r;
    r.c0 = fmaf(T, back.c0, front.c0); r.c1 = fmaf(T, back.c1, front.c1); r.c2 = fmaf(T, back.c2, front.c2);
    r.a = fmaf(T, back.a, front.a);
    return r;
}
// ---- cross-lane plumbing: DPP moves (no LDS traffic, a few cycles of latency) ---------------------------
// ctrl: 0x111/0x112/0x114/0x118 = row_shr:1/2/4/8 (within a 16-lane row), 0x142 = row_bcast:15 (lane 15 of a
// row to the next row), 0x143 = row_bcast:31 (lane 31 to the upper half). Lanes without a source keep `old`.
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, false));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xf, 0xf, false); }
template <int CTRL>
__device__ __forceinline__ Over dpp_over(const Over &v) {
    Over r;
    r.c0 = dpp_f<CTRL>(v.c0); r.c1 = dpp_f<CTRL>(v.c1); r.c2 = dpp_f<CTRL>(v.c2); r.a = dpp_f<CTRL>(v.a);
    return r;
}
// Does the DPP source lane of step K exist and lie at or after lane `sl` (the first lane of this lane's segment)?
template <int K>
__device__ __forceinline__ bool scan_src_ok(int lane, int sl) {
    if (K < 4) return (lane & 15) >= (1 << K) && lane - (1 << K) >= sl;
    if (K == 4) return (lane & 16) && ((lane & ~15) - 1) >= sl;
    return lane >= 32 && 31 >= sl;
}
// DPP move whose lanes without a source read 0 (bound_ctrl): the destination needs no initial value, so the
// compiler does not spend a v_mov on it. Every use below ignores what such lanes receive.
template <int CTRL>
__device__ __forceinline__ float dpp0_f(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
template <int CTRL>
__device__ __forceinline__ Over dpp0_over(const Over &v) {
    Over r;
    r.c0 = dpp0_f<CTRL>(v.c0); r.c1 = dpp0_f<CTRL>(v.c1); r.c2 = dpp0_f<CTRL>(v.c2); r.a = dpp0_f<CTRL>(v.a);
    return r;
}
// x = T*x + o in x's own register (the compiler's choice, v_fmac into the DPP temporary, needs a move back per channel)
__device__ __forceinline__ void fma_into(float &x, float T, float o) {
    asm("v_fma_f32 %0, %1, %0, %2" : "+v"(x) : "v"(T), "v"(o));
}
// segmented inclusive scan of "over": segments are runs of lanes sharing sl
__device__ __forceinline__ Over seg_scan_over(Over v, int lane, int sl) {
#define DR_OVER_STEP(CTRL, K)                                                                       \
    {                                                                                               \
        const Over o = dpp0_over<CTRL>(v);                                                          \
        if (scan_src_ok<K>(lane, sl)) {                                                             \
            const float T = 1.0f - o.a;                                                             \
            fma_into(v.c0, T, o.c0); fma_into(v.c1, T, o.c1); fma_into(v.c2, T, o.c2); fma_into(v.a, T, o.a); \
        }                                                                                           \
    }
    DR_OVER_STEP(0x111, 0) DR_OVER_STEP(0x112, 1) DR_OVER_STEP(0x114, 2) DR_OVER_STEP(0x118, 3)
    DR_OVER_STEP(0x142, 4) DR_OVER_STEP(0x143, 5)
#undef DR_OVER_STEP
    return v;
}
// The backward needs the composites only through gC . C and A (the tape-free identity's suffix term): it scans the PAIR
// (w, a), w = gC . c with the lane's own upstream colour gradient (constant over a segment = one ray), under the same
// "over" -- half the scan of the forward's four channels.
struct Over2 { float w, a; };
__device__ __forceinline__ Over2 over2(const Over2 &front, const Over2 &back) {
    const float T = 1.0f - front.a;
    Over2 r;
    r.w = fmaf(T, back.w, front.w); r.a = fmaf(T, back.a, front.a);
    return r;
}
__device__ __forceinline__ Over2 seg_scan_over2(Over2 v, int lane, int sl) {
#define DR_OVER2_STEP(CTRL, K)                                                    \
    {                                                                             \
        const float ow = dpp0_f<CTRL>(v.w), oa = dpp0_f<CTRL>(v.a);               \
        if (scan_src_ok<K>(lane, sl)) {                                           \
            const float T = 1.0f - oa;                                            \
            fma_into(v.w, T, ow); fma_into(v.a, T, oa);                           \
        }                                                                         \
    }
    DR_OVER2_STEP(0x111, 0) DR_OVER2_STEP(0x112, 1) DR_OVER2_STEP(0x114, 2) DR_OVER2_STEP(0x118, 3)
    DR_OVER2_STEP(0x142, 4) DR_OVER2_STEP(0x143, 5)
#undef DR_OVER2_STEP
    return v;
}
// segmented inclusive SUM of NV values (same segment convention)
template <int NV>
__device__ __forceinline__ void seg_scan_sum(float (&v)[NV], int lane, int sl) {
#define DR_SUM_STEP(CTRL, K)                                             \
    {                                                                    \
        const bool ok = scan_src_ok<K>(lane, sl);                        \
        _Pragma("unroll") for (int i = 0; i < NV; ++i) {                 \
            const float t = v[i] + dpp_f<CTRL>(v[i]);                    \
            v[i] = ok ? t : v[i];                                        \
        }                                                                \
    }
    DR_SUM_STEP(0x111, 0) DR_SUM_STEP(0x112, 1) DR_SUM_STEP(0x114, 2) DR_SUM_STEP(0x118, 3)
    DR_SUM_STEP(0x142, 4) DR_SUM_STEP(0x143, 5)
#undef DR_SUM_STEP
}
// segmented inclusive PRODUCT of one value (transmittance of the alpha pre-pass)
__device__ __forceinline__ float seg_scan_prod(float v, int lane, int sl) {
    { const float o = dpp_f<0x111>(v); if (scan_src_ok<0>(lane, sl)) v *= o; }
    { const float o = dpp_f<0x112>(v); if (scan_src_ok<1>(lane, sl)) v *= o; }
    { const float o = dpp_f<0x114>(v); if (scan_src_ok<2>(lane, sl)) v *= o; }
    { const float o = dpp_f<0x118>(v); if (scan_src_ok<3>(lane, sl)) v *= o; }
    { const float o = dpp_f<0x142>(v); if (scan_src_ok<4>(lane, sl)) v *= o; }
    { const float o = dpp_f<0x143>(v); if (scan_src_ok<5>(lane, sl)) v *= o; }
    return v;
}
// inclusive max scan of an int (used to propagate run starts)
__device__ __forceinline__ int scan_max(int v, int lane) {
    { const int o = dpp_i<0x111>(v); if ((lane & 15) >= 1) v = max(v, o); }
    { const int o = dpp_i<0x112>(v); if ((lane & 15) >= 2) v = max(v, o); }
    { const int o = dpp_i<0x114>(v); if ((lane & 15) >= 4) v = max(v, o); }
    { const int o = dpp_i<0x118>(v); if ((lane & 15) >= 8) v = max(v, o); }
    { const int o = dpp_i<0x142>(v); if (lane & 16) v = max(v, o); }
    { const int o = dpp_i<0x143>(v); if (lane >= 32) v = max(v, o); }
    return v;
}
// largest of a non-negative float over the wave, wave-uniform (DPP; non-negative floats order like their bit patterns)
__device__ __forceinline__ float wave_max_nonneg(float x) {
    // unsigned max: 0 (what a lane without a DPP source reads) is its identity, so each step folds into one v_max_u32_dpp
    unsigned int v = __float_as_uint(x);
    v = max(v, (unsigned int)dpp_i<0x111>((int)v));   // row_shr:1
    v = max(v, (unsigned int)dpp_i<0x112>((int)v));
    v = max(v, (unsigned int)dpp_i<0x114>((int)v));
    v = max(v, (unsigned int)dpp_i<0x118>((int)v));   // lane 15 of every row: the row's maximum
    v = max(v, (unsigned int)dpp_i<0x142>((int)v));   // row_bcast:15
    v = max(v, (unsigned int)dpp_i<0x143>((int)v));   // row_bcast:31 -> lane 63: the wave's maximum
    return __uint_as_float((unsigned int)__builtin_amdgcn_readlane((int)v, 63));
}
__device__ __forceinline__ Over readlane_over(const Over &v, int lane) {
    Over r;
    r.c0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.c0), lane));
    r.c1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.c1), lane));
    r.c2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.c2), lane));
    r.a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v.a), lane));
    return r;
}
// whole-wave shifts by one lane with DPP (wave_shr:1 = 0x138, wave_shl:1 = 0x130; gfx9 family): lane 0 / lane 63
// keep the `edge` value. No LDS crossbar traffic, unlike __shfl_up/__shfl_down (ds_bpermute).
__device__ __forceinline__ float wave_up1(float v, float edge) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(edge), __float_as_int(v), 0x138, 0xf, 0xf, false));
}
__device__ __forceinline__ int wave_up1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x138, 0xf, 0xf, false); }
__device__ __forceinline__ int wave_down1(int v, int edge) { return __builtin_amdgcn_update_dpp(edge, v, 0x130, 0xf, 0xf, false); }
__device__ __forceinline__ Over shfl_up1_over(const Over &v) {
    Over r;
    r.c0 = wave_up1(v.c0, 0.f); r.c1 = wave_up1(v.c1, 0.f); r.c2 = wave_up1(v.c2, 0.f); r.a = wave_up1(v.a, 0.f);
    return r;
}

}  // namespace dr
