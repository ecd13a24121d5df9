// march_baseline.hip -- DR_VARIANT_BASELINE march kernels for gfx950: one lane per ray, a wave per
// 8x8 pixel tile, direct global gathers, TF table in LDS, global float atomics for d_vol.
// Kept as the simplest correct device implementation; the optimised kernels are tested against it
// and against the CPU oracle.
//
// Forward   : VR.py:261-306 (raycast) + 363-372 (get_final_image); nondiff: VR.py:308-361.
// Backward  : tape-free equivalent of raycast.grad/get_final_image.grad (VR.py:460-461,470-471).
//   With T_s = 1 - A_{s-1} and q_s = gC.(L_s rgb_s) + gA the adjoint of the sample opacity is
//   d/d op_s = T_s q_s - [gC.(C_final - C_s) + gA.(A_final - A_s)] / (1 - op_s)   (C_s, A_s = composite up to s),
//   which needs only a forward-order walk and the saved forward output (no per-sample tape).
#include "dr_device.h"
#include "dr_kernels.h"
#include "../../include/differender_hip.h"

namespace dr {

template <typename VT>
struct MarchParams {
    VolView<VT> vol; int64_t vol_vs;
    const float4 *tf; int64_t tf_vs; int R; float tf_len;
    const float *cam, *entry, *exit_, *rays; const int32_t *nsamp;
    int W, H, S; float sr, inv_sr;
    float *out; int32_t *steps;
    const float *grad_out, *out_fwd;
    GradView dvol; int64_t dvol_vs;
    float *d_tf; int64_t dtf_vs;
    const uint8_t *only_flagged;
    const unsigned int *ws_mark; unsigned int ws_mark_expect;  // see MarchArgs
    const unsigned int *ws_aux; unsigned int ws_aux_expect;
};

__device__ __forceinline__ bool tile_pixel(int W, int H, int &i, int &j) {
    const int tiles_j = (H + 7) >> 3;
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    i = (wave / tiles_j) * 8 + (lane >> 3);
    j = (wave % tiles_j) * 8 + (lane & 7);
    return i < W && j < H;
}

// TF_LDS: the transfer function is staged in LDS (R <= 10176 entries = 159 KiB: what a workgroup is really granted, less headroom); a larger one is read where it lies, through the
// caches -- the reference has no limit on the TF resolution, and neither have the plain kernels.
template <typename VT, int MODE, bool TF_LDS>
__global__ __launch_bounds__(256) void march_fwd_baseline_kernel(MarchParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_tf_[];
    const int view = blockIdx.y;
    const float4 *tfg = P.tf + view * P.tf_vs;
    if (TF_LDS) {
        for (int k = threadIdx.x; k < P.R; k += 256) lds_tf_[k] = tfg[k];
        __syncthreads();
    }
    const float4 *lds_tf = TF_LDS ? lds_tf_ : tfg;

    int i, j;
    if (!tile_pixel(P.W, P.H, i, j)) return;
    const size_t p = ((size_t)view * P.W + i) * P.H + j;
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    const float cx = P.cam[3 * view], cy = P.cam[3 * view + 1], cz = P.cam[3 * view + 2];
    const f3 light = make_f3(cx + 0.0f, cy + 1.0f, cz + 0.0f);

    RayGeom rg;
    rg.n = P.nsamp[p]; rg.entry = P.entry[p]; rg.exit_ = P.exit_[p];
    rg.vx = P.rays[3 * p]; rg.vy = P.rays[3 * p + 1]; rg.vz = P.rays[3 * p + 2];
    rg.t0 = rg.entry + 0.5f * (rg.exit_ - rg.entry) / (float)rg.n;
    const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
    const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;

    float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;
    int cnt = 0;
    for (int s = 0; s < nmarch; ++s) {
        if (!(A < 0.99f)) break;
        Sample sm;
        sample_pos(rg, cx, cy, cz, s, sm.px, sm.py, sm.pz);
        classify(vol, lds_tf, P.R, P.tf_len, P.inv_sr, sm);
        ++cnt;
        if (MODE == DR_MODE_NONDIFF && !(sm.a > 1e-3f)) continue;
        shade(vol, light, vd, MODE == DR_MODE_DIFF, sm);
        const float T = 1.0f - A;
        C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
        C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
        C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
        A = fmaf(T, sm.op, A);
    }
    if (MODE == DR_MODE_NONDIFF) {
        C0 = fminf(1.0f, C0); C1 = fminf(1.0f, C1); C2 = fminf(1.0f, C2); A = fminf(1.0f, A);
    }
    reinterpret_cast<float4 *>(P.out)[p] = make_float4(C0, C1, C2, A);
    if (P.steps) P.steps[p] = cnt;
}

__device__ __forceinline__ float finite_or_zero(float x) { return (x == x) ? fminf(fmaxf(x, -1.0e30f), 1.0e30f) : 0.0f; }

// TABLES: what the workgroup keeps in LDS --
//   2: [R] TF + [R][4] d_tf accumulators in double (48 R bytes: R <= 3392);
//   1: the TF only; d_tf contributions go straight to the caller's tensor with float atomics (16 R bytes: R <= 10176; at such
//      resolutions a texel collects few samples of a workgroup, so the summation noise the double table exists for -- below -- is
//      not a concern);
//   0: nothing: the TF is read where it lies (any R; the reference has no limit on the TF resolution).
template <typename VT, int TABLES>
__global__ __launch_bounds__(256) void march_bwd_baseline_kernel(MarchParams<VT> P) {
    // TABLES == 2: [R] TF, then [R][4] dTF accumulators in DOUBLE: thousands of samples of a workgroup land on a few texels (all of them on
    // one when R = 1), and f32 atomics in thread order put 1e-3 of summation noise on d_tf at sampling rate 16 (fuzz seed 603239,
    // round 4); the oracle sums d_tf in double for the same reason (its d_tf accumulators), so this keeps the twin a twin
    extern __shared__ __attribute__((aligned(16))) float4 lds_tf_[];
    const int view = blockIdx.y;
    const float4 *tfg = P.tf + view * P.tf_vs;
    double *lds_dtf = reinterpret_cast<double *>(lds_tf_ + P.R);
    if (TABLES >= 1) for (int k = threadIdx.x; k < P.R; k += 256) lds_tf_[k] = tfg[k];
    if (TABLES == 2) for (int k = threadIdx.x; k < 4 * P.R; k += 256) lds_dtf[k] = 0.0;
    if (TABLES >= 1) __syncthreads();
    const float4 *lds_tf = TABLES >= 1 ? lds_tf_ : tfg;
    float *dtf_g = P.d_tf ? P.d_tf + view * P.dtf_vs * 4 : nullptr;

    int i, j;
    bool active = tile_pixel(P.W, P.H, i, j);
    // second pass of the brick-centric backward: only the rays the forward flagged -- unless the workspace is not that
    // forward's (uniform): then the flags mean nothing, B1 has done nothing, and every ray is marched here
    const bool stale_ws = (P.ws_mark != nullptr && *P.ws_mark != P.ws_mark_expect) || (P.ws_aux != nullptr && *P.ws_aux != P.ws_aux_expect);
    if (stale_ws && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0)
        atomicAdd(const_cast<unsigned int *>(P.ws_mark) + 6, 1u);   // header word 9 (ST_STALE_BWD): the host layer warns
    // (flag 1: irregular rays; 2 = rays of the exact pass, whose backward is ray_exact_bwd_kernel's)
    if (active && P.only_flagged && !stale_ws) active = P.only_flagged[((size_t)view * P.W + i) * P.H + j] == 1;
    if (active) {
        const size_t p = ((size_t)view * P.W + i) * P.H + j;
        VolView<VT> vol = P.vol;
        vol.p += view * P.vol_vs;
        GradView dv = P.dvol;
        const bool want_vol = dv.p != nullptr;
        const bool want_tf = P.d_tf != nullptr;
        if (want_vol) dv.p += view * P.dvol_vs;
        const float cx = P.cam[3 * view], cy = P.cam[3 * view + 1], cz = P.cam[3 * view + 2];
        const f3 light = make_f3(cx + 0.0f, cy + 1.0f, cz + 0.0f);

        RayGeom rg;
        rg.n = P.nsamp[p]; rg.entry = P.entry[p]; rg.exit_ = P.exit_[p];
        rg.vx = P.rays[3 * p]; rg.vy = P.rays[3 * p + 1]; rg.vz = P.rays[3 * p + 2];
        rg.t0 = rg.entry + 0.5f * (rg.exit_ - rg.entry) / (float)rg.n;
        const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
        const int nmarch = rg.n > P.S ? P.S : rg.n;

        const float4 go = reinterpret_cast<const float4 *>(P.grad_out)[p];
        const float4 of = reinterpret_cast<const float4 *>(P.out_fwd)[p];
        const float delta = 1e-3f;

        float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;
        for (int s = 0; s < nmarch; ++s) {
            if (!(A < 0.99f)) break;
            Sample sm;
            sample_pos(rg, cx, cy, cz, s, sm.px, sm.py, sm.pz);
            classify(vol, lds_tf, P.R, P.tf_len, P.inv_sr, sm);
            shade(vol, light, vd, true, sm);
            const float T = 1.0f - A;
            C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
            C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
            C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
            A = fmaf(T, sm.op, A);
            const bool last = (s == nmarch - 1) || !(A < 0.99f);
            // what the samples after s contribute to the loss: gC.(C_final - C_s) + gA.(A_final - A_s)
            const float suffix = (go.x * (of.x - C0) + go.y * (of.y - C1) + go.z * (of.z - C2)) + go.w * (of.w - A);
            SampleAdj ad;
            sample_adjoint(sm, vd, T, suffix, last, go, P.inv_sr, ad);
            float I_bar = want_vol ? intensity_adjoint(sm, lds_tf[sm.lo], lds_tf[sm.hi], ad, P.tf_len) : 0.0f;
            if (P.only_flagged) {
                // second pass of the brick-centric backward (irregular rays): that path promises finite gradients, so a NaN
                // adjoint is dropped and infinities are clamped here too (dr_march_bwd_variant). On its own, the baseline
                // propagates NaN like the reference.
                ad.r_bar = finite_or_zero(ad.r_bar); ad.g_bar = finite_or_zero(ad.g_bar); ad.b_bar = finite_or_zero(ad.b_bar);
                ad.a_bar = finite_or_zero(ad.a_bar); ad.gx = finite_or_zero(ad.gx); ad.gy = finite_or_zero(ad.gy);
                ad.gz = finite_or_zero(ad.gz); I_bar = finite_or_zero(I_bar);
            }

            if (want_tf && TABLES < 2) {
                const float w0 = 1.0f - sm.fr, w1 = sm.fr;
                float *d0 = dtf_g + 4 * sm.lo, *d1 = dtf_g + 4 * sm.hi;
                const float v8[8] = {w0 * ad.r_bar, w0 * ad.g_bar, w0 * ad.b_bar, w0 * ad.a_bar, w1 * ad.r_bar, w1 * ad.g_bar, w1 * ad.b_bar, w1 * ad.a_bar};
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    if (P.only_flagged) { atomic_add_sat(d0 + q, v8[q]); atomic_add_sat(d1 + q, v8[4 + q]); }
                    else { unsafeAtomicAdd(d0 + q, v8[q]); unsafeAtomicAdd(d1 + q, v8[4 + q]); }
                }
            } else if (want_tf) {
                const float w0 = 1.0f - sm.fr, w1 = sm.fr;
                double *d0 = lds_dtf + 4 * sm.lo, *d1 = lds_dtf + 4 * sm.hi;   // (the products are f32, as in the oracle)
                atomicAdd(d0 + 0, (double)(w0 * ad.r_bar)); atomicAdd(d0 + 1, (double)(w0 * ad.g_bar));
                atomicAdd(d0 + 2, (double)(w0 * ad.b_bar)); atomicAdd(d0 + 3, (double)(w0 * ad.a_bar));
                atomicAdd(d1 + 0, (double)(w1 * ad.r_bar)); atomicAdd(d1 + 1, (double)(w1 * ad.g_bar));
                atomicAdd(d1 + 2, (double)(w1 * ad.b_bar)); atomicAdd(d1 + 3, (double)(w1 * ad.a_bar));
            }
            if (want_vol) {
                tri_scatter_global(vol, dv, sm.px, sm.py, sm.pz, I_bar);
                if (!sm.flat) {
                    tri_scatter_global(vol, dv, sm.px + delta, sm.py, sm.pz, ad.gx);
                    tri_scatter_global(vol, dv, sm.px - delta, sm.py, sm.pz, -ad.gx);
                    tri_scatter_global(vol, dv, sm.px, sm.py + delta, sm.pz, ad.gy);
                    tri_scatter_global(vol, dv, sm.px, sm.py - delta, sm.pz, -ad.gy);
                    tri_scatter_global(vol, dv, sm.px, sm.py, sm.pz + delta, ad.gz);
                    tri_scatter_global(vol, dv, sm.px, sm.py, sm.pz - delta, -ad.gz);
                }
            }
        }
    }
    if (P.d_tf && TABLES == 2) {
        __syncthreads();
        float *dtf = dtf_g;
        for (int k = threadIdx.x; k < 4 * P.R; k += 256) {
            const float v = (float)lds_dtf[k];
            if (v == 0.0f) continue;
            if (P.only_flagged) atomic_add_sat(dtf + k, v);  // second pass of the sanitising backward: the sum stays finite
            else unsafeAtomicAdd(dtf + k, v);
        }
    }
}

template <typename VT>
static MarchParams<VT> make_params(const MarchArgs &a) {
    MarchParams<VT> P;
    P.vol.p = static_cast<const VT *>(a.vol);
    P.vol.sx = a.sx; P.vol.sy = a.sy; P.vol.sz = a.sz;
    P.vol.VX = a.VX; P.vol.VY = a.VY; P.vol.VZ = a.VZ;
    P.vol.scx = (float)((double)a.VX - 1.0 - 1e-4);
    P.vol.scy = (float)((double)a.VY - 1.0 - 1e-4);
    P.vol.scz = (float)((double)a.VZ - 1.0 - 1e-4);
    P.vol_vs = a.vol_vs;
    P.tf = reinterpret_cast<const float4 *>(a.tf); P.tf_vs = a.tf_vs / 4; P.R = a.R; P.tf_len = (float)(a.R - 1);
    P.cam = a.cam; P.entry = a.entry; P.exit_ = a.exit_; P.rays = a.rays; P.nsamp = a.nsamp;
    P.W = a.W; P.H = a.H; P.S = a.S; P.sr = a.sr; P.inv_sr = 1.0f / a.sr;
    P.out = a.out; P.steps = a.steps;
    P.grad_out = a.grad_out; P.out_fwd = a.out_fwd;
    P.dvol.p = a.d_vol; P.dvol.sx = a.dsx; P.dvol.sy = a.dsy; P.dvol.sz = a.dsz; P.dvol_vs = a.dvol_vs;
    P.d_tf = a.d_tf; P.dtf_vs = a.dtf_vs / 4;
    P.only_flagged = a.only_flagged;
    P.ws_mark = a.ws_mark; P.ws_mark_expect = a.ws_mark_expect;
    P.ws_aux = a.ws_aux; P.ws_aux_expect = a.ws_aux_expect;
    return P;
}

hipError_t allow_lds_impl(const void *kernel, size_t bytes);  // capi.hip
template <typename K>
static hipError_t big_lds(K kernel, size_t bytes) {  // dynamic LDS above 64 KB needs an explicit opt-in (once per kernel)
    if (bytes <= 64 * 1024) return hipSuccess;
    return allow_lds_impl(reinterpret_cast<const void *>(kernel), bytes);
}

// What one workgroup may really ask for: the runtime does not grant a CU's full 160 KiB (163 232 B launched, 163 616 B did not:
// tools/lds_limit_probe.py, round 5) -- 1 KiB of headroom, as brick_path_supported keeps. A table that does not fit drops to the
// next tier instead of failing at launch (R = 3409 .. 3413 and 10 227 .. 10 240 picked a tier the launch refused).
constexpr size_t LDS_PER_CU = 159 * 1024;
static dim3 tile_grid(const MarchArgs &a) {
    const int tiles = ((a.W + 7) / 8) * ((a.H + 7) / 8);
    return dim3((tiles + 3) / 4, a.n_views);
}

template <typename VT>
static int fwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    size_t lds = (size_t)a.R * sizeof(float4);
    const bool tf_lds = lds <= LDS_PER_CU;   // (R <= 10176; a larger TF is read from global memory)
    if (!tf_lds) lds = 0;
    MarchParams<VT> P = make_params<VT>(a);
#define DR_FWD_BASE(MODE_, LDS_)                                                                                             \
    {                                                                                                                        \
        if (big_lds(march_fwd_baseline_kernel<VT, MODE_, LDS_>, lds) != hipSuccess) return DR_EUNSUPPORTED;                  \
        hipLaunchKernelGGL((march_fwd_baseline_kernel<VT, MODE_, LDS_>), tile_grid(a), dim3(256), lds, stream, P);           \
    }
    if (a.mode == DR_MODE_DIFF) { if (tf_lds) DR_FWD_BASE(DR_MODE_DIFF, true) else DR_FWD_BASE(DR_MODE_DIFF, false) }
    else { if (tf_lds) DR_FWD_BASE(DR_MODE_NONDIFF, true) else DR_FWD_BASE(DR_MODE_NONDIFF, false) }
#undef DR_FWD_BASE
    return (int)hipGetLastError();
}

int launch_march_fwd_baseline(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? fwd_dispatch<__half>(a, stream) : fwd_dispatch<float>(a, stream);
}

template <typename VT>
static int bwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    // TF + its double-precision gradient table in LDS while they fit (R <= 3392), then the TF alone (R <= 10176), then nothing
    const size_t lds2 = (size_t)a.R * (sizeof(float4) + 4 * sizeof(double)), lds1 = (size_t)a.R * sizeof(float4);
    const int tables = lds2 <= LDS_PER_CU ? 2 : (lds1 <= LDS_PER_CU ? 1 : 0);
    const size_t lds = tables == 2 ? lds2 : (tables == 1 ? lds1 : 0);
    MarchParams<VT> P = make_params<VT>(a);
#define DR_BWD_BASE(T_)                                                                                          \
    {                                                                                                            \
        if (big_lds(march_bwd_baseline_kernel<VT, T_>, lds) != hipSuccess) return DR_EUNSUPPORTED;               \
        hipLaunchKernelGGL((march_bwd_baseline_kernel<VT, T_>), tile_grid(a), dim3(256), lds, stream, P);        \
    }
    if (tables == 2) DR_BWD_BASE(2) else if (tables == 1) DR_BWD_BASE(1) else DR_BWD_BASE(0)
#undef DR_BWD_BASE
    return (int)hipGetLastError();
}

int launch_march_bwd_baseline(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? bwd_dispatch<__half>(a, stream) : bwd_dispatch<float>(a, stream);
}

}  // namespace dr
