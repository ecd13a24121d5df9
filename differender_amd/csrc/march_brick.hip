// march_brick.hip -- brick-centric march kernels for gfx950 (DR_VARIANT_AUTO fast path).
//
// Why: one marched sample needs 56 voxel fetches (7 trilinear taps, VR.py:153-203). Served as global
// gathers they cost >= 25 cycles per wave-load on a CU (measured, tools/microbench); from LDS they cost
// 2-6. So the volume is processed brick by brick: a workgroup stages one 16^3-cell brick (+apron, 19^3
// voxels, 27 KB) in LDS with coalesced reads and marches every ray segment that crosses it.
//
//   F1 brick_fwd_kernel   per (brick, view): LDS box + TF; list the ray segments inside the brick, sort
//                         them by length (so the 64 lanes of a wave run segments of equal length), march
//                         them from LDS, write one partial RGBA + sample count per (ray, layer).
//   F2 ray_compose_kernel per ray: composite the partials front to back ("over" is associative); the one
//                         segment in which alpha crosses 0.99 is re-marched sample by sample so early
//                         termination stays exact (VR.py:267); irregular rays are marched whole.
//                         Leaves the prefix (C,A) before every segment in the workspace for the backward.
//   B1 brick_bwd_kernel   per (brick, view): recompute the segment from the stored prefix, scatter-add
//                         d_volume into an LDS box and d_tf into an LDS table, flush each once with
//                         global float atomics (one coalesced flush per brick instead of 56 atomics/sample).
//                         The LDS accumulators are 64-bit FIXED POINT: on gfx950 ds_add_f32 is serialised
//                         (~193 cycles per wave-instruction whatever the addresses, measured in
//                         tools/microbench/lds_atomic_bench) while ds_add_u64 takes 9-12. The scale comes from
//                         max|grad_out| (a small reduction kernel) so 2^-28 of it is the resolution and
//                         2^34 of it the headroom; integer adds also make the brick sums order-independent.
//   B2                    irregular rays: the baseline backward restricted to flagged rays.
//
// Reference functions replaced: raycast / raycast_nondiff / get_final_image[_nondiff] (VR.py:261-372) and
// their Taichi-autodiff twins (VR.py:460-461,470-471).
#include "dr_brick_common.h"

namespace dr {

// ------------------------------------------------------------------------------------------------ F1
template <typename VT, int MODE>
__global__ __launch_bounds__(256) void brick_fwd_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int view = blockIdx.y;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    BrickCtx c;
    brick_setup(P, blockIdx.x, cam, c);
    if (c.i0 > c.i1 || c.j0 > c.j1) return;  // uniform: the brick projects outside the image

    LdsLayout L = carve(smem, P.R, false, false);
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    load_tf_and_box(P, vol, c, P.tf + view * P.tf_vs, L);
    const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
    const int NP = P.W * P.H;
    const size_t seg_base = ((size_t)view * P.g.NL + c.layer) * NP;
    const int ncand = (c.i1 - c.i0 + 1) * (c.j1 - c.j0 + 1);

    for (int cbase = 0; cbase < ncand; cbase += ECHUNK) {
        const int nE = build_entries<VT, false>(P, c, cam, view, cbase, ncand, MODE, L);  // syncs inside
        for (int e = threadIdx.x; e < nE; e += 256) {
            const int ei = L.order[e];
            const int pl = L.e_pix[ei], s0 = L.e_s0[ei], cnt = L.e_cnt[ei];
            RayGeom rg;
            load_ray(P.entry, P.exit_, P.rays, P.nsamp, (size_t)view * NP + pl, rg);
            const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
            float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;
            int valid = 0;
            for (int s = s0; s < s0 + cnt; ++s) {
                Sample sm; TapCoords t;
                if (!sample_coords(vol, c, rg, cam, s, sm, t)) continue;
                ++valid;
                float dx, dy, dz;
                sample_taps_lds(L.box, t, sm.I, dx, dy, dz);
                classify_from_I(L.tf, P.R, P.tf_len, P.inv_sr, sm);
                if (MODE == DR_MODE_NONDIFF && !(sm.a > 1e-3f)) continue;
                shade_from_grad<true>(dx, dy, dz, light, vd, MODE == DR_MODE_DIFF, sm);
                const float T = 1.0f - A;
                C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
                C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
                C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
                A = fmaf(T, sm.op, A);
            }
            if (valid > 0) {
                P.seg_rgba[seg_base + pl] = make_float4(C0, C1, C2, A);
                P.seg_cnt[seg_base + pl] = (uint16_t)min(valid, 65535);
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------------------------------------ F2
// Layer (dr_brick.h) of the brick that holds sample s of a ray. Layers grow monotonically along a ray, so the
// bricks of samples [0, n) all have layers between those of sample 0 and sample n-1: the per-ray passes only look
// at that range of the [layer][pixel] workspace (typically 45-60 of 127 layers at 512^3).
template <typename VT>
__device__ __forceinline__ int sample_layer(const BrickParams<VT> &P, const RayGeom &rg, f3 cam, int s) {
    float px, py, pz, fr;
    int x0, y0, z0;
    sample_pos(rg, cam.x, cam.y, cam.z, s, px, py, pz);
    axis_coord(px, P.vol.scx, x0, fr); axis_coord(py, P.vol.scy, y0, fr); axis_coord(pz, P.vol.scz, z0, fr);
    const int cbx = cam_brick(cam.x, P.vol.scx), cby = cam_brick(cam.y, P.vol.scy), cbz = cam_brick(cam.z, P.vol.scz);
    const int lmin = axis_layer_min(cbx, P.g.NBx) + axis_layer_min(cby, P.g.NBy) + axis_layer_min(cbz, P.g.NBz);
    return abs(x0 / BRK - cbx) + abs(y0 / BRK - cby) + abs(z0 / BRK - cbz) - lmin;
}

template <typename VT, int MODE>
__global__ __launch_bounds__(256) void ray_compose_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_tf[];
    const int view = blockIdx.y;
    const float4 *tfg = P.tf + view * P.tf_vs;
    for (int k = threadIdx.x; k < P.R; k += 256) lds_tf[k] = tfg[k];
    __syncthreads();
    const int NP = P.W * P.H;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= NP) return;
    const size_t p = (size_t)view * NP + pl;
    RayGeom rg;
    load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
    float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;
    int steps = 0;
    uint8_t flag = 0;
    if (rg.n > 0) {
        VolView<VT> vol = P.vol;
        vol.p += view * P.vol_vs;
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
        const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
        int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
        const int nfull = nmarch;
        const size_t seg0 = (size_t)view * P.g.NL * NP + pl;
        bool regular = ray_is_regular(rg.n, rg.entry);
        // with an alpha pre-pass the bricks marched exactly the live samples of the ray: no crossing to look for
        const bool use_live = P.use_live && P.stats[2 + view] != 0u;
        if (regular && use_live) nmarch = min(nmarch, P.ws_steps[p]);
        int l_lo = 0, l_hi = P.g.NL - 1;
        if (regular && nmarch > 0) {
            l_lo = max(sample_layer(P, rg, cam, 0), 0);
            l_hi = min(sample_layer(P, rg, cam, nmarch - 1), P.g.NL - 1);
        }
        if (regular) {
            // safety net: the segments must account for every sample, else march this ray whole
            int total = 0;
            for (int l = l_lo; l <= l_hi; ++l) total += P.seg_cnt[seg0 + (size_t)l * NP];
            if (total != nmarch) { regular = false; nmarch = nfull; atomicAdd(&P.stats[0], 1u); }
        }
        int s_from = 0, s_to = 0;  // samples to march one by one with early termination
        if (!regular) {
            flag = 1; s_to = nmarch;
        } else {
            int sacc = 0;
            steps = nmarch;
            for (int l = l_lo; l <= l_hi; ++l) {
                const size_t si = seg0 + (size_t)l * NP;
                const int cnt = P.seg_cnt[si];
                if (cnt == 0) continue;
                const float4 sg = P.seg_rgba[si];
                if (MODE == DR_MODE_DIFF) P.seg_rgba[si] = make_float4(C0, C1, C2, A);  // prefix for the backward
                const float T = 1.0f - A;
                const float A_after = fmaf(T, sg.w, A);
                if (!use_live && !(A_after < 0.99f)) {  // alpha crosses 0.99 inside this segment
                    s_from = sacc; s_to = sacc + cnt;
                    break;
                }
                C0 = fmaf(T, sg.x, C0); C1 = fmaf(T, sg.y, C1); C2 = fmaf(T, sg.z, C2);
                A = A_after;
                sacc += cnt;
            }
        }
        if (s_to > s_from) {
            steps = s_from;
            for (int s = s_from; s < s_to; ++s) {
                if (!(A < 0.99f)) break;
                Sample sm;
                sample_pos(rg, cam.x, cam.y, cam.z, s, sm.px, sm.py, sm.pz);
                classify(vol, lds_tf, P.R, P.tf_len, P.inv_sr, sm);
                ++steps;
                if (MODE == DR_MODE_NONDIFF && !(sm.a > 1e-3f)) continue;
                shade(vol, light, vd, MODE == DR_MODE_DIFF, sm);
                const float T = 1.0f - A;
                C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
                C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
                C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
                A = fmaf(T, sm.op, A);
            }
            // a regular ray that crossed 0.99 exactly on the last sample of its segment keeps marching
            // nothing further: later segments are ignored (their first sample would see A >= 0.99).
            if (!flag && A < 0.99f) {
                // the partial composite crossed 0.99 but the exact recurrence did not (rounding): fall
                // back to marching the rest of the ray sample by sample.
                for (int s = s_to; s < nmarch; ++s) {
                    if (!(A < 0.99f)) break;
                    Sample sm;
                    sample_pos(rg, cam.x, cam.y, cam.z, s, sm.px, sm.py, sm.pz);
                    classify(vol, lds_tf, P.R, P.tf_len, P.inv_sr, sm);
                    ++steps;
                    if (MODE == DR_MODE_NONDIFF && !(sm.a > 1e-3f)) continue;
                    shade(vol, light, vd, MODE == DR_MODE_DIFF, sm);
                    const float T = 1.0f - A;
                    C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
                    C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
                    C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
                    A = fmaf(T, sm.op, A);
                }
                flag = 1;  // its stored prefixes beyond s_to are stale: let B2 handle the whole ray
            }
        }
    }
    if (MODE == DR_MODE_NONDIFF) {
        C0 = fminf(1.0f, C0); C1 = fminf(1.0f, C1); C2 = fminf(1.0f, C2); A = fminf(1.0f, A);
    }
    reinterpret_cast<float4 *>(P.out)[p] = make_float4(C0, C1, C2, A);
    if (P.steps) P.steps[p] = steps;
    P.ws_steps[p] = steps;
    P.rayflag[p] = flag;
}

// ------------------------------------------------------------------------------------------------ P2
// Alpha pre-pass, per ray: accumulated alpha does not depend on lighting (A_s = A_{s-1} + (1-A_{s-1}) op_s with
// op_s a function of the centre tap only), so the sample at which a ray terminates can be found from alpha-only
// partials at a fraction of the cost. Composes the alpha partials of the bricks front to back; the segment in which
// alpha crosses 0.99 is re-marched sample by sample (centre tap from global memory) -- the same decisions, in the
// same arithmetic, as ray_compose_kernel would take. Writes ws_steps[p] = exact number of live samples.
template <typename VT, int MODE>
__global__ __launch_bounds__(256) void ray_alpha_kernel(BrickParams<VT> P) {
    const int view = blockIdx.y;
    if (P.stats[2 + view] == 0u) return;  // uniform: no ray of this view can terminate, nothing to find
    const int NP = P.W * P.H;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= NP) return;
    const size_t p = (size_t)view * NP + pl;
    if (!P.pp_first && P.ws_steps[p] == -1) return;  // crossing found in an earlier phase
    RayGeom rg;
    load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
    const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
    if (ray_is_regular(rg.n, rg.entry)) {
        // state carried from phase to phase in the (not yet written) output buffer: alpha so far, samples so far
        float4 *park = reinterpret_cast<float4 *>(P.out) + p;
        float A = 0.f;
        int sacc = 0;
        if (!P.pp_first) { const float4 st = *park; A = st.x; sacc = __float_as_int(st.y); }
        const size_t seg0 = (size_t)view * P.g.NL * NP + pl;
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        const int l_lo = max(max(sample_layer(P, rg, cam, 0), 0), P.pp_l0);
        const int l_hi = min(min(sample_layer(P, rg, cam, nmarch - 1), P.g.NL - 1), P.pp_l1 - 1);
        for (int l = l_lo; l <= l_hi; ++l) {
            const size_t si = seg0 + (size_t)l * NP;
            const int cnt = P.seg_cnt[si];
            if (cnt == 0) continue;
            const float A_after = fmaf(1.0f - A, P.seg_rgba[si].w, A);
            if (!(A_after < 0.99f - 1e-5f)) {  // the crossing segment (with a margin for the re-associated partials)
                                              // starts at sample sacc: resolved by ray_cross_kernel
                *park = make_float4(A, __int_as_float(sacc), 0.f, 0.f);
                P.ws_steps[p] = -1;
                return;
            }
            A = A_after;
            sacc += cnt;
        }
        *park = make_float4(A, __int_as_float(sacc), 0.f, 0.f);
    }
    if (P.pp_first) P.ws_steps[p] = nmarch;  // alive (so far): every planned sample is live
}

// P2b: the rays whose accumulated alpha crosses 0.99 (ws_steps == -1, crossing segment and alpha before it parked in
// the not-yet-written output buffer). One WAVE per ray: 64 consecutive samples per pass -- positions, centre taps (the
// lanes read neighbouring voxels) and TF lookups in parallel, then the exact sequential recurrence
// A <- fma(1 - A, op_s, A) of VR.py:318-349 over the 64 opacities (a one-thread-per-ray loop spent ~370 dependent
// global gathers per ray at sampling rate 8). Passes whose total transmittance keeps alpha clear of the threshold are
// skipped with the (re-associated) wave product; the pass that can cross is evaluated in F2's sequential arithmetic.
template <typename VT, int MODE>
__global__ __launch_bounds__(256) void ray_cross_kernel(BrickParams<VT> P) {
    const int view = blockIdx.y;
    if (P.stats[2 + view] == 0u) return;  // uniform
    const int NP = P.W * P.H;
    const int lane = threadIdx.x & 63;
    // waves walk over the rays (a bounded grid: gated off, the launch costs a few thousand workgroup exits)
    for (int pl = blockIdx.x * 4 + (threadIdx.x >> 6); pl < NP; pl += 4 * (int)gridDim.x) {  // wave-uniform
    const size_t p = (size_t)view * NP + pl;
    if (P.ws_steps[p] != -1) continue;  // wave-uniform: no crossing to resolve
    const float4 parked = reinterpret_cast<const float4 *>(P.out)[p];
    RayGeom rg;
    load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
    const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    const float4 *tfg = P.tf + view * P.tf_vs;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    // Round 1 starts at the crossing segment with the alpha the (re-associated) partial composites give, ~1e-7 off
    // the sequential value. If the decision it reaches is closer to the threshold than that error could matter
    // (2e-6), round 2 repeats the recurrence from the first sample in exact sequential arithmetic -- the decision
    // is then the oracle's, bit for bit (deviation D3 of DESIGN.md does not arise on this path).
    float A = parked.x;
    int s = __float_as_int(parked.y);
    for (int round = 0; round < 2; ++round) {
        float A_prev = A;
        bool done = false;
        for (int base = s; base < nmarch && !done; base += 64) {
            const int sl = base + lane;
            float op = 0.0f;
            if (sl < nmarch) {
                Sample sm;
                sample_pos(rg, cam.x, cam.y, cam.z, sl, sm.px, sm.py, sm.pz);
                classify(vol, tfg, P.R, P.tf_len, P.inv_sr, sm);
                op = (MODE == DR_MODE_NONDIFF && !(sm.a > 1e-3f)) ? 0.0f : sm.op;  // skipped sample: A unchanged (fma(T, 0, A) == A)
            }
            const int cnt = min(64, nmarch - base);
            if (round == 0) {
                // Transmittance of the whole pass (wave product): if even its end stays clear of the threshold (by far
                // more than re-association can move it) no sample of the pass terminates the ray: skip the sequential part.
                float Tw = 1.0f - op;  // inactive lanes: op = 0
                for (int o = 32; o > 0; o >>= 1) Tw *= __shfl_xor(Tw, o);
                const float A_end = fmaf(1.0f - A, 1.0f - Tw, A);
                if (A < 0.99f && A_end < 0.99f - 1e-4f) { A_prev = A = A_end; s += cnt; continue; }  // uniform
            }
            for (int i = 0; i < cnt; ++i) {  // uniform
                if (!(A < 0.99f)) { done = true; break; }
                const float opi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(op), i));
                A_prev = A;
                A = fmaf(1.0f - A, opi, A);
                ++s;
            }
        }
        // decided at A (>= 0.99, or the ray ran out of samples) with A_prev (< 0.99) before it
        const bool ambiguous = fabsf(A - 0.99f) < 2e-6f || fabsf(A_prev - 0.99f) < 2e-6f;
        if (round == 1 || !ambiguous) break;  // uniform
        A = 0.0f; s = 0;
    }
    if (lane == 0) P.ws_steps[p] = s;
    }
}

// ------------------------------------------------------------------------------------------------ B1
template <typename VT, bool WANT_VOL, bool WANT_TF>
__global__ __launch_bounds__(256) void brick_bwd_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int view = blockIdx.y;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    BrickCtx c;
    brick_setup(P, blockIdx.x, cam, c);
    if (c.i0 > c.i1 || c.j0 > c.j1) return;

    LdsLayout L = carve(smem, P.R, WANT_VOL, WANT_TF);
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    load_tf_and_box(P, vol, c, P.tf + view * P.tf_vs, L);
    if (WANT_VOL) for (int k = threadIdx.x; k < BOX_LDS; k += 256) L.dbox[k] = 0ull;
    if (WANT_TF) for (int k = threadIdx.x; k < 4 * P.R; k += 256) L.dtf[k] = 0ull;
    const FixScale fs = make_fix_scale(P.stats[1]);
    const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
    const int NP = P.W * P.H;
    const size_t seg_base = ((size_t)view * P.g.NL + c.layer) * NP;
    const int ncand = (c.i1 - c.i0 + 1) * (c.j1 - c.j0 + 1);
    bool any = false;

    for (int cbase = 0; cbase < ncand; cbase += ECHUNK) {
        const int nE = build_entries<VT, true>(P, c, cam, view, cbase, ncand, DR_MODE_DIFF, L);
        any = any || nE > 0;
        for (int e = threadIdx.x; e < nE; e += 256) {
            const int ei = L.order[e];
            const int pl = L.e_pix[ei], s0 = L.e_s0[ei], cnt = L.e_cnt[ei];
            const size_t p = (size_t)view * NP + pl;
            if (P.seg_cnt[seg_base + pl] == 0) continue;  // no sample of this ray lies in this brick
            RayGeom rg;
            load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
            const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
            const int live = P.ws_steps[p];
            const float4 pre = P.seg_rgba[seg_base + pl];  // composite before this segment (written by F2)
            const float4 go = reinterpret_cast<const float4 *>(P.grad_out)[p];
            const float4 of = reinterpret_cast<const float4 *>(P.out_fwd)[p];
            float C0 = pre.x, C1 = pre.y, C2 = pre.z, A = pre.w;
            // d_tf: consecutive samples of a ray mostly fall between the same two texels, so the eight
            // texel contributions are summed in registers and only flushed when the texel pair changes
            int tf_lo = -1, tf_hi = -1;
            float t0r = 0.f, t0g = 0.f, t0b = 0.f, t0a = 0.f, t1r = 0.f, t1g = 0.f, t1b = 0.f, t1a = 0.f;
            for (int s = s0; s < s0 + cnt; ++s) {
                Sample sm; TapCoords t;
                if (!sample_coords(vol, c, rg, cam, s, sm, t)) continue;
                float dx, dy, dz;
                sample_taps_lds(L.box, t, sm.I, dx, dy, dz);
                classify_from_I(L.tf, P.R, P.tf_len, P.inv_sr, sm);
                shade_from_grad<true>(dx, dy, dz, light, vd, true, sm);
                const float T = 1.0f - A;
                C0 = fmaf(T, sm.L * sm.r * sm.op, C0);
                C1 = fmaf(T, sm.L * sm.g * sm.op, C1);
                C2 = fmaf(T, sm.L * sm.b * sm.op, C2);
                A = fmaf(T, sm.op, A);
                const bool last = (s == live - 1);
                const float suffix = (go.x * (of.x - C0) + go.y * (of.y - C1) + go.z * (of.z - C2)) + go.w * (of.w - A);
                SampleAdj ad;
                sample_adjoint<true>(sm, vd, T, suffix, last, go, P.inv_sr, ad);
                if (WANT_TF) {
                    if (sm.lo != tf_lo) {
                        if (tf_lo >= 0) {
                            fix_add(L.dtf + 4 * tf_lo + 0, fix_clamp(t0r, fs), fs); fix_add(L.dtf + 4 * tf_lo + 1, fix_clamp(t0g, fs), fs);
                            fix_add(L.dtf + 4 * tf_lo + 2, fix_clamp(t0b, fs), fs); fix_add(L.dtf + 4 * tf_lo + 3, fix_clamp(t0a, fs), fs);
                            fix_add(L.dtf + 4 * tf_hi + 0, fix_clamp(t1r, fs), fs); fix_add(L.dtf + 4 * tf_hi + 1, fix_clamp(t1g, fs), fs);
                            fix_add(L.dtf + 4 * tf_hi + 2, fix_clamp(t1b, fs), fs); fix_add(L.dtf + 4 * tf_hi + 3, fix_clamp(t1a, fs), fs);
                        }
                        tf_lo = sm.lo; tf_hi = sm.hi;
                        t0r = t0g = t0b = t0a = t1r = t1g = t1b = t1a = 0.f;
                    }
                    const float w0 = 1.0f - sm.fr, w1 = sm.fr;
                    t0r = fmaf(w0, ad.r_bar, t0r); t0g = fmaf(w0, ad.g_bar, t0g); t0b = fmaf(w0, ad.b_bar, t0b); t0a = fmaf(w0, ad.a_bar, t0a);
                    t1r = fmaf(w1, ad.r_bar, t1r); t1g = fmaf(w1, ad.g_bar, t1g); t1b = fmaf(w1, ad.b_bar, t1b); t1a = fmaf(w1, ad.a_bar, t1a);
                }
                if (WANT_VOL) {
                    const float I_bar = fix_clamp(intensity_adjoint(sm, L.tf[sm.lo], L.tf[sm.hi], ad, P.tf_len), fs);
                    const int by = t.ly * BOX_SY, bz = t.lz, bx = t.lx * BOX_SX;
                    tri_scatter_lds(L.dbox, bx + by + bz, t.fx, t.fy, t.fz, I_bar, fs);
                    if (!sm.flat) {
                        const float gx = fix_clamp(ad.gx, fs), gy = fix_clamp(ad.gy, fs), gz = fix_clamp(ad.gz, fs);
                        tri_scatter_lds(L.dbox, t.lxp * BOX_SX + by + bz, t.fxp, t.fy, t.fz, gx, fs);
                        tri_scatter_lds(L.dbox, t.lxm * BOX_SX + by + bz, t.fxm, t.fy, t.fz, -gx, fs);
                        tri_scatter_lds(L.dbox, bx + t.lyp * BOX_SY + bz, t.fx, t.fyp, t.fz, gy, fs);
                        tri_scatter_lds(L.dbox, bx + t.lym * BOX_SY + bz, t.fx, t.fym, t.fz, -gy, fs);
                        tri_scatter_lds(L.dbox, bx + by + t.lzp, t.fx, t.fy, t.fzp, gz, fs);
                        tri_scatter_lds(L.dbox, bx + by + t.lzm, t.fx, t.fy, t.fzm, -gz, fs);
                    }
                }
            }
            if (WANT_TF && tf_lo >= 0) {
                fix_add(L.dtf + 4 * tf_lo + 0, fix_clamp(t0r, fs), fs); fix_add(L.dtf + 4 * tf_lo + 1, fix_clamp(t0g, fs), fs);
                fix_add(L.dtf + 4 * tf_lo + 2, fix_clamp(t0b, fs), fs); fix_add(L.dtf + 4 * tf_lo + 3, fix_clamp(t0a, fs), fs);
                fix_add(L.dtf + 4 * tf_hi + 0, fix_clamp(t1r, fs), fs); fix_add(L.dtf + 4 * tf_hi + 1, fix_clamp(t1g, fs), fs);
                fix_add(L.dtf + 4 * tf_hi + 2, fix_clamp(t1b, fs), fs); fix_add(L.dtf + 4 * tf_hi + 3, fix_clamp(t1a, fs), fs);
            }
        }
        __syncthreads();
    }
    if (!any) return;  // uniform
    // flush: one pass of global float atomics per brick, walking the gradient's fastest axis
    if (WANT_VOL) {
        GradView dv = P.dvol;
        dv.p += view * P.dvol_vs;
        const int fast = (dv.sx <= dv.sy && dv.sx <= dv.sz) ? 0 : ((dv.sy <= dv.sz) ? 1 : 2);
        for (int idx = threadIdx.x; idx < BOX_VOX; idx += 256) {
            const int a = idx % BOX, b = (idx / BOX) % BOX, d = idx / (BOX * BOX);
            int lx, ly, lz;
            if (fast == 0) { lx = a; ly = b; lz = d; } else if (fast == 1) { ly = a; lx = b; lz = d; } else { lz = a; ly = b; lx = d; }
            const unsigned long long raw = L.dbox[lx * BOX_SX + ly * BOX_SY + lz];
            if (raw != 0ull) {
                const float v = fix_to_float(raw, fs);
                const int gx = c.ox + lx, gy = c.oy + ly, gz = c.oz + lz;  // in range whenever v != 0
                unsafeAtomicAdd(dv.p + gx * dv.sx + gy * dv.sy + gz * dv.sz, v);
            }
        }
    }
    if (WANT_TF) {
        float *dtf = P.d_tf + view * P.dtf_vs * 4;
        for (int k = threadIdx.x; k < 4 * P.R; k += 256) {
            const unsigned long long raw = L.dtf[k];
            if (raw != 0ull) unsafeAtomicAdd(dtf + k, fix_to_float(raw, fs));
        }
    }
}

// ------------------------------------------------------------------------------------------------ host
bool brick_path_supported(int VX, int VY, int VZ, int R) {
    const int m = VX > VY ? (VX > VZ ? VX : VZ) : (VY > VZ ? VY : VZ);
    if (m - 1 >= 2000) return false;       // normal taps must stay within one voxel of the centre cell
    if (brick_lds_bytes(R, true, true) > 160 * 1024) return false;
    return true;
}

size_t brick_workspace_bytes(int n_views, int W, int H, int VX, int VY, int VZ) {
    const BrickGrid g = make_brick_grid(VX, VY, VZ);
    return ws_layout(nullptr, n_views, W * H, g, nullptr);
}

template <typename VT>
static int brick_fwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    const size_t need = ws_layout(a.workspace, a.n_views, NP, g, &w);
    if (!a.workspace || a.workspace_bytes < need) return DR_EINVAL;
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    hipError_t e = hipMemsetAsync(w.stats, 0, 256, stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(w.seg_cnt, 0, w.cnt_bytes, stream);
    if (e != hipSuccess) return (int)e;
    const size_t lds = brick_lds_bytes(a.R, false, false);
    const dim3 grid1(g.NBx * g.NBy * g.NBz, a.n_views);
    if (a.mode == DR_MODE_DIFF) {
        if ((e = allow_lds(brick_fwd_kernel<VT, DR_MODE_DIFF>, lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((brick_fwd_kernel<VT, DR_MODE_DIFF>), grid1, dim3(256), lds, stream, P);
    } else {
        if ((e = allow_lds(brick_fwd_kernel<VT, DR_MODE_NONDIFF>, lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((brick_fwd_kernel<VT, DR_MODE_NONDIFF>), grid1, dim3(256), lds, stream, P);
    }
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    return launch_ray_compose(a, stream);
}

// F2 for any of the brick pipelines: the workspace must hold the F1 output of the same call.
template <typename VT>
static int ray_compose_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    ws_layout(a.workspace, a.n_views, NP, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    const dim3 grid2((NP + 255) / 256, a.n_views);
    const size_t lds2 = (size_t)a.R * 16;
    if (a.mode == DR_MODE_DIFF)
        hipLaunchKernelGGL((ray_compose_kernel<VT, DR_MODE_DIFF>), grid2, dim3(256), lds2, stream, P);
    else
        hipLaunchKernelGGL((ray_compose_kernel<VT, DR_MODE_NONDIFF>), grid2, dim3(256), lds2, stream, P);
    return (int)hipGetLastError();
}

int launch_ray_compose(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_compose_dispatch<__half>(a, stream) : ray_compose_dispatch<float>(a, stream);
}

template <typename VT>
static int ray_alpha_dispatch(const MarchArgs &a, hipStream_t stream, bool cross) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    ws_layout(a.workspace, a.n_views, NP, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    const dim3 grid2((NP + 255) / 256, a.n_views), grid3((NP + 3) / 4 < 8192 ? (NP + 3) / 4 : 8192, a.n_views);
    if (a.mode == DR_MODE_DIFF) {
        if (!cross) hipLaunchKernelGGL((ray_alpha_kernel<VT, DR_MODE_DIFF>), grid2, dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((ray_cross_kernel<VT, DR_MODE_DIFF>), grid3, dim3(256), 0, stream, P);
    } else {
        if (!cross) hipLaunchKernelGGL((ray_alpha_kernel<VT, DR_MODE_NONDIFF>), grid2, dim3(256), 0, stream, P);
        else hipLaunchKernelGGL((ray_cross_kernel<VT, DR_MODE_NONDIFF>), grid3, dim3(256), 0, stream, P);
    }
    return (int)hipGetLastError();
}

int launch_ray_alpha(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_alpha_dispatch<__half>(a, stream, false) : ray_alpha_dispatch<float>(a, stream, false);
}
int launch_ray_cross(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_alpha_dispatch<__half>(a, stream, true) : ray_alpha_dispatch<float>(a, stream, true);
}

int launch_march_fwd_brick(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? brick_fwd_dispatch<__half>(a, stream) : brick_fwd_dispatch<float>(a, stream);
}

template <typename VT>
static int brick_bwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    const size_t need = ws_layout(a.workspace, a.n_views, NP, g, &w);
    if (!a.workspace || a.workspace_bytes < need) return DR_EINVAL;
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    const bool wv = a.d_vol != nullptr, wt = a.d_tf != nullptr;
    const size_t lds = brick_lds_bytes(a.R, wv, wt);
    const dim3 grid1(g.NBx * g.NBy * g.NBz, a.n_views);
    hipError_t e = hipMemsetAsync(w.stats + 1, 0, 4, stream);
    if (e != hipSuccess) return (int)e;
    const size_t ng = (size_t)a.n_views * NP * 4;
    hipLaunchKernelGGL(absmax_kernel, dim3((unsigned)((ng + 256 * 16 - 1) / (256 * 16)) > 1024u ? 1024u : (unsigned)((ng + 256 * 16 - 1) / (256 * 16))),
                       dim3(256), 0, stream, a.grad_out, ng, w.stats + 1);
    if (wv && wt) {
        if ((e = allow_lds(brick_bwd_kernel<VT, true, true>, lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((brick_bwd_kernel<VT, true, true>), grid1, dim3(256), lds, stream, P);
    } else if (wv) {
        if ((e = allow_lds(brick_bwd_kernel<VT, true, false>, lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((brick_bwd_kernel<VT, true, false>), grid1, dim3(256), lds, stream, P);
    } else {
        if ((e = allow_lds(brick_bwd_kernel<VT, false, true>, lds)) != hipSuccess) return (int)e;
        hipLaunchKernelGGL((brick_bwd_kernel<VT, false, true>), grid1, dim3(256), lds, stream, P);
    }
    e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    // B2: irregular rays through the baseline backward, restricted to the rays F2 flagged
    MarchArgs b = a;
    b.only_flagged = w.rayflag;
    return launch_march_bwd_baseline(b, stream);
}

int launch_march_bwd_brick(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? brick_bwd_dispatch<__half>(a, stream) : brick_bwd_dispatch<float>(a, stream);
}

}  // namespace dr
