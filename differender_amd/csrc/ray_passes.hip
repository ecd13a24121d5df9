// ray_passes.hip -- the per-RAY passes of the brick-centric pipeline (gfx950); the per-BRICK passes (F1, the alpha
// pre-pass P1, B1) live in march_flat.hip.
//
//   F2  ray_compose_kernel  per ray: composite the per-(ray, layer) partials of F1 front to back ("over" is
//                           associative); without an alpha pre-pass the one segment in which alpha crosses 0.99 is
//                           re-marched sample by sample so early termination stays exact (VR.py:267); irregular rays
//                           are marched whole. Leaves the prefix (C, A) before every segment in the workspace for B1.
//   P2  ray_alpha_kernel    per ray: composes the alpha-only partials of one phase of the pre-pass, finds the segment
//                           in which the ray terminates.
//   P2b ray_cross_kernel    one wave per terminating ray: the exact sample at which alpha reaches 0.99.
//
// Reference functions replaced: get_final_image[_nondiff] (VR.py:353-372) and the early-termination test of
// raycast / raycast_nondiff (VR.py:267,318).
#include "dr_brick_common.h"
#include "dr_wave.h"
#include "dr_tuning.h"

namespace dr {

// ------------------------------------------------------------------------------------------------ F2
// Brick that holds sample s of a ray (canonical cell computation). A ray's layers (dr_brick.h) run from 0 -- the brick
// of its first sample -- to the layer of its last marched sample: the per-ray passes only look at that range of the
// [layer][pixel] workspace (typically 45-60 of 127 layers at 512^3).
template <typename VT>
__device__ __forceinline__ void sample_brick(const BrickParams<VT> &P, const RayGeom &rg, f3 cam, int s, int &bx, int &by, int &bz) {
    float px, py, pz, fr;
    int x0, y0, z0;
    sample_pos(rg, cam.x, cam.y, cam.z, s, px, py, pz);
    axis_coord(px, P.vol.scx, x0, fr); axis_coord(py, P.vol.scy, y0, fr); axis_coord(pz, P.vol.scz, z0, fr);
    bx = x0 / BRK; by = y0 / BRK; bz = z0 / BRK;
}
// layer of the ray's last marched sample (= number of layers the per-ray passes walk, minus one)
template <typename VT>
__device__ __forceinline__ int last_layer(const BrickParams<VT> &P, const RayGeom &rg, f3 cam, int nmarch, int &ebx, int &eby, int &ebz) {
    int bx, by, bz;
    sample_brick(P, rg, cam, 0, ebx, eby, ebz);
    sample_brick(P, rg, cam, nmarch - 1, bx, by, bz);
    return min(ray_layer(bx, by, bz, ebx, eby, ebz), P.g.NL - 1);
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
    bool want_exact = false;   // the pixel is recomputed sample by sample (ray_exact_kernel)
    bool want_exact_bwd = false;   // ... and so is its backward (ray_exact_bwd_kernel)
    if (pl == 0 && P.hint_noterm && P.vflags[view] != 0u) atomicAdd(&P.stats[ST_HINT_BAD], 1u);  // (see below)
    if (rg.n > 0) {
        VolView<VT> vol = P.vol;
        vol.p += view * P.vol_vs;
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
        const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
        int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
        const int nfull = nmarch;
        const size_t seg0 = (size_t)view * P.g.NL * NP + pl;
        bool regular = ray_is_regular(rg.n);
        // DR_HINT_NO_EARLY_TERMINATION was given (no pre-pass was launched) but this view's TF CAN make rays opaque: the
        // bricks marched every planned sample, early termination was not applied -- march every ray of the view whole
        // (uniform over the view; correct, slow, counted)
        if (P.hint_noterm && P.vflags[view] != 0u) regular = false;
        // DR_TAPE_TF: the per-sample tape has room for the longest ray the volume allows at THIS call's sampling rate; a longer one (ray
        // buffers made for another rate) is marched whole here and by the per-ray backward
        if (P.tape_stride > 0 && nmarch > P.tape_stride) regular = false;
        // with an alpha pre-pass the bricks marched exactly the live samples of the ray
        const bool use_live = P.use_live && P.vflags[view] != 0u;
        if (regular && use_live) nmarch = min(nmarch, P.ws_steps[p]);
        int l_hi = -1;
        if (regular && nmarch > 0) {
            int ebx, eby, ebz;
            l_hi = last_layer(P, rg, cam, nmarch, ebx, eby, ebz);
        }
        if (regular) {
            // Composite the partials front to back, leaving the prefix before each segment for the backward. Without a
            // pre-pass no ray can reach alpha 0.99 (may_terminate() bounds it below 0.98), so there is no crossing to look
            // for. Safety net: the segments must account for every sample, else this ray is marched whole below.
            // (four layers per step, their loads issued together: the walk is a chain of memory latencies otherwise)
            int total = 0;
            D4Bound b0 = d4_zero(), b1 = d4_zero(), b2 = d4_zero(), b3 = d4_zero();   // D4 bounds of the four channels (dr_brick_common.h)

            constexpr int FW = DR_F2_WIDE;
            for (int l = 0; l <= l_hi; l += FW) {
                int cnt[FW];
                float4 sg[FW];
#pragma unroll
                for (int k = 0; k < FW; ++k) cnt[k] = (l + k <= l_hi) ? (int)P.seg_cnt[seg0 + (size_t)(l + k) * NP] : 0;   // (count | SEG_CNT_TINY)
                // (the partials are requested together with the counts, not after them: one memory round trip per step instead
                // of two -- 80.5 -> 73.9 us at 512^2; nearly every layer between a ray's first and last brick holds samples, so
                // little is read in vain. The kernel moves 330 MB in 74 us: it is HBM-bound, wider steps change nothing.)
#pragma unroll
                for (int k = 0; k < FW; ++k) sg[k] = (l + k <= l_hi) ? P.seg_rgba[seg0 + (size_t)(l + k) * NP] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int k = 0; k < FW; ++k) {
                    if (cnt[k] == 0) continue;
                    // (samples of tiny opacity in this segment: rare -- a divergent load)
                    const float tiny = (cnt[k] & SEG_CNT_TINY) ? (float)P.seg_tiny[seg0 + (size_t)(l + k) * NP] : 0.0f;
                    cnt[k] &= SEG_CNT_MAX;
                    if (MODE == DR_MODE_DIFF) P.seg_rgba[seg0 + (size_t)(l + k) * NP] = make_float4(C0, C1, C2, A);  // prefix for the backward
                    const float T = 1.0f - A;
                    // (D4: how far can the reference's sample-by-sample rounding have taken this segment from its partial? -- dr_brick_common.h;
                    //  what the alpha bound so far does to this segment's colours first: d C_after / d A = -partial)
                    const float cf = (float)cnt[k], rcf = __builtin_amdgcn_rcpf(cf);
                    {
                        const float ea = b3.total();
                        b0.lin = fmaf(ea, fabsf(sg[k].x), b0.lin); b1.lin = fmaf(ea, fabsf(sg[k].y), b1.lin); b2.lin = fmaf(ea, fabsf(sg[k].z), b2.lin);
                    }
                    const float n0 = fmaf(T, sg[k].x, C0), n1 = fmaf(T, sg[k].y, C1), n2 = fmaf(T, sg[k].z, C2), n3 = fmaf(T, sg[k].w, A);
                    d4_risk(n0, T * sg[k].x, cf, rcf, tiny, b0); d4_risk(n1, T * sg[k].y, cf, rcf, tiny, b1);
                    d4_risk(n2, T * sg[k].z, cf, rcf, tiny, b2); d4_risk(n3, T * sg[k].w, cf, rcf, tiny, b3);
                    C0 = n0; C1 = n1; C2 = n2; A = n3;
                    total += cnt[k];
                }
            }
            steps = nmarch;
            if (total != nmarch) { regular = false; nmarch = nfull; atomicAdd(&P.stats[ST_REPAIR], 1u); }
            else {
                if (MODE == DR_MODE_DIFF) P.fin[p] = make_float4(C0, C1, C2, A);   // what the stored prefixes add up to: the backward's "final"
                const float e0 = b0.total(), e1 = b1.total(), e2 = b2.total(), e3 = b3.total();
                // (3): the random walk of nmarch sequential roundings, each in ulps of a running value that never exceeds the final one
                // (the non-differentiable image is clamped to 1, VR.py:358: what a channel does above 1 is not seen)
                const float cmax = fminf(fmaxf(fmaxf(C0, C1), fmaxf(C2, A)), MODE == DR_MODE_NONDIFF ? 0.99999994f : 3.0e38f);
                const bool long_walk = 0.87f * ulp_of(cmax) * __builtin_amdgcn_sqrtf((float)nmarch) > DR_D4_WALK;
                want_exact = fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)) > DR_D4_BUDGET || long_walk;
                want_exact_bwd = want_exact && fmaxf(fmaxf(e0, e1), fmaxf(e2, e3)) > DR_D4_BWD_BUDGET;   // (only a listed ray: B3 walks the list)
            }
        }
        if (!regular) {
            // single-sample rays (and repaired ones): the sequential march of the baseline kernels
            flag = 1;
            C0 = C1 = C2 = A = 0.f;
            steps = 0;
            atomicAdd(&P.stats[ST_BASELINE_RAYS], 1u);
            for (int s = 0; s < nmarch; ++s) {
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
        }
    }
    if (MODE == DR_MODE_NONDIFF) {
        C0 = fminf(1.0f, C0); C1 = fminf(1.0f, C1); C2 = fminf(1.0f, C2); A = fminf(1.0f, A);
    }
    reinterpret_cast<float4 *>(P.out)[p] = make_float4(C0, C1, C2, A);
    if (P.steps) P.steps[p] = steps;
    P.ws_steps[p] = steps;
    // 1: a ray F2 marched whole (B2 takes its backward); 2: a ray F3 recomputes whose partials are too far from the sequential composites
    // for a backward from them (B3 takes it); a listed ray below that keeps B1
    P.rayflag[p] = want_exact_bwd ? (uint8_t)2 : flag;
    // rays for ray_exact_kernel: one list for all views, one atomic per wave
    const unsigned long long xm = __ballot(want_exact);
    if (xm != 0ull) {   // wave-uniform
        const int lane = threadIdx.x & 63;
        unsigned int base = 0u;
        if (lane == __ffsll((long long)xm) - 1) base = atomicAdd(&P.stats[ST_EXACT_RAYS], (unsigned int)__popcll(xm));
        base = (unsigned int)__shfl((int)base, __ffsll((long long)xm) - 1);
        if (want_exact) P.exact_list[base + (unsigned int)__popcll(xm & ((1ull << lane) - 1ull))] = (unsigned int)p;
    }
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
    if (P.vflags[view] == 0u) return;  // uniform: no ray of this view can terminate, nothing to find
    const int NP = P.W * P.H;
    const int pl = blockIdx.x * 256 + threadIdx.x;
    if (pl >= NP) return;
    const size_t p = (size_t)view * NP + pl;
    if (!P.pp_first && P.ws_steps[p] == -1) return;  // crossing found in an earlier phase
    RayGeom rg;
    load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
    const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
    const bool inside = P.vflags[P.n_views + view] != 0u;  // camera inside the volume: the pre-pass is one phase
    if (inside && !P.pp_first) return;                      // uniform
    if (ray_is_regular(rg.n)) {
        // state carried from phase to phase in the (not yet written) output buffer: alpha so far, samples so far
        float4 *park = reinterpret_cast<float4 *>(P.out) + p;
        float A = 0.f;
        D4Bound bA = d4_zero();   // how far the reference's sequential alpha may be from this one (D4, dr_brick_common.h)
        int sacc = 0;
        int slit = 0;   // ... of them in segments that moved alpha at all (a segment of air has the partial 0 exactly and rounds nothing)
        // (between the phases of the pre-pass the bound travels as ONE number, its total so far: the quadrature part is folded in and
        //  the drift history starts afresh -- a phase's first two segments are then not charged for standing still)
        if (!P.pp_first) { const float4 st = *park; A = st.x; sacc = __float_as_int(st.y); bA.lin = st.z; slit = __float_as_int(st.w); }
        const size_t seg0 = (size_t)view * P.g.NL * NP + pl;
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        // this phase's bricks have camera-based layers [pp_l0, pp_l1); along a ray from a camera outside the volume
        // those are the ray's own layers plus the camera-based layer of its first brick
        int ebx, eby, ebz;
        int l_hi = last_layer(P, rg, cam, nmarch, ebx, eby, ebz);
        int l_lo = 0;
        if (!inside) {
            const int off = camera_layer(ebx, eby, ebz, cam, P.vol.scx, P.vol.scy, P.vol.scz, P.g.NBx, P.g.NBy, P.g.NBz);
            l_lo = max(P.pp_l0 - off, 0);
            l_hi = min(l_hi, P.pp_l1 - 1 - off);
        }
        // (four layers per step, counts and alpha partials requested together, as in ray_compose_kernel: the walk is a chain of memory
        //  round trips -- two per layer when the partial was only asked for after its count had arrived; 14.2 -> 11.6 us at 256^2)
        constexpr int AW = DR_F2_WIDE;
        for (int l = l_lo; l <= l_hi; l += AW) {
            int c16[AW];
            float sa4[AW];
#pragma unroll
            for (int k = 0; k < AW; ++k) c16[k] = (l + k <= l_hi) ? (int)P.seg_cnt[seg0 + (size_t)(l + k) * NP] : 0;
#pragma unroll
            for (int k = 0; k < AW; ++k) sa4[k] = (l + k <= l_hi) ? P.seg_rgba[seg0 + (size_t)(l + k) * NP].w : 0.0f;
#pragma unroll
            for (int k = 0; k < AW; ++k) {
                const int cnt = c16[k] & SEG_CNT_MAX;
                if (cnt == 0) continue;
                const float tiny = (c16[k] & SEG_CNT_TINY) ? (float)P.seg_tiny[seg0 + (size_t)(l + k) * NP] : 0.0f;
                const float sa = sa4[k];
                const float A_after = fmaf(1.0f - A, sa, A);
                D4Bound b2 = bA;
                d4_risk(A_after, (1.0f - A) * sa, (float)cnt, __builtin_amdgcn_rcpf((float)cnt), tiny, b2);
                // the crossing segment (with a margin for the re-associated partials -- 1e-5, or the random walk of the sequential roundings
                // where a ray has tens of thousands of samples behind it: cross_band -- and for what sequential rounding may have done
                // systematically so far) starts at sample sacc: resolved by ray_cross_kernel
                if (!(A_after < 0.99f - fmaxf(1e-5f, 5.2e-8f * __builtin_amdgcn_sqrtf((float)(slit + cnt))) - b2.total())) {
                    *park = make_float4(A, __int_as_float(sacc), bA.total(), __int_as_float(slit));
                    P.ws_steps[p] = -1;
                    return;
                }
                A = A_after;
                bA = b2;
                sacc += cnt;
                slit += (sa != 0.0f) ? cnt : 0;
            }
        }
        *park = make_float4(A, __int_as_float(sacc), bA.total(), __int_as_float(slit));
    }
    if (P.pp_first) P.ws_steps[p] = nmarch;  // alive (so far): every planned sample is live
}

// P2b: the rays whose accumulated alpha crosses 0.99 (ws_steps == -1; the first sample of the crossing segment and the
// alpha before it are parked in the not-yet-written output buffer). One WAVE per ray, 64 consecutive samples per pass:
// positions, centre taps (the lanes read neighbouring voxels) and TF lookups (LDS) in parallel, then
//   round 0  an inclusive wave scan of the transmittances (DPP, 6 steps) gives the alpha after every sample of the pass,
//            re-associated: ~1e-7 off the sequential value, like the parked start. The first lane at or above 0.99 is the
//            terminating sample -- unless the alpha there, or the one before it, lies within 2e-6 of the threshold:
//   round 1  (0.4-3 % of the terminating rays) the exact sequential recurrence A <- fma(1 - A, op_s, A) of VR.py:267-302
//            from the ray's FIRST sample, 64 opacities per pass walked with v_readlane: the decision is then the oracle's,
//            bit for bit (deviation D3 of DESIGN.md does not arise on this path).
// (Round 1 of the build walked every crossing segment with the v_readlane loop, ~35 cycles per sample with one lane's
// worth of work: 2.7 ms of the reference-style sr-8 ground-truth render; a one-lane-per-ray walk is worse still, its
// dependent gathers take ~1 us per step.)
__device__ __forceinline__ float wave_incl_prod(float v) {  // inclusive product over the lanes 0..lane
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x111, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x112, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x114, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x118, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x142, 0xa, 0xf, false)); v *= o; }  // row_bcast:15 -> rows 1, 3
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x143, 0xc, 0xf, false)); v *= o; }  // row_bcast:31 -> rows 2, 3
    return v;
}
// opacity of sample sl of a ray (0 beyond the ray's end; a skipped nondiff sample leaves A unchanged: fma(T, 0, A) == A).
// rg may differ from lane to lane (ray_cross_quad_kernel: one ray per 16-lane row).
template <typename VT, int MODE>
__device__ __forceinline__ float cross_opacity(const BrickParams<VT> &P, const VolView<VT> &vol, const float4 *lds_tf, const RayGeom &rg,
                                               f3 cam, int nmarch, int sl) {
    Sample sm;
    sm.a = 0.0f;
    if (sl < nmarch) {
        // (the quotient s / (n - 1) from the ray's reciprocal: the correctly rounded quotient in three instructions instead of the
        //  division's twelve -- sample_pos_rcp, as in the brick kernels; rg.inv_nm1 is set where the ray is loaded)
        sample_pos_rcp(rg.t0, rg.exit_, (float)(rg.n - 1), rg.inv_nm1, rg.vx, rg.vy, rg.vz, cam.x, cam.y, cam.z, sl, sm.px, sm.py, sm.pz);
        sm.I = tri_sample(vol, sm.px, sm.py, sm.pz);
        tf_lookup_from_I(lds_tf, P.R, P.tf_len, sm);
    }
    // (the power behind the opacity only where a lane needs it: transparent stretches are wave-uniform)
    float op = 0.0f;
    if constexpr (MODE == DR_MODE_NONDIFF) {
        const bool vis = sl < nmarch && sm.a > 1e-3f;
        if (__any(vis)) {
            if (vis) op = opacity_of_alpha(sm.a, P.inv_sr);
        }
    } else {
        if (sl < nmarch) op = opacity_of_alpha(sm.a, P.inv_sr);
    }
    return op;
}

// Round 1 of the crossing search: the exact sequential recurrence from the ray's first sample, by ONE WAVE (rg, nmarch
// wave-uniform), 256 samples' gathers in flight per step. Returns the number of live samples.
// In sequential f32 arithmetic alpha can STAGNATE just below the threshold: once (1 - A) * op_s is under half
// an ulp of A, fma(1 - A, op_s, A) returns A again, for every sample whose opacity is no larger (the product
// is monotone in op_s). These are the rays that get here -- the re-associated alpha crossed, the sequential
// one does not -- and they then march to their last sample: if even the largest opacity of a pass leaves A
// unchanged, so do all 64.
template <typename VT, int MODE>
__device__ __forceinline__ int cross_exact_walk(const BrickParams<VT> &P, const VolView<VT> &vol, const float4 *lds_tf, const RayGeom &rg,
                                                f3 cam, int nmarch, int lane) {
    float A = 0.0f;
    int s = 0;
    bool done = false;
    constexpr int CB = DR_CROSS_BATCH;
    for (int base = 0; base < nmarch && !done; base += 64 * CB) {  // uniform
        float op4[CB];
#pragma unroll
        for (int j = 0; j < CB; ++j) op4[j] = cross_opacity<VT, MODE>(P, vol, lds_tf, rg, cam, nmarch, base + 64 * j + lane);
#pragma unroll
        for (int j = 0; j < CB; ++j) {
            const int cnt = min(64, nmarch - (base + 64 * j));
            if (cnt <= 0 || done) continue;  // uniform
            float opmax = op4[j];
            for (int o = 32; o > 0; o >>= 1) opmax = fmaxf(opmax, __shfl_xor(opmax, o));
            if (A < 0.99f && fmaf(1.0f - A, opmax, A) == A) {  // uniform
                s += cnt; continue;
            }
            if (cnt == 64) {
                // A whole pass without looking at the threshold: alpha never decreases, so if it is still below 0.99
                // after the 64th sample it was below it before every one of them -- the same 64 roundings as the
                // checked walk below, as a straight chain of sub + fma (constant lane indices: no scalar hazards, no
                // branch per sample: ~4 x faster). Only the pass in which the ray crosses is walked with the test.
                float Ab = A;
#pragma unroll
                for (int i = 0; i < 64; ++i)
                    Ab = fmaf(1.0f - Ab, __int_as_float(__builtin_amdgcn_readlane(__float_as_int(op4[j]), i)), Ab);
                if (Ab < 0.99f) { A = Ab; s += 64; continue; }  // uniform
            }
            for (int i = 0; i < cnt; ++i) {  // uniform
                if (!(A < 0.99f)) { done = true; break; }
                const float opi = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(op4[j]), i));
                A = fmaf(1.0f - A, opi, A);
                ++s;
            }
        }
    }
    return s;
}

// How far the reference's SEQUENTIAL alpha may be from the re-associated one after s samples, from ordinary rounding alone: a random
// walk of 0.29 ulp per sample (an ulp of an alpha below 1: 6e-8), three sigma -- 5.2e-8 sqrt(s). The 2e-6 the band has had since
// round 2 is that for 1 500 samples; rays of 10 000 samples and more (sampling rate 16 through a 288-voxel volume: fuzz seed 90320 at
// FUZZ_SCALE=3, round 6 -- the sequential alpha crossed one sample early, 3.5e-6 from the re-associated one) need the wider one.
// s counts the samples that can have rounded anything: those of the segments before the crossing one whose alpha partial is not 0
// (ray_alpha_kernel parks their number) and the crossing segment's own -- the air a ray crosses first does not widen its band.
__device__ __forceinline__ float cross_band(int s) { return fmaxf(2e-6f, 5.2e-8f * __builtin_amdgcn_sqrtf((float)s)); }

template <typename VT, int MODE>
__global__ __launch_bounds__(256) void ray_cross_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_tf[];
    const int view = blockIdx.y;
    if (P.vflags[view] == 0u) return;  // uniform: no ray of this view can terminate
    const float4 *tfg = P.tf + view * P.tf_vs;
    for (int k = threadIdx.x; k < P.R; k += 256) lds_tf[k] = tfg[k];
    __syncthreads();
    const int NP = P.W * P.H;
    const int lane = threadIdx.x & 63;
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    // one ray per wave; everything the wave may need is requested at once (the flag decides afterwards)
    for (int pl = blockIdx.x * 4 + (threadIdx.x >> 6); pl < NP; pl += 4 * (int)gridDim.x) {  // wave-uniform
        const size_t p = (size_t)view * NP + pl;
        const int flag = P.ws_steps[p];
        const float4 parked = reinterpret_cast<const float4 *>(P.out)[p];
        RayGeom rg;
        load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
        if (flag != -1) continue;  // wave-uniform: no crossing to resolve
        rg.inv_nm1 = 1.0f / (float)(rg.n - 1);   // (n >= 2: a crossing ray is regular)
        const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
        float A = parked.x, A_prev = A;
        // D4: bound on |sequential alpha - this one| up to the crossing segment (ray_alpha_kernel); from there on the samples of tiny
        // opacity are counted pass by pass (each may be off by half an ulp of an alpha below 1: 3e-8)
        int n_tiny = 0;
        int s = __float_as_int(parked.y);
        // ---- round 0: re-associated alphas of 64 samples per pass
        for (int base = s; base < nmarch && A < 0.99f; base += 64) {  // uniform
            const float op = cross_opacity<VT, MODE>(P, vol, lds_tf, rg, cam, nmarch, base + lane);
            // alpha after every sample of the pass: A_i = A + (1 - A) (1 - prod_{j <= i} (1 - op_j))
            const float Ai = fmaf(1.0f - A, 1.0f - wave_incl_prod(1.0f - op), A);
            n_tiny += __popcll(__ballot(op != 0.0f && op < DR_D4_TINY_OP));   // uniform (scalar unit)
            const unsigned long long over = __ballot(!(Ai < 0.99f));
            if (over == 0ull) { A_prev = A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ai), 63)); s += min(64, nmarch - base); continue; }
            const int ix = __ffsll((long long)over) - 1;  // first sample at or above the threshold: the last live one
            A_prev = ix > 0 ? __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ai), ix - 1)) : A;
            A = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Ai), ix));
            s += ix + 1;
            break;
        }
        // decided at A (>= 0.99, or the ray ran out of samples) with A_prev (< 0.99) before it
        const float band = cross_band(__float_as_int(parked.w) + (s - __float_as_int(parked.y))) + parked.z + (float)n_tiny * 3.0e-8f;
        bool ambiguous = fabsf(A - 0.99f) < band || fabsf(A_prev - 0.99f) < band;
        if (ambiguous) s = cross_exact_walk<VT, MODE>(P, vol, lds_tf, rg, cam, nmarch, lane);  // ---- round 1
        if (lane == 0) P.ws_steps[p] = s;
    }
}

// The same search with FOUR rays per wave, one per 16-lane row: at sampling rates below 2 a crossing segment holds at most
// ~40 samples (12 cells x sqrt 3 x the rate), and a 64-sample pass per ray leaves two thirds of the lanes without work.
// Sixteen samples per ray and pass, products within the row (DPP row shifts), the pass's first crossing from the row's 16
// bits of the ballot. The association of the products differs from the one-ray kernel's (and from the pre-pass's) by
// rounding, which is what the ambiguity band is for: a ray whose alpha lands within 2e-6 of the threshold is walked
// exactly (round 1, by the whole wave, one ambiguous ray after the other), so the decision is the oracle's either way.
__device__ __forceinline__ float row_incl_prod(float v) {  // inclusive product over the lanes of a 16-lane row
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x111, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x112, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x114, 0xf, 0xf, false)); v *= o; }
    { const float o = __int_as_float(__builtin_amdgcn_update_dpp(0x3f800000, __float_as_int(v), 0x118, 0xf, 0xf, false)); v *= o; }
    return v;
}
template <typename VT, int MODE>
__global__ __launch_bounds__(256) void ray_cross_quad_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) float4 lds_tf[];
    const int view = blockIdx.y;
    if (P.vflags[view] == 0u) return;  // uniform: no ray of this view can terminate
    const float4 *tfg = P.tf + view * P.tf_vs;
    for (int k = threadIdx.x; k < P.R; k += 256) lds_tf[k] = tfg[k];
    __syncthreads();
    const int NP = P.W * P.H;
    const int lane = threadIdx.x & 63, row = lane >> 4, rl = lane & 15;
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    for (int pl4 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; pl4 < NP; pl4 += 16 * (int)gridDim.x) {  // wave-uniform: 4 neighbouring pixels
        const int pl = pl4 + row;
        const bool in = pl < NP;
        const size_t p = (size_t)view * NP + (in ? pl : NP - 1);
        const int flag = P.ws_steps[p];
        const float4 parked = reinterpret_cast<const float4 *>(P.out)[p];
        RayGeom rg;
        load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
        rg.inv_nm1 = 1.0f / (float)max(rg.n - 1, 1);
        const bool act = in && flag == -1;  // row-uniform
        const unsigned long long actm = __ballot(act);
        if (actm == 0ull) continue;  // wave-uniform: none of the four rays crosses
        const int nmarch = (MODE == DR_MODE_DIFF && rg.n > P.S) ? P.S : rg.n;
        float A = parked.x, A_prev = A;
        int n_tiny = 0;   // (D4: samples of tiny opacity seen by this row's search, as in ray_cross_kernel)
        int s = __float_as_int(parked.y);
        bool open = act;  // the row is still looking for its crossing
        // ---- round 0: re-associated alphas of 16 samples per ray and pass
        while (true) {
            const bool go = open && s < nmarch && A < 0.99f;  // row-uniform
            if (!__any(go)) break;                            // wave-uniform
            const float op = cross_opacity<VT, MODE>(P, vol, lds_tf, rg, cam, go ? nmarch : 0, s + rl);
            const float Ai = fmaf(1.0f - A, 1.0f - row_incl_prod(1.0f - op), A);
            const unsigned long long over = __ballot(go && !(Ai < 0.99f));
            const unsigned long long tinym = __ballot(go && op != 0.0f && op < DR_D4_TINY_OP);
            const unsigned int m = (unsigned int)(over >> (16 * row)) & 0xffffu;
            const int ix = m ? __ffs((int)m) - 1 : 15;   // first sample of the row at or above the threshold (15: none -- the row's last)
            const float a_ix = __shfl(Ai, 16 * row + ix);
            const float a_before = __shfl(Ai, 16 * row + max(ix - 1, 0));
            if (go) {
                n_tiny += __popc((unsigned int)(tinym >> (16 * row)) & 0xffffu);
                if (m == 0u) { A_prev = A = a_ix; s += min(16, nmarch - s); }
                else { A_prev = ix > 0 ? a_before : A; A = a_ix; s += ix + 1; open = false; }
            }
        }
        // decided at A (>= 0.99, or the ray ran out of samples) with A_prev (< 0.99) before it
        const float band = cross_band(__float_as_int(parked.w) + (s - __float_as_int(parked.y))) + parked.z + (float)n_tiny * 3.0e-8f;
        bool ambiguous = act && (fabsf(A - 0.99f) < band || fabsf(A_prev - 0.99f) < band);
        const unsigned long long ambm = __ballot(ambiguous);
        if (ambm != 0ull) {  // wave-uniform, rare
            for (int r = 0; r < 4; ++r) {
                if (!((ambm >> (16 * r)) & 1ull)) continue;  // uniform
                // ---- round 1 for the ray of row r: its geometry to every lane, then the one-ray walk by the whole wave
                RayGeom ru;
                ru.entry = __shfl(rg.entry, 16 * r); ru.exit_ = __shfl(rg.exit_, 16 * r);
                ru.vx = __shfl(rg.vx, 16 * r); ru.vy = __shfl(rg.vy, 16 * r); ru.vz = __shfl(rg.vz, 16 * r);
                ru.n = __shfl(rg.n, 16 * r); ru.t0 = __shfl(rg.t0, 16 * r); ru.inv_nm1 = __shfl(rg.inv_nm1, 16 * r);
                const int nm_u = __shfl(nmarch, 16 * r);
                const int s_u = cross_exact_walk<VT, MODE>(P, vol, lds_tf, ru, cam, nm_u, lane);
                if (row == r) s = s_u;
            }
        }
        if (act && rl == 0) P.ws_steps[p] = s;
    }
}

// ------------------------------------------------------------------------------------------------ F3
// The rays F2 listed (runs of contributions that sequential float32 rounding treats alike, where the reference's
// sample-by-sample result and the partials part company): their pixels once more, in the reference's order --
// tape[s] = (1 - tape[s-1].w) * shaded + tape[s-1], one rounding per sample and channel (VR.py:300-302) -- with the sequential
// kernels' arithmetic bit for bit (classify / shade of dr_device.h: exact normalisations, taps from global memory). One
// four-wave WORKGROUP per ray: 256 samples at a time are evaluated side by side and parked in LDS (two buffers), and the first
// wave composites them one after the other while the others already evaluate the next 256 -- the chain (about 20 cycles per
// sample: it cannot be parallelised, the roundings are the point) of one ray hides behind the evaluations of the other
// workgroups of its CU. The ray's live sample count is F2's (exact: D3),
// so no sample is evaluated in vain and no threshold is looked at. The transfer function is read where it lies (the list mixes
// views). Only the image changes: sample counts, ray flags, the stored prefixes and `fin` stay -- the backward's tape-free
// identity wants a final value consistent with its prefixes, and gets it from `fin`.
// Two shapes, both launched, one of them leaves at once: up to EXACT_FEW rays are a matter of LATENCY (a ray of 3 500 samples in
// 256-sample rounds takes 14 of them one after the other) and get 1024 threads each (one workgroup per ray and CU); more are a
// matter of throughput and get four-wave workgroups, five to a CU. (What a forward that lists nothing pays: 9 200 waves that load
// one word and leave.)
constexpr unsigned int EXACT_FEW = 256;
template <typename VT, int MODE, int EXACT_NT>
__global__ __launch_bounds__(EXACT_NT) void ray_exact_kernel(BrickParams<VT> P) {
    __shared__ float4 park[2][EXACT_NT];
    const unsigned int count = P.stats[ST_EXACT_RAYS];   // uniform
    if (count == 0u || (count <= EXACT_FEW) != (EXACT_NT == 1024)) return;
    const int NP = P.W * P.H;
    const int wave = threadIdx.x >> 6;
    for (unsigned int i = blockIdx.x; i < count; i += gridDim.x) {   // uniform
        const size_t p = P.exact_list[i];
        const int view = (int)(p / (size_t)NP);
        const float4 *tfg = P.tf + view * P.tf_vs;
        VolView<VT> vol = P.vol;
        vol.p += view * P.vol_vs;
        RayGeom rg;
        load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
        const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
        const int nmarch = P.ws_steps[p];   // live samples (F2)
        float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;   // (first wave)
        int it = 0;
        for (int base = 0; base < nmarch; base += EXACT_NT, ++it) {   // uniform
            const int s = base + (int)threadIdx.x;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);   // (a sample the nondiff march skips leaves the composite as it is: fma(T, 0, C) == C)
            if (s < nmarch) {
                Sample sm;
                sample_pos(rg, cam.x, cam.y, cam.z, s, sm.px, sm.py, sm.pz);
                classify(vol, tfg, P.R, P.tf_len, P.inv_sr, sm);
                if (MODE != DR_MODE_NONDIFF || sm.a > 1e-3f) {
                    shade(vol, light, vd, MODE == DR_MODE_DIFF, sm);
                    c = make_float4(sm.L * sm.r * sm.op, sm.L * sm.g * sm.op, sm.L * sm.b * sm.op, sm.op);
                }
            }
            float4 *buf = park[it & 1];
            buf[threadIdx.x] = c;
            __syncthreads();   // (the buffer written two rounds ago is free again: the first wave finished with it before it got here)
            if (wave == 0) {
                // the sequential recurrence, one sample after the other: every lane reads the same LDS word (a broadcast) and
                // carries the same composite -- one load, one subtraction and four fused multiply-adds per sample
                const int cnt = min(EXACT_NT, nmarch - base);
#pragma unroll 8
                for (int k = 0; k < cnt; ++k) {   // uniform
                    const float4 v = buf[k];
                    const float T = 1.0f - A;
                    C0 = fmaf(T, v.x, C0); C1 = fmaf(T, v.y, C1); C2 = fmaf(T, v.z, C2);
                    A = fmaf(T, v.w, A);
                }
            }
        }
        if (MODE == DR_MODE_NONDIFF) { C0 = fminf(1.0f, C0); C1 = fminf(1.0f, C1); C2 = fminf(1.0f, C2); A = fminf(1.0f, A); }
        if (threadIdx.x == 0) reinterpret_cast<float4 *>(P.out)[p] = make_float4(C0, C1, C2, A);
        __syncthreads();   // the next ray starts in park[0] again
    }
}

// ------------------------------------------------------------------------------------------------ B3
// The backward of the rays F3 recomputed. Their image is the reference's sequential composite, and the reference's adjoint is built
// from the sequential composites too -- the tape-free identity with the prefix BEFORE each sample and the final value as the
// sequential recurrence produced them (VR.py:460-461 on the tape of :300-302). Where the partials and the sequential values part
// company by 1e-4 of a composite (what such a ray was listed for), a backward from the partials is 1e-4 off per ray, and a TF whose
// transparent ranges carry tiny alphas lists a sixth of the rays: d_tf 1e-4 .. 5e-4 of its maximum at 256^3 and 512^3 (tools/
// diff_sweep.py, round 6). So F2 flags the listed rays whose bound exceeds DR_D4_BWD_BUDGET (rayflag 2): B1 and the tape pass leave
// them alone, the per-ray pass B2 (one LANE per ray: 8 ms for a handful of rays) does too, and this kernel serves them -- one
// four-wave workgroup per ray, four to a CU: 256 samples at a time evaluated side by side with the sequential kernels' arithmetic
// (classify / shade: exact normalisations, global taps), their contributions chained by the first wave into the composite before
// every sample (LDS), then every thread forms its sample's adjoint exactly as march_bwd_baseline_kernel does and scatters it
// (d_volume: global float atomics, one set of eight per run of samples in a cell; d_tf: run sums into a double table in LDS,
// flushed per view). A resident grid that leaves at once when nothing is flagged. It is a slow path -- the per-ray gathers and the
// chain are what they are: a TF with tiny alphas all over (28 % of the rays flagged at 512^3) pays 10-20 x the brick backward.
// the eight corner weights of tri_scatter_global (dr_device.h), times adj -- same products, same order
__device__ __forceinline__ void tri_corner_weights(const Cell &c, float adj, float (&w)[8]) {
    const float gx = 1.0f - c.fx, gy = 1.0f - c.fy, gz = 1.0f - c.fz;
    w[0] = gx * gy * gz * adj; w[1] = c.fx * gy * gz * adj; w[2] = gx * c.fy * gz * adj; w[3] = c.fx * c.fy * gz * adj;
    w[4] = gx * gy * c.fz * adj; w[5] = c.fx * gy * c.fz * adj; w[6] = gx * c.fy * c.fz * adj; w[7] = c.fx * c.fy * c.fz * adj;
}
constexpr int EXACT_BWD_NT = 256;
template <typename VT>
__global__ __launch_bounds__(EXACT_BWD_NT) void ray_exact_bwd_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_b3[];
    unsigned long long *lds_dtf = reinterpret_cast<unsigned long long *>(smem_b3);                      // [R][4] doubles
    float4 *park = reinterpret_cast<float4 *>(smem_b3 + (size_t)P.R * 32);                              // [NT] contributions
    float4 *pre = park + EXACT_BWD_NT;                                                                  // [NT] composite before the sample
    // only after the forward of THIS call (else B2 marches every ray and nothing is left to do here)
    if (P.stats[ST_MARK] != P.mark) return;   // uniform
    if (P.tape_stride > 0 && P.stats[ST_TAPE_STRIDE] != (unsigned int)P.tape_stride) return;
    const unsigned int count = P.stats[ST_EXACT_RAYS];
    if (count == 0u) return;
    const int NP = P.W * P.H;
    const bool want_vol = P.dvol.p != nullptr, want_tf = P.d_tf != nullptr;
    const float delta = 1e-3f;
    int cur_view = -1;   // whose d_tf the LDS table holds
    auto flush = [&]() {   // (all threads)
        __syncthreads();
        if (want_tf && cur_view >= 0) {
            for (int k = threadIdx.x; k < 4 * P.R; k += EXACT_BWD_NT) {
                const unsigned long long raw = lds_dtf[k];
                if (raw != 0ull) dtf64_add(P.dtf64, cur_view, k, raw);   // (the call's double table: dtf_commit_kernel)
                lds_dtf[k] = 0ull;
            }
        }
        __syncthreads();
    };
    for (unsigned int i = blockIdx.x; i < count; i += gridDim.x) {   // uniform
        const size_t p = P.exact_list[i];
        if (P.rayflag[p] != 2) continue;   // uniform: listed for its image only (a bound between the two budgets, a long ray): B1 has it
        const int view = (int)(p / (size_t)NP);
        if (view != cur_view) {
            if (cur_view >= 0) flush();
            else { if (want_tf) for (int k = threadIdx.x; k < 4 * P.R; k += EXACT_BWD_NT) lds_dtf[k] = 0ull; __syncthreads(); }
            cur_view = view;
        }
        const float4 *tfg = P.tf + view * P.tf_vs;
        VolView<VT> vol = P.vol;
        vol.p += view * P.vol_vs;
        GradView dv = P.dvol;
        if (want_vol) dv.p += view * P.dvol_vs;
        RayGeom rg;
        load_ray(P.entry, P.exit_, P.rays, P.nsamp, p, rg);
        const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
        const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
        const f3 vd = make_f3(rg.vx, rg.vy, rg.vz);
        const int live = P.ws_steps[p];   // the samples the sequential march takes (F2; exact: D3)
        const float4 go = reinterpret_cast<const float4 *>(P.grad_out)[p];
        const float4 of = reinterpret_cast<const float4 *>(P.out_fwd)[p];   // the image: F3's sequential composite
        float C0 = 0.f, C1 = 0.f, C2 = 0.f, A = 0.f;   // (first wave)
        for (int base = 0; base < live; base += EXACT_BWD_NT) {   // uniform
            const int s = base + (int)threadIdx.x;
            const bool have = s < live;
            Sample sm;
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (have) {
                sample_pos(rg, cam.x, cam.y, cam.z, s, sm.px, sm.py, sm.pz);
                classify(vol, tfg, P.R, P.tf_len, P.inv_sr, sm);
                shade(vol, light, vd, true, sm);
                c = make_float4(sm.L * sm.r * sm.op, sm.L * sm.g * sm.op, sm.L * sm.b * sm.op, sm.op);
            }
            park[threadIdx.x] = c;
            __syncthreads();
            if (threadIdx.x < 64) {
                // the sequential recurrence; every lane carries the same composite, lane j keeps the one before sample kb + j in
                // registers and the wave stores 64 of them at a time (a store inside the chain would order every broadcast load
                // behind it: 143 cycles per sample instead of 25). Samples beyond the ray's last are parked as zeros: fma(T, 0, C) = C.
                const int cnt = min(EXACT_BWD_NT, live - base);
                const int lane = (int)threadIdx.x;
                for (int kb = 0; kb < cnt; kb += 64) {   // uniform
                    float4 mine = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
                    for (int j = 0; j < 64; ++j) {
                        const float4 v = park[kb + j];
                        if (j == lane) mine = make_float4(C0, C1, C2, A);
                        const float T = 1.0f - A;
                        C0 = fmaf(T, v.x, C0); C1 = fmaf(T, v.y, C1); C2 = fmaf(T, v.z, C2);
                        A = fmaf(T, v.w, A);
                    }
                    pre[kb + lane] = mine;
                }
            }
            __syncthreads();
            // every thread forms its sample's adjoint exactly as march_bwd_baseline_kernel does
            float V8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // d_tf addends: (r, g, b, a) for the lower texel, then the upper
            int key = -1 - (int)(threadIdx.x & 63), key_hi = 0;       // TF cell (no sample: a key of its own)
            float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};  // d_volume addends for the eight corners of the sample's cell
            int cell_key = -1 - (int)(threadIdx.x & 63);              // ... which cell
            float *cb00 = nullptr, *cb10 = nullptr, *cb01 = nullptr, *cb11 = nullptr;
            int64_t co0 = 0, co1 = 0;
            if (have) {
                const float4 b = pre[threadIdx.x];
                const float T = 1.0f - b.w;
                const float a0 = fmaf(T, c.x, b.x), a1 = fmaf(T, c.y, b.y), a2 = fmaf(T, c.z, b.z), a3 = fmaf(T, c.w, b.w);   // after the sample
                const bool last = s == live - 1;
                const float suffix = (go.x * (of.x - a0) + go.y * (of.y - a1) + go.z * (of.z - a2)) + go.w * (of.w - a3);
                SampleAdj ad;
                sample_adjoint(sm, vd, T, suffix, last, go, P.inv_sr, ad);
                float I_bar = want_vol ? intensity_adjoint(sm, tfg[sm.lo], tfg[sm.hi], ad, P.tf_len) : 0.0f;
                // (the fast backward promises finite gradients: a NaN adjoint is dropped, infinities are clamped -- as in B2)
                ad.r_bar = acc_sanitise(ad.r_bar); ad.g_bar = acc_sanitise(ad.g_bar); ad.b_bar = acc_sanitise(ad.b_bar);
                ad.a_bar = acc_sanitise(ad.a_bar); ad.gx = acc_sanitise(ad.gx); ad.gy = acc_sanitise(ad.gy);
                ad.gz = acc_sanitise(ad.gz); I_bar = acc_sanitise(I_bar);
                if (want_tf) {
                    const float w0 = 1.0f - sm.fr, w1 = sm.fr;   // (the products are f32, as in the oracle)
                    V8[0] = w0 * ad.r_bar; V8[1] = w0 * ad.g_bar; V8[2] = w0 * ad.b_bar; V8[3] = w0 * ad.a_bar;
                    V8[4] = w1 * ad.r_bar; V8[5] = w1 * ad.g_bar; V8[6] = w1 * ad.b_bar; V8[7] = w1 * ad.a_bar;
                    key = sm.lo; key_hi = sm.hi;
                }
                if (want_vol) {
                    // the seven taps of a sample lie 1e-3 apart: nearly always in ONE cell -- their adjoints are added up per corner in
                    // registers (a tap in a neighbouring cell goes on its own), and so are, below, the samples of consecutive lanes
                    // that share the cell: eight atomics per run of samples in a cell instead of 56 per sample
                    Cell c0;
                    tri_cell(vol, sm.px, sm.py, sm.pz, c0);
                    tri_corner_weights(c0, I_bar, acc);
                    auto tap = [&](float qx, float qy, float qz, float adj) {
                        Cell cq;
                        tri_cell(vol, qx, qy, qz, cq);
                        if (cq.x0 == c0.x0 && cq.y0 == c0.y0 && cq.z0 == c0.z0) {
                            float wq[8];
                            tri_corner_weights(cq, adj, wq);
#pragma unroll
                            for (int q = 0; q < 8; ++q) acc[q] += wq[q];
                        } else {
                            tri_scatter_global(vol, dv, qx, qy, qz, adj);
                        }
                    };
                    if (!sm.flat) {
                        tap(sm.px + delta, sm.py, sm.pz, ad.gx); tap(sm.px - delta, sm.py, sm.pz, -ad.gx);
                        tap(sm.px, sm.py + delta, sm.pz, ad.gy); tap(sm.px, sm.py - delta, sm.pz, -ad.gy);
                        tap(sm.px, sm.py, sm.pz + delta, ad.gz); tap(sm.px, sm.py, sm.pz - delta, -ad.gz);
                    }
                    // (11 bits per coordinate, the top one dropped: consecutive samples never lie 512 cells apart; fast-path volumes are below 2048)
                    cell_key = (c0.x0 | (c0.y0 << 11) | (c0.z0 << 22)) & 0x7fffffff;
                    cb00 = dv.p + c0.x0 * dv.sx + c0.y0 * dv.sy; cb10 = dv.p + c0.x1 * dv.sx + c0.y0 * dv.sy;
                    cb01 = dv.p + c0.x0 * dv.sx + c0.y1 * dv.sy; cb11 = dv.p + c0.x1 * dv.sx + c0.y1 * dv.sy;
                    co0 = c0.z0 * dv.sz; co1 = c0.z1 * dv.sz;
                }
            }
            if (want_vol) {   // uniform
                const int lane = (int)(threadIdx.x & 63);
                const int prev = wave_up1(cell_key, cell_key);
                const bool start = lane == 0 || cell_key != prev;
                const unsigned long long starts = __ballot(start);
                const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                const int rs = 63 - __clzll((long long)(starts & upto));   // first lane of this lane's run of samples in one cell
                seg_scan_sum<8>(acc, lane, rs);
                const bool run_end = lane == 63 || ((starts >> ((lane + 1) & 63)) & 1ull);
                if (cell_key >= 0 && run_end) {
                    unsafeAtomicAdd(cb00 + co0, acc[0]); unsafeAtomicAdd(cb10 + co0, acc[1]); unsafeAtomicAdd(cb01 + co0, acc[2]); unsafeAtomicAdd(cb11 + co0, acc[3]);
                    unsafeAtomicAdd(cb00 + co1, acc[4]); unsafeAtomicAdd(cb10 + co1, acc[5]); unsafeAtomicAdd(cb01 + co1, acc[6]); unsafeAtomicAdd(cb11 + co1, acc[7]);
                }
            }
            if (want_tf) {   // uniform
                // lanes are consecutive samples: runs of them share a TF cell -- summed across the lanes (DPP), one set of LDS adds per run
                const int lane = (int)(threadIdx.x & 63);
                const int prev = wave_up1(key, key);
                const bool start = lane == 0 || key != prev;
                const unsigned long long starts = __ballot(start);
                const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                const int rs = 63 - __clzll((long long)(starts & upto));   // first lane of this lane's run
                seg_scan_sum<8>(V8, lane, rs);
                const bool run_end = lane == 63 || ((starts >> ((lane + 1) & 63)) & 1ull);
                if (key >= 0 && run_end) {
                    unsigned long long *d0 = lds_dtf + 4 * key, *d1 = lds_dtf + 4 * key_hi;
#pragma unroll
                    for (int q = 0; q < 4; ++q) { acc_add_f64(d0 + q, V8[q]); acc_add_f64(d1 + q, V8[4 + q]); }
                }
            }
            __syncthreads();   // park / pre are free again
        }
    }
    if (cur_view >= 0) flush();
}

// ------------------------------------------------------------------------------------------------ host
size_t brick_workspace_bytes(int n_views, int W, int H, int VX, int VY, int VZ) {
    const BrickGrid g = make_brick_grid(VX, VY, VZ);
    return ws_layout(nullptr, n_views, W * H, g, nullptr);
}
// Samples per ray the DR_TAPE_TF tape reserves: the longest chord of the box at this sampling rate (VR.py:251-253: n = floor(sr * len *
// |shape - 1|) + 1, len <= 2 sqrt 3), cut at max_samples like the march itself (VR.py:268); even, so that a lane's two samples share
// one 16-byte load.
int tape_stride_for(int VX, int VY, int VZ, float sr, int max_samples) {
    const double diag = sqrt((double)(VX - 1) * (VX - 1) + (double)(VY - 1) * (VY - 1) + (double)(VZ - 1) * (VZ - 1));
    double n_max = floor((double)sr * 2.0 * sqrt(3.0) * diag) + 2.0;
    if (n_max > (double)max_samples) n_max = (double)max_samples;
    if (n_max < 2.0) n_max = 2.0;
    if (n_max > 1.0e9) n_max = 1.0e9;
    return ((int)n_max + 1) & ~1;
}
size_t brick_workspace_bytes_tape(int n_views, int W, int H, int VX, int VY, int VZ, int max_samples, float sr) {
    const BrickGrid g = make_brick_grid(VX, VY, VZ);
    return ws_layout(nullptr, n_views, W * H, g, nullptr, tape_stride_for(VX, VY, VZ, sr, max_samples));
}

// F2: the workspace must hold the F1 output of the same call.
template <typename VT>
static int ray_compose_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    ws_layout(a.workspace, a.n_views, NP, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    // (a forward that leaves a per-sample tape: F2 needs its stride, to keep rays longer than that off the tape path)
    if ((a.hints & DR_TAPE_TF) && a.mode == DR_MODE_DIFF) P.tape_stride = tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S);
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

// B3: after B1 / the tape pass and B2 (what every backward pays: one small launch)
template <typename VT>
static int ray_exact_bwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    Workspace w;
    ws_layout(a.workspace, a.n_views, a.W * a.H, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    if ((a.hints & DR_TAPE_TF) && !a.d_vol) P.tape_stride = tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S);   // (checked against the header)
    const size_t lds = (size_t)a.R * 32 + (size_t)EXACT_BWD_NT * 32;
    const hipError_t e = allow_lds(ray_exact_bwd_kernel<VT>, lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((ray_exact_bwd_kernel<VT>), dim3(DR_EXACT_BWD_GRID), dim3(EXACT_BWD_NT), lds, stream, P);
    return (int)hipGetLastError();
}
int launch_ray_exact_bwd(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_exact_bwd_dispatch<__half>(a, stream) : ray_exact_bwd_dispatch<float>(a, stream);
}

// F3: the rays F2 listed, if any -- a resident grid that reads the count from the workspace header and leaves at once when it is 0
// (what every forward pays: one small launch)
template <typename VT>
static int ray_exact_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    Workspace w;
    ws_layout(a.workspace, a.n_views, a.W * a.H, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    if (a.mode == DR_MODE_DIFF) {
        hipLaunchKernelGGL((ray_exact_kernel<VT, DR_MODE_DIFF, 1024>), dim3(EXACT_FEW), dim3(1024), 0, stream, P);
        hipLaunchKernelGGL((ray_exact_kernel<VT, DR_MODE_DIFF, 256>), dim3(DR_EXACT_GRID), dim3(256), 0, stream, P);
    } else {
        hipLaunchKernelGGL((ray_exact_kernel<VT, DR_MODE_NONDIFF, 1024>), dim3(EXACT_FEW), dim3(1024), 0, stream, P);
        hipLaunchKernelGGL((ray_exact_kernel<VT, DR_MODE_NONDIFF, 256>), dim3(DR_EXACT_GRID), dim3(256), 0, stream, P);
    }
    return (int)hipGetLastError();
}
int launch_ray_exact(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_exact_dispatch<__half>(a, stream) : ray_exact_dispatch<float>(a, stream);
}

template <typename VT>
static int ray_alpha_dispatch(const MarchArgs &a, hipStream_t stream, bool cross) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    ws_layout(a.workspace, a.n_views, NP, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    // The crossing search runs as ~8 192 workgroups over all views (32 768 waves: four rounds of what the chip holds), each staging
    // the TF once and striding over the rays -- not one workgroup per four rays: 131 072 of them for the demo's 8 x 256^2 rays,
    // each paying the TF load and a barrier first (demo loop 10.25 -> 10.20 ms; 2 048 workgroups: +0.1 ms, rays queue up behind
    // long ones). DR_CROSS_GRID = workgroups over all views.
    const int cross_cap = (DR_CROSS_GRID + a.n_views - 1) / a.n_views < 64 ? 64 : (DR_CROSS_GRID + a.n_views - 1) / a.n_views;
    const dim3 grid2((NP + 255) / 256, a.n_views), grid3((NP + 3) / 4 < cross_cap ? (NP + 3) / 4 : cross_cap, a.n_views);
    const size_t lds3 = (size_t)a.R * 16;
    // below sampling rate 2 a crossing segment is short: four rays per wave (ray_cross_quad_kernel)
    const bool quad = a.sr < DR_CROSS_QUAD_BELOW;
    const dim3 grid3q((NP + 15) / 16 < cross_cap ? (NP + 15) / 16 : cross_cap, a.n_views);
    if (a.mode == DR_MODE_DIFF) {
        if (!cross) hipLaunchKernelGGL((ray_alpha_kernel<VT, DR_MODE_DIFF>), grid2, dim3(256), 0, stream, P);
        else if (quad) hipLaunchKernelGGL((ray_cross_quad_kernel<VT, DR_MODE_DIFF>), grid3q, dim3(256), lds3, stream, P);
        else hipLaunchKernelGGL((ray_cross_kernel<VT, DR_MODE_DIFF>), grid3, dim3(256), lds3, stream, P);
    } else {
        if (!cross) hipLaunchKernelGGL((ray_alpha_kernel<VT, DR_MODE_NONDIFF>), grid2, dim3(256), 0, stream, P);
        else if (quad) hipLaunchKernelGGL((ray_cross_quad_kernel<VT, DR_MODE_NONDIFF>), grid3q, dim3(256), lds3, stream, P);
        else hipLaunchKernelGGL((ray_cross_kernel<VT, DR_MODE_NONDIFF>), grid3, dim3(256), lds3, stream, P);
    }
    return (int)hipGetLastError();
}

int launch_ray_alpha(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_alpha_dispatch<__half>(a, stream, false) : ray_alpha_dispatch<float>(a, stream, false);
}
int launch_ray_cross(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? ray_alpha_dispatch<__half>(a, stream, true) : ray_alpha_dispatch<float>(a, stream, true);
}

}  // namespace dr
