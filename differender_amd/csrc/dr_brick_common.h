// dr_brick_common.h -- pieces shared by the brick-centric kernel files (march_flat.hip: the per-brick passes, one
// lane per sample; ray_passes.hip: the per-ray passes): parameters, brick geometry, tap evaluation, 64-bit
// fixed-point LDS accumulation, workspace layout.
#pragma once
#include "dr_brick.h"
#include "dr_kernels.h"
#include "../../include/differender_hip.h"
#include <string.h>

namespace dr {

// Overflow work item: candidates [c0, c1) of one brick of one view (march_flat.hip, brick_flat_items_kernel).
struct BrickItem { int view, brick, c0, c1; };   // brick: its DISPATCH SLOT (the index of its record, see near_first_brick)
constexpr int ITEM_CAP = 1 << 18;     // items the workspace holds, 4 MB (a 512^2 view from inside a 512^3 volume makes ~160 000;
                                      // items beyond the cap are dropped: their rays fail the count check and are marched whole)
constexpr int CTX_MAIN_CAND = 1024;   // candidates of a brick the main launch takes (= MAIN_CAND of march_flat.hip)
constexpr int CTX_ITEM_MAX_CAND = 4096;  // candidates per item at most (its list of hits: 16-bit offsets, 8 KB of LDS -- two backward workgroups still fit a CU)

constexpr int DTF64_R = 2048;   // TF entries the workspace's double d_tf table has room for (the fast path serves R <= 2030)
template <typename VT>
struct BrickParams {
    VolView<VT> vol; int64_t vol_vs;
    const float4 *tf; int64_t tf_vs; int R; float tf_len;
    const float *cam, *entry, *exit_, *rays; const int32_t *nsamp;
    int W, H, S; float sr, inv_sr;
    int imgW, row0;      // band of a wider image (see MarchArgs)
    float near_, near_w, near_h;
    BrickGrid g;
    float4 *seg_rgba;    // [view][NL][NP]: F1 partial composite, then (F2) prefix before the segment
    uint16_t *seg_cnt;   // [view][NL][NP]: samples of the ray inside the brick of that layer (low 15 bits, saturating at 32767: a
                         // longer in-brick run -- sampling rates in the thousands -- fails the count check and the
                         // ray is marched whole) | SEG_CNT_TINY
    uint16_t *seg_tiny;  // [view][NL][NP]: how many of those samples have a TINY opacity (0 < op < DR_D4_TINY_OP); valid where seg_cnt
                         // carries SEG_CNT_TINY (never cleared: read only under that flag)
    uint8_t *rayflag;    // [view][NP]: 1 = irregular ray (marched whole by F2 / B2), 2 = recomputed by F3, its backward by B3
    int32_t *ws_steps;   // [view][NP]: live samples per ray (F2 -> B1)
    float4 *fin;         // [view][NP]: the ray's composite as F2 put it together from the partials (differentiable march): the backward's
                         //   tape-free identity needs a final value that is CONSISTENT with the stored prefixes -- the image itself may
                         //   have been recomputed sample by sample (ray_exact_kernel)
    unsigned int *exact_list;  // [view * NP]: rays F2 sent to ray_exact_kernel (stats[ST_EXACT_RAYS] of them)
    double *dtf64;       // [view][DTF64_R][4]: d_tf of ONE backward call, summed across its workgroups in double (global f64 atomics) and
                         //   committed to the caller's float tensor by dtf_commit_kernel, which leaves it zero again (the forward zeroes it too).
                         //   80 000 per-brick partials met in float atomics before: 1e-4 of d_tf's maximum where a texel collects 1e8
                         //   cancelling terms (the air's texels under a random upstream gradient: tools/diff_sweep.py, round 6)
    float2 *tape;        // DR_TAPE_TF: [view][NP][tape_stride] (intensity, lighting term) of every marched sample (forward -> tf_tape.hip)
    int tape_stride;     //   samples reserved per ray (0: no tape)
    unsigned long long *unlit;  // [view][lm_words][NP]: non-differentiable renders with an alpha pre-pass -- bit l of a ray's mask: the pre-pass
    int lm_words;               //   marched the ray's segment of layer l and found NO sample with alpha > 1e-3 (the colour march
                                //   skips it: its count and its zero partial are already in place). 0: feature off (NL > 128).
    unsigned int *stats; // workspace header, words ST_* below
    unsigned int *vflags; // [view] 1 if some ray of the view may reach alpha >= 0.99 (the alpha pre-pass runs for it);
                          // [n_views + view] 1 if the camera sits inside (or on) the volume: rays start behind the eye, the
                          // camera-based brick layers are not monotone along them and the pre-pass runs as ONE phase
    int n_views;
    unsigned int mark;   // fingerprint of this forward/backward pair (ws_fingerprint): the forward leaves it in stats[ST_MARK],
                         // the backward trusts records, live flags, ray flags and work items only if it finds it there
    int nondiff;         // forward: non-differentiable march (a sample composites only if alpha > 1e-3, VR.py:334; differentiable: if alpha != 0)
    int hint_noterm;     // forward: the caller said no ray can terminate early and the pre-pass was not launched; F2 checks
    int use_live;        // forward: ws_steps holds each ray's exact live sample count (from the alpha pre-pass)
    int count_eval;      // DR_COUNT_EVALUATED: the brick kernels add their evaluated samples to stats[ST_EVAL_*] (measurement only)
    int pp_l0, pp_l1, pp_first;  // alpha pre-pass phase: brick layers [pp_l0, pp_l1); later phases skip terminated rays
    const struct BrickCtxRec *ctx;  // [view][brick]: brick geometry + pixel rectangle, filled once per forward call
    BrickItem *items;               // overflow work items of heavy bricks
    unsigned int *n_items;          // ... and how many there are
    float *out; int32_t *steps;
    const float *grad_out, *out_fwd;
    GradView dvol; int64_t dvol_vs;
    float *d_tf; int64_t dtf_vs;
};

// Workspace header (32-bit words). Written by the device only; read back by functional.workspace_stats().
enum {
    ST_REPAIR = 0,         // rays whose segments failed the sample-count check and were marched whole (expected 0)
    ST_BASELINE_RAYS = 2,  // rays the per-ray fallback marched in the last forward (irregular rays + repaired ones)
    ST_MARK = 3,           // fingerprint of the flat forward that wrote the brick records, flags, items and coarse tape
    ST_TICKET = 6,         // work-item launches: next item to hand out / workgroups done (the last one resets both to 0)
    ST_DONE = 7,
    ST_NITEMS = 5,         // diagnostic copy of the number of overflow work items of the last forward (heavy bricks: BrickItem)
    ST_HINT_BAD = 8,       // forward: views for which DR_HINT_NO_EARLY_TERMINATION was wrong (their rays were marched one by one)
    ST_UNLIT_SKIPPED = 12, // forward: (ray, layer) segments the colour march dropped because the alpha pre-pass had found them unlit ...
    ST_EMPTY_BRICKS = 13,  // ... and (view, brick) workgroups (all passes) that took the empty-brick path: both SAMPLED, every 64th workgroup reports
    ST_MASKS = 14,         // forward (differentiable): words per ray of "unlit" layer masks it left behind seg_cnt (0: none) -- read by the
                           // d_volume-only backward
    ST_EXACT_RAYS = 15,    // forward: rays whose image value was recomputed in the reference's sequential float32 order (ray_exact_kernel:
                           // contributions of the size of an ulp of the running composite)
    ST_STALE_BWD = 9,      // backward calls that did not find their forward's fingerprint here and marched every ray one by one
    ST_F64_BRICKS = 4,     // backward: (view, brick) pairs whose d_volume box accumulated in double (wide local range of |grad_out|)
    ST_TAPE_STRIDE = 56,   // forward: samples per ray of the DR_TAPE_TF tape it left behind the workspace (0: none)
    ST_EVAL_PRE = 58, ST_EVAL_FWD = 60, ST_EVAL_BWD = 62,   // DR_COUNT_EVALUATED: u64 each -- samples whose taps the alpha pre-pass / the colour march / the backward evaluated
    ST_TIMING = 16,        // words 16-55: clocks / counters of diagnostic builds (tools/patches/closed_experiments.patch), the pixel to trace of a D4 debug build
    ST_WORDS = 512         // header size in words (2 KiB)
};

constexpr int SEG_CNT_MAX = 0x7fff, SEG_CNT_TINY = 0x8000;   // seg_cnt: count | "seg_tiny holds a count"

struct BrickCtx {
    int bx, by, bz, layer;        // layer: camera-based (pre-pass phases only; see dr_brick.h)
    int ox, oy, oz;               // voxel index of LDS box element 0 along each axis (12*b - 1)
    float lo[3], hi[3];           // world AABB of the brick's cells, with slack
    int i0, i1, j0, j1;           // candidate pixel rectangle (inclusive); empty if i0 > i1
    int live;                     // from the workspace record (see BrickCtxRec)
    int maybe_empty;              // forward: bit 0 = nine probes of the brick found nothing that composites (brick_ctx_kernel); bit 1 = the
                                  // workgroup that staged it DECIDED it empty (brick_empty_decide) -- the d_volume-only backward skips it
};

// BrickCtx as stored in the workspace (64 B): every workgroup of F1 / P1 / B1 reads its record with scalar loads
// instead of re-deriving it (8 corner projections, ~700 VALU per wave).
struct BrickCtxRec {
    int bx, by, bz, layer;
    float lo[3]; int i0;
    float hi[3]; int i1;
    int j0, j1;
    int live;   // set by the flat forward when the brick marched at least one sample of the view (backward skips the others)
    int maybe_empty;   // forward: the brick's corner and centre voxels all map to transfer-function texels that composite nothing
};
__device__ __forceinline__ void brick_ctx_load(const BrickCtxRec *rec, BrickCtx &c) {
    const BrickCtxRec r = *rec;
    c.bx = r.bx; c.by = r.by; c.bz = r.bz; c.layer = r.layer;
    c.ox = c.bx * BRK - 1; c.oy = c.by * BRK - 1; c.oz = c.bz * BRK - 1;
    for (int k = 0; k < 3; ++k) { c.lo[k] = r.lo[k]; c.hi[k] = r.hi[k]; }
    c.i0 = r.i0; c.i1 = r.i1; c.j0 = r.j0; c.j1 = r.j1; c.live = r.live; c.maybe_empty = r.maybe_empty;
}

#ifndef DR_RECT_SLACK
#define DR_RECT_SLACK 0.05f
#endif
__device__ __forceinline__ f3 cross3b(f3 a, f3 b) {
    return make_f3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// Brick geometry + its projected pixel rectangle (pinhole model of VR.py:127-151).
template <typename VT>
__device__ __forceinline__ void brick_setup(const BrickParams<VT> &P, int b, f3 cam, BrickCtx &c) {
    const BrickGrid &g = P.g;
    c.bz = b % g.NBz; c.by = (b / g.NBz) % g.NBy; c.bx = b / (g.NBz * g.NBy);
    c.ox = c.bx * BRK - 1; c.oy = c.by * BRK - 1; c.oz = c.bz * BRK - 1;
    c.layer = camera_layer(c.bx, c.by, c.bz, cam, P.vol.scx, P.vol.scy, P.vol.scz, g.NBx, g.NBy, g.NBz);
    const int bb[3] = {c.bx, c.by, c.bz};
    const int nb[3] = {g.NBx, g.NBy, g.NBz};
    const float sc[3] = {P.vol.scx, P.vol.scy, P.vol.scz};
    for (int k = 0; k < 3; ++k) {
        // cell range [16b, 16b+16) in scaled coordinates q = (0.5 x + 0.5) sc  ->  x = 2 q / sc - 1
        float lo = 2.0f * (float)(bb[k] * BRK) / sc[k] - 1.0f;
        float hi = 2.0f * (float)(bb[k] * BRK + BRK) / sc[k] - 1.0f;
        if (bb[k] == 0) lo = -1.0f;          // positions are clamped into the edge cells
        if (bb[k] == nb[k] - 1) hi = 1.0f;
        c.lo[k] = lo - BRICK_EPS; c.hi[k] = hi + BRICK_EPS;
    }
    const f3 vdir = normalized3(make_f3(-cam.x, -cam.y, -cam.z));
    const f3 right = normalized3(cross3b(vdir, make_f3(0.f, 1.f, 0.f)));
    const f3 up = normalized3(cross3b(right, vdir));
    // Which pixels' LINES (the reference marches from tmin even when it is negative: a camera inside the volume sees
    // samples behind the eye, VR.py:28-53) can meet the brick? The pinhole map X -> (u, v) is the same central projection
    // on both sides of the eye, so the part of the brick in front of it (depth >= eps) and the part behind it
    // (depth <= -eps) are each convex and project into the bounding box of their vertices: the brick's corners on that
    // side plus the points where its 12 edges cross the plane depth = +-eps (those land far outside the image unless the
    // edge passes close to the eye, and are clipped to it below). The sliver |depth| < eps holds less than a sample
    // spacing of any line; a miss there would be caught by the per-ray sample-count check.
    float pxmin = 1e30f, pxmax = -1e30f, pymin = 1e30f, pymax = -1e30f;
    const float eps = 1e-3f;
    const float ku = P.near_ / P.near_w, kv = P.near_ / P.near_h;
    f3 dk[8];
    float dep[8], du[8], dv[8];
    for (int k = 0; k < 8; ++k) {
        dk[k] = make_f3(((k & 1) ? c.hi[0] : c.lo[0]) - cam.x, ((k & 2) ? c.hi[1] : c.lo[1]) - cam.y,
                        ((k & 4) ? c.hi[2] : c.lo[2]) - cam.z);
        dep[k] = dot3(dk[k], vdir); du[k] = dot3(dk[k], right); dv[k] = dot3(dk[k], up);
    }
    auto project = [&](float r, float u_, float depth) {  // (r, u_) = components along right / up
        const float inv = 1.0f / depth;
        const float px = (r * inv * ku + 0.5f) * (float)P.imgW - 0.5f, py = (u_ * inv * kv + 0.5f) * (float)P.H - 0.5f;  // full-image rows
        pxmin = fminf(pxmin, px); pxmax = fmaxf(pxmax, px); pymin = fminf(pymin, py); pymax = fmaxf(pymax, py);
    };
    for (int k = 0; k < 8; ++k)
        if (fabsf(dep[k]) >= eps) project(du[k], dv[k], dep[k]);
    for (int k = 0; k < 8; ++k) {
        for (int bit = 1; bit < 8; bit <<= 1) {
            const int j = k | bit;
            if (j == k) continue;  // each edge once: from the corner with the bit clear
            for (int sgn = -1; sgn <= 1; sgn += 2) {
                const float pl = (float)sgn * eps;
                if ((dep[k] - pl) * (dep[j] - pl) < 0.0f) {
                    const float t = (pl - dep[k]) / (dep[j] - dep[k]);
                    project(fmaf(t, du[j] - du[k], du[k]), fmaf(t, dv[j] - dv[k], dv[k]), pl);
                }
            }
        }
    }
    if (!(pxmin <= pxmax)) { c.i0 = 0; c.i1 = P.W - 1; c.j0 = 0; c.j1 = P.H - 1; return; }  // (a brick thinner than the sliver)
    pxmin = fmaxf(pxmin, -2.0f); pymin = fmaxf(pymin, -2.0f);
    pxmax = fminf(pxmax, (float)P.imgW + 2.0f); pymax = fminf(pymax, (float)P.H + 2.0f);
    // Pixel (i, j) has its ray through (px, py) = (i, j) exactly (the inverse of ray_setup's mapping), and the brick
    // (a convex box in front of the camera, already widened by BRICK_EPS) projects inside the bounding box of its
    // corners: the candidates are the integer points of [pxmin, pxmax] x [pymin, pymax]. DR_RECT_SLACK (pixels) covers
    // the float error of this projection (~1e-4 px); a miss would still be caught by the per-ray sample-count check.
    c.i0 = max(0, (int)ceilf(pxmin - DR_RECT_SLACK) - P.row0); c.i1 = min(P.W - 1, (int)floorf(pxmax + DR_RECT_SLACK) - P.row0);  // band rows
    c.j0 = max(0, (int)ceilf(pymin - DR_RECT_SLACK)); c.j1 = min(P.H - 1, (int)floorf(pymax + DR_RECT_SLACK));
}

// Dispatch order: workgroup i of a view takes the i-th brick counted from the corner of the volume NEAREST the camera
// (perspective puts the most rays, hence the most samples, into the bricks close to the eye: with the plain index order
// an orbit camera on the +x side had its heaviest bricks dispatched last, and the launch ended on a few long workgroups).
template <typename VT>
__device__ __forceinline__ int near_first_brick(const BrickParams<VT> &P, int i, int view) {
    const int NBx = P.g.NBx, NBy = P.g.NBy, NBz = P.g.NBz;
    int iz = i % NBz, iy = (i / NBz) % NBy, ix = i / (NBz * NBy);
    if (P.cam[3 * view] > 0.0f) ix = NBx - 1 - ix;
    if (P.cam[3 * view + 1] > 0.0f) iy = NBy - 1 - iy;
    if (P.cam[3 * view + 2] > 0.0f) iz = NBz - 1 - iz;
    return (ix * NBy + iy) * NBz + iz;
}

// The records are STORED in that order (record i of a view = the brick dispatched i-th), so that a workgroup's first load
// does not wait for the camera position; a work item names its brick by the same dispatch slot.
// One thread per (brick, view): brick_setup once, for all passes of a forward/backward pair. The forward leaves a
// FINGERPRINT of its call in stats[ST_MARK] (ws_fingerprint: sizes, sampling rate, the addresses of the ray buffers and of
// the volume); the backward does not re-derive anything -- records, live flags, ray flags, work items and the coarse tape
// are the forward's -- and therefore only touches them when it finds the fingerprint of ITS OWN arguments there. A
// workspace that holds something else (never filled, filled by another call, forward served by the baseline kernels, which
// clear the mark) makes B1 return at once and B2 march every ray: slow, correct, and no index read from garbage.
static_assert(ST_MARK == WS_MARK_WORD, "capi.hip clears this word through flat_invalidate_workspace");
static_assert(ST_STALE_BWD == WS_MARK_WORD + 6, "march_baseline.hip counts stale backward calls at ws_mark[6]");
static inline unsigned int ws_fingerprint(const MarchArgs &a) {
    unsigned long long h = 0xcbf29ce484222325ull;
    auto mix = [&h](unsigned long long v) { for (int k = 0; k < 8; ++k) { h ^= (v >> (8 * k)) & 0xffu; h *= 0x100000001b3ull; } };
    mix((unsigned long long)a.n_views); mix((unsigned long long)a.W); mix((unsigned long long)a.H);
    mix((unsigned long long)(a.img_W > 0 ? a.img_W : a.W)); mix((unsigned long long)(a.img_W > 0 ? a.row0 : 0));
    mix((unsigned long long)a.VX); mix((unsigned long long)a.VY); mix((unsigned long long)a.VZ);
    mix((unsigned long long)a.R); mix((unsigned long long)a.S);
    unsigned int srb; memcpy(&srb, &a.sr, 4); mix(srb);
    mix((unsigned long long)a.vol_dtype); mix((unsigned long long)a.sx); mix((unsigned long long)a.sy); mix((unsigned long long)a.sz);
    mix((unsigned long long)a.vol_vs);
    mix((unsigned long long)(uintptr_t)a.vol); mix((unsigned long long)(uintptr_t)a.entry); mix((unsigned long long)(uintptr_t)a.exit_);
    mix((unsigned long long)(uintptr_t)a.rays); mix((unsigned long long)(uintptr_t)a.nsamp);
    // (not the camera's address: hosts hand over `cam.contiguous()` -- a fresh temporary per call for an expanded or
    //  converted look_from -- and a backward that did not recognise its own forward would silently run 10-40 x slower;
    //  the ray buffers, which are derived from the camera, identify the call)
    const unsigned int f = (unsigned int)(h ^ (h >> 32));
    return f ? f : 1u;   // 0 = "nobody's" (what flat_invalidate_workspace writes)
}
// Can any ray of a view terminate early? Upper bound from the largest TF alpha: after n_max samples of opacity
// op_max the accumulated alpha is 1 - (1 - op_max)^n_max. Evaluated by one wave.
__device__ __forceinline__ unsigned int may_terminate(const float4 *t, int R, float inv_sr, float n_max) {
    const int lane = threadIdx.x & 63;
    float amax = 0.0f;
    for (int k = lane; k < R; k += 64) {
        const float a = t[k].w;
        amax = (a != a) ? 1.0f : fmaxf(amax, a);  // NaN alpha: assume anything can happen
    }
    for (int o = 32; o > 0; o >>= 1) amax = fmaxf(amax, __shfl_xor(amax, o));
    const float op = 1.0f - powf(fmaxf(1.0f - fminf(amax, 1.0f), 0.0f), inv_sr);
    const float remain = powf(1.0f - op, n_max);  // transmittance left after the longest possible ray
    return (remain <= 0.02f) ? 1u : 0u;           // 0.01 is the exact bound; keep a margin
}
// ---- Empty bricks (round 5) ------------------------------------------------------------------------------------------------
// A sample composites nothing when its TF alpha is <= 1e-3 (non-differentiable march, VR.py:334: the sample is skipped) or
// exactly 0 (differentiable march: opacity 0, colour L * rgb * 0). CT-like data under the reference's presets is mostly such
// samples -- air. A brick ALL of whose samples composite nothing needs no tap, no TF lookup, no shading: only its rays' sample
// counts (F2's check) and all-zero partials. Deciding that exactly takes the brick's voxel range, which the workgroup has when
// its box is staged (brick_empty_test in march_flat.hip); whether the test is worth running is guessed here, once per brick,
// from nine voxels: a brick whose corners and centre all composite nothing is a candidate.
__device__ __forceinline__ bool texel_composites(float a, int nondiff) { return nondiff ? !(a <= 1e-3f * (1.0f - 4.8e-7f)) : !(a == 0.0f); }
template <typename VT>
__device__ __forceinline__ int brick_probe_empty(const BrickParams<VT> &P, int view, const BrickCtx &c) {
    const VT *vp = P.vol.p + view * P.vol_vs;
    const float4 *tf = P.tf + view * P.tf_vs;
    bool ok = true;
    for (int k = 0; k < 9 && ok; ++k) {
        // corners of the brick's cell range (voxels ox+1 .. ox+1+BRK, clamped into the volume) and its centre
        const int dx = k == 8 ? BRK / 2 : ((k & 1) ? BRK : 0), dy = k == 8 ? BRK / 2 : ((k & 2) ? BRK : 0), dz = k == 8 ? BRK / 2 : ((k & 4) ? BRK : 0);
        const int x = min(c.ox + 1 + dx, P.vol.VX - 1), y = min(c.oy + 1 + dy, P.vol.VY - 1), z = min(c.oz + 1 + dz, P.vol.VZ - 1);
        const float v = ld_voxel(vp + ((long long)x * P.vol.sx + (long long)y * P.vol.sy + (long long)z * P.vol.sz));
        int lo; float fr;
        low_high_frac(v * P.tf_len, lo, fr);
        lo = min(lo, P.R - 1);
        ok = (v == v) && !texel_composites(tf[lo].w, P.nondiff) && !texel_composites(tf[min(lo + 1, P.R - 1)].w, P.nondiff);
    }
    return ok ? 1 : 0;
}

// Also initialises the workspace header for this call: counters, per-view "may terminate" flags (n_max > 0: the alpha
// pre-pass is available) and the mark -- no separate memset / flag kernel. Runs once per forward; the backward of the same
// inputs reuses the records (and the live flags the forward march leaves in them).
template <typename VT>
static __global__ __launch_bounds__(256) void brick_ctx_kernel(BrickParams<VT> P, BrickCtxRec *out, int nbricks, int forward,
                                                              float n_max) {
    const int b = blockIdx.x * 256 + threadIdx.x, view = blockIdx.y;
    if (blockIdx.x == 0 && threadIdx.x < 64) {  // first wave of the view's first block
        if (forward) {
            const unsigned int flag = n_max > 0.0f ? may_terminate(P.tf + view * P.tf_vs, P.R, P.inv_sr, n_max) : 0u;
            if (threadIdx.x == 0) {
                const float cx = P.cam[3 * view], cy = P.cam[3 * view + 1], cz = P.cam[3 * view + 2];
                const float lim = 1.0f + 1e-3f;
                P.vflags[view] = flag;
                P.vflags[P.n_views + view] = (fabsf(cx) <= lim && fabsf(cy) <= lim && fabsf(cz) <= lim) ? 1u : 0u;
                if (view == 0) { P.stats[ST_REPAIR] = 0u; P.stats[ST_BASELINE_RAYS] = 0u; P.stats[ST_HINT_BAD] = 0u; P.stats[ST_STALE_BWD] = 0u; P.stats[ST_F64_BRICKS] = 0u; P.stats[ST_TICKET] = 0u; P.stats[ST_DONE] = 0u; P.stats[ST_UNLIT_SKIPPED] = 0u; P.stats[ST_EMPTY_BRICKS] = 0u; P.stats[ST_EXACT_RAYS] = 0u; P.stats[ST_TAPE_STRIDE] = P.tape ? (unsigned int)P.tape_stride : 0u; for (int k = ST_EVAL_PRE; k < ST_EVAL_PRE + 6; ++k) P.stats[k] = 0u; P.stats[ST_MASKS] = P.nondiff ? 0u : (unsigned int)P.lm_words; P.stats[ST_MARK] = P.mark; }
            }
        }
    }
    if (forward)   // the double d_tf table of the backward that follows starts at zero (dtf_commit_kernel leaves it so as well)
        for (int k = b; k < 4 * DTF64_R; k += (int)gridDim.x * 256) P.dtf64[(size_t)view * 4 * DTF64_R + k] = 0.0;
    if (b >= nbricks) return;
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    BrickCtx c;
    brick_setup(P, b, cam, c);
    BrickCtxRec r;
    r.bx = c.bx; r.by = c.by; r.bz = c.bz; r.layer = c.layer;
    for (int k = 0; k < 3; ++k) { r.lo[k] = c.lo[k]; r.hi[k] = c.hi[k]; }
    r.i0 = c.i0; r.i1 = c.i1; r.j0 = c.j0; r.j1 = c.j1;
    r.maybe_empty = forward ? brick_probe_empty(P, view, c) : 0;
    const int slot_b = near_first_brick(P, b, view);  // (an involution: slot -> brick and brick -> slot are the same flips)
    r.live = forward ? 0 : out[(size_t)view * nbricks + slot_b].live;
    out[(size_t)view * nbricks + slot_b] = r;
    if (forward && c.i0 <= c.i1 && c.j0 <= c.j1) {
        // a brick with more candidate pixels than the main launch takes: cut the rest into work items
        // (item size: 1024 to 4096 candidates, a full-image brick makes at most 64 items up to 512^2 pixels; most candidates of
        // such bricks fail the item's geometric pre-test and cost nothing further)
        const int ncand = (c.i1 - c.i0 + 1) * (c.j1 - c.j0 + 1);
        const int item_cand = min(max(1024, ((P.W * P.H / 64) + 255) & ~255), CTX_ITEM_MAX_CAND);
        // (one reservation per brick: its items are consecutive, so that a workgroup taking a run of items keeps the brick's
        // voxel box staged across them)
        const int n_it = ncand > CTX_MAIN_CAND ? (ncand - CTX_MAIN_CAND + item_cand - 1) / item_cand : 0;
        unsigned int slot = n_it ? atomicAdd(P.n_items, (unsigned int)n_it) : 0u;
        for (int c0 = CTX_MAIN_CAND; c0 < ncand; c0 += item_cand, ++slot)
            if (slot < (unsigned int)ITEM_CAP) P.items[slot] = BrickItem{view, slot_b, c0, min(c0 + item_cand, ncand)};
    }
}

// Conservative sample-index range [s0, s1) of ray p inside the brick (exact membership is decided per
// sample from its cell). Returns false when the ray cannot touch the brick.
__device__ __forceinline__ bool segment_range(const BrickCtx &c, f3 cam, f3 vd, float t0, float exit_, int n,
                                              int nmarch, int &s0, int &s1) {
    const float o[3] = {cam.x, cam.y, cam.z}, d[3] = {vd.x, vd.y, vd.z};
    float ta = -1e30f, tb = 1e30f;
    for (int k = 0; k < 3; ++k) {
        if (fabsf(d[k]) < 1e-12f) {
            if (o[k] < c.lo[k] || o[k] > c.hi[k]) return false;
        } else {
            const float inv = __builtin_amdgcn_rcpf(d[k]);  // 1 ulp: the range is conservative (BRICK_EPS, +1 sample)
            const float t1 = (c.lo[k] - o[k]) * inv, t2 = (c.hi[k] - o[k]) * inv;
            ta = fmaxf(ta, fminf(t1, t2)); tb = fminf(tb, fmaxf(t1, t2));
        }
    }
    if (!(ta <= tb)) return false;
    const float scale = (float)(n - 1) / (exit_ - t0);
    // samples with ta <= t_s <= tb: ceil(xa) .. floor(xb). The slab carries BRICK_EPS (0.045 samples at 512^3, never
    // below 0.005) of slack, an order of magnitude above the rounding of this inverse map (<= 1e-7 * n samples). A miss would
    // still be caught by the per-ray sample-count check (and the ray marched whole).
    float sa = ceilf((ta - t0) * scale), sb = floorf((tb - t0) * scale) + 1.0f;
    sa = fminf(fmaxf(sa, 0.0f), (float)nmarch); sb = fminf(fmaxf(sb, 0.0f), (float)nmarch);
    s0 = (int)sa; s1 = (int)sb;
    return s1 > s0;
}

// ---- Sequential float32 compositing and the partials ----------------------------------------------------------
// The reference composites one sample at a time, C <- fma(T, c_s, C) (VR.py:300-302): each contribution is rounded to a whole
// number of ulps of the running composite -- dropped altogether below half an ulp. While consecutive contributions stay alike to
// within a fraction of an ulp they are rounded the SAME way, sample after sample, and the errors add up linearly (a transparent
// range at alpha 1e-6 behind an opaque structure: 1e-8 onto 0.8, all dropped; 0.67 ulp each: every one rounded UP to a whole ulp);
// contributions that drift through several ulps from sample to sample round this way and that, and that random walk is what the
// 1e-5 bar (and the crossing search's 2e-6 band) already absorb. The brick kernels sum a segment's samples among themselves first
// and reproduce none of this. The per-ray passes therefore BOUND |sequential - partials| along each ray, from two things:
//   (1) the brick passes count, per (ray, layer) segment, its samples of TINY opacity, 0 < op < DR_D4_TINY_OP (seg_tiny; flag
//       SEG_CNT_TINY in seg_cnt) -- the samples whose contributions can be of the size of an ulp at all (T >= 0.01 before a ray
//       terminates, an ulp of a composite below 1 is 6e-8: op >= 1e-4 contributes 17 ulps and more). Each may be off by half
//       an ulp of the running value, in every channel: tiny * ulp / 2 -- linearly from DR_D4_TINY_RUN such samples in a segment
//       on (a stretch of like samples), in quadrature below that (the foot of a TF ramp: opacities that sweep up from 0 round this
//       way and that). (Segments of 50-180 samples MIX such stretches with lit samples: their means say nothing -- the first
//       design of this bound, which looked at means only, missed exactly those.)
//   (2) the segment's MEAN contribution per sample and how far it moved over the ray's last three segments: a mean of at least half
//       an ulp that stays put to within half an ulp (a flat region: constant intensity, ambient lighting) is rounded the same way
//       throughout: min(contribution, samples * ulp / 2), added linearly like (1). Two segments alone do not count: along a smooth
//       ray the contributions pass through extrema, where neighbouring segments share a mean while their samples still vary by
//       many ulps. (A mean that DRIFTS leaves a sawtooth of roundings that cancels but for a partial cycle of random sign; charging
//       those remainders was tried and flagged thousands of rays whose true error was below 1e-6 -- a segment's sample count says
//       nothing about how many of its samples contribute at all: profiles/r06_ab_experiments.txt. They belong to the random walk.)
//   (3) long rays: the random walk of the sequential roundings alone (0.29 ulp per sample; the ulp of the ray's largest final
//       channel bounds every running value's) leaves no room under the 1e-5 bar once 3 sigma = 0.87 ulp sqrt(samples) passes
//       DR_D4_WALK = 5.7e-6 -- 12 000 samples of a composite in [0.5, 1), 3 000 of a colour in [1, 2) (unclamped highlights of the
//       differentiable march), 48 000 of a dim one: recomputed whatever (1) and (2) say.
// An error in alpha also moves every later contribution: bound(alpha) * later partial (the callers add that). A ray whose bound
// exceeds DR_D4_BUDGET has its pixel recomputed sample by sample (ray_exact_kernel); in the crossing search the bound widens the
// band inside which the early-termination decision is repeated exactly.
// Not covered: colour contributions below half an ulp from samples of ORDINARY opacity (a nearly black TF colour behind a bright
// structure): their number is not known per segment, and a small mean alone cannot tell them from a sparse segment.
#ifdef DR_D4_BUDGET_OVERRIDE   // (what-if builds: a huge budget switches the exact pass off -- marked in dr_experiment.h)
#define DR_D4_BUDGET DR_D4_BUDGET_OVERRIDE
#else
#define DR_D4_BUDGET 3.0e-6f
#endif
// ... and a ray whose bound exceeds DR_D4_BWD_BUDGET has its BACKWARD formed from the sequential composites too (ray_exact_bwd_kernel):
// the tape-free identity takes prefixes and final value from the composites, an adjoint built on partials that are `bound` away is
// ~7 bound off in relative terms -- per RAY, and a voxel of d_volume hears from a handful of rays only. Measured (tools/d4_bound_hist.py,
// 512^3): the bench TF's listed rays, 0-4 per camera, have bounds of 3e-6 .. 5e-6 and keep the brick backward (nothing to pay at the
// headline); a TF with tiny alphas all over has 2 500 rays per view there, 10 000 between 5e-6 and 1e-5, 60 000 above. (At 1e-5 the
// real-size sweep still found d_volume 1.2e-4 .. 2.7e-4 off in 0.25 % of such configurations.)
#define DR_D4_BWD_BUDGET 5.0e-6f
#define DR_D4_TINY_OP 1.0e-4f
#define DR_D4_TINY_RUN 16.0f
#define DR_D4_WALK 5.68e-6f   // = 0.87 x 5.96e-8 x sqrt(12 000)
__device__ __forceinline__ float ulp_of(float x) {   // spacing of the floats at |x| (0 below 2^-100: nothing to lose there)
    const unsigned int e = __float_as_uint(x) & 0x7f800000u;
    return e > (27u << 23) ? __uint_as_float(e - (23u << 23)) : 0.0f;
}
// One channel's running bound: `lin` adds up linearly, `sq` is a sum of squares.
struct D4Bound {
    float lin, sq, mprev, dprev;   // mprev: the mean contribution per sample of the ray's previous segment; dprev: how far THAT had moved
    __device__ __forceinline__ float total() const { return lin + 2.0f * __builtin_amdgcn_sqrtf(sq); }
};
__device__ __forceinline__ D4Bound d4_zero() { D4Bound b; b.lin = b.sq = b.mprev = 0.0f; b.dprev = 3.0e38f; return b; }
// `contrib` = T * partial of the channel over `cnt` samples (rcnt = 1 / cnt), `tiny` of them of tiny opacity, taking the running
// value to `post`. The ulp is that of the value AFTER the segment -- the largest the running value gets inside it: a segment that
// builds most of the composite itself (a ray that spends 800 samples in its first brick) rounds against its own growing sum, not
// against the prefix it started from (fuzz seed 4100169, round 6: alpha off by 1.3e-5 on a ray whose first segment held 660 of
// its 861 samples, all of tiny opacity, on a prefix of 0).
__device__ __forceinline__ void d4_risk(float post, float contrib, float cnt, float rcnt, float tiny, D4Bound &b) {
    const float hu = 0.5f * ulp_of(post), c = fabsf(contrib);
    const float m = c * rcnt;
    // drift of the mean: the LARGER of the last two differences -- along a smooth ray the contributions pass through extrema, where two
    // neighbouring segments have the same mean by coincidence while the samples inside them still vary by many ulps; only a run of
    // three segments alike (a flat region, a constant tiny alpha) is taken for contributions that are rounded alike
    const float d1 = fabsf(m - b.mprev);
    const float d = fmaxf(d1, b.dprev);
    b.mprev = m; b.dprev = d1;
    const float worst = fminf(c, cnt * hu);          // every sample off by half an ulp, or dropped
    // steady: the mean stays put AND is itself at least half an ulp -- a smaller mean is either a run of tiny samples (counted
    // by `tiny`) or a SPARSE segment, a few ordinary samples among many that contribute nothing (a thin shell, the skipped samples
    // of a non-differentiable march): nothing is rounded alike there
    const bool steady = !(d > hu) && m >= hu;
    // (a handful of tiny samples in a segment is the foot of a TF ramp -- opacities that sweep from 0 upwards round this way and that:
    //  charged in quadrature; from D4_TINY_RUN on they count as a stretch of like samples: linearly)
    const float tq = tiny * hu;
    const bool run = tiny >= DR_D4_TINY_RUN;
    b.lin += fmaxf(run ? tq : 0.0f, steady ? worst : 0.0f);
    b.sq = fmaf(run ? 0.0f : tq, tq, b.sq);
}

__host__ __device__ inline size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

struct TapCoords {
    int lx, ly, lz;       // local (box) cell of the centre tap
    float fx, fy, fz;
    int lxp, lxm, lyp, lym, lzp, lzm;  // local cells of the +-delta taps along each axis
    float fxp, fxm, fyp, fym, fzp, fzm;
};

// Centre tap and the six normal taps from the LDS box, in two halves (the normal taps are skipped when no lane needs the
// lighting term). Until round 4 these were seven independent tri_lds() calls -- 56 LDS words and 49 lerps per sample; the
// arithmetic of each tap is unchanged below, the taps now share what they have in common.
// ---- Seven taps with SHARED partial lerps (round 4) -----------------------------------------------------------------------
// The reference interpolates x -> y -> z (VR.py:173-189). The +-delta taps along y keep the centre tap's x cell and x
// fraction, so their four x-lerps are -- bit for bit -- the centre's own (rows ly, ly+1) plus, when the tap has crossed into the
// neighbouring cell, ONE more row (ly+2 or ly-1); the taps along z keep x and y, so their two bilinear planes are the centre's
// zl, zh plus at most one more plane. With delta < half a voxel (volume edges up to 991: NARROW) at most one of the two taps
// of an axis leaves the centre's cell, so one extra row and one extra plane per sample suffice: 8 + 4 + 4 LDS words and
// 7 + 2+6 + 3+2 lerps for five taps instead of 40 words and 35 lerps, every tap still the oracle's sequence of roundings on the
// oracle's operands. (The taps along x change the FIRST lerp's fraction: nothing of the centre's can be reused but the
// voxels themselves, and choosing them lane by lane costs more selects than the reads it saves: they stay as they were.)
// The forward's LDS traffic is worth 7-10 % of its time (profiles/r04_ab_experiments.txt, half-reads what-if).
struct CentreLerps { float a0, b0, a1, b1, zl, zh; };
__device__ __forceinline__ float sample_centre_lds_keep(const float *box, const TapCoords &t, CentreLerps &c) {
    const int base = t.lx * BOX_SX + t.ly * BOX_SY + t.lz;
    c.a0 = mixf(box[base], box[base + BOX_SX], t.fx);
    c.b0 = mixf(box[base + BOX_SY], box[base + BOX_SX + BOX_SY], t.fx);
    c.zl = mixf(c.a0, c.b0, t.fy);
    c.a1 = mixf(box[base + 1], box[base + BOX_SX + 1], t.fx);
    c.b1 = mixf(box[base + BOX_SY + 1], box[base + BOX_SX + BOX_SY + 1], t.fx);
    c.zh = mixf(c.a1, c.b1, t.fy);
    return mixf(c.zl, c.zh, t.fz);
}
// NARROW: delta below half a voxel -- at most one of an axis's two taps leaves the centre's cell, one extra row / plane is
// read (chosen per lane); otherwise (edges of 992 .. 2000 voxels, e.g. 1024^3: delta = 0.51) both may, and both extra rows /
// planes are read: 8 + 8 + 8 words for five taps, still half of what five separate taps take.
template <bool NARROW>
__device__ __forceinline__ void sample_normal_taps_shared_lds(const float *box, const TapCoords &t, const CentreLerps &c,
                                                              float &dx, float &dy, float &dz) {
    // every address is the centre's plus a per-lane constant: the taps sit in the centre's cell or in the one next to it
    // (delta < 1 voxel), so a select + an add replaces the multiply-add chain per tap
    const int base = t.lx * BOX_SX + t.ly * BOX_SY + t.lz;
    {   // x: two whole taps (their first lerp has its own fraction: nothing of the centre's is reusable but the voxels)
        const int ip = base + ((t.lxp != t.lx) ? BOX_SX : 0), im = base - ((t.lxm != t.lx) ? BOX_SX : 0);
        dx = tri_lds(box, ip, t.fxp, t.fy, t.fz) - tri_lds(box, im, t.fxm, t.fy, t.fz);
    }
    {   // y: the x-lerps of row ly+2 (for a +delta tap in the cell above) and of row ly-1 (a -delta tap in the cell below)
        const bool up = t.lyp != t.ly, dn = t.lym != t.ly;
        const int ru = base + (NARROW ? (up ? 2 * BOX_SY : -BOX_SY) : 2 * BOX_SY);
        const float u0 = mixf(box[ru], box[ru + BOX_SX], t.fx);
        const float u1 = mixf(box[ru + 1], box[ru + BOX_SX + 1], t.fx);
        float d0 = u0, d1 = u1;
        if (!NARROW) {
            const int rd = base - BOX_SY;
            d0 = mixf(box[rd], box[rd + BOX_SX], t.fx);
            d1 = mixf(box[rd + 1], box[rd + BOX_SX + 1], t.fx);
        }
        const float p = mixf(mixf(up ? c.b0 : c.a0, up ? u0 : c.b0, t.fyp), mixf(up ? c.b1 : c.a1, up ? u1 : c.b1, t.fyp), t.fz);
        const float m = mixf(mixf(dn ? d0 : c.a0, dn ? c.a0 : c.b0, t.fym), mixf(dn ? d1 : c.a1, dn ? c.a1 : c.b1, t.fym), t.fz);
        dy = p - m;
    }
    {   // z: the bilinear planes lz+2 and lz-1
        const bool up = t.lzp != t.lz, dn = t.lzm != t.lz;
        const int pu = base + (NARROW ? (up ? 2 : -1) : 2);
        const float zu = mixf(mixf(box[pu], box[pu + BOX_SX], t.fx), mixf(box[pu + BOX_SY], box[pu + BOX_SX + BOX_SY], t.fx), t.fy);
        float zd = zu;
        if (!NARROW) {
            const int pd = base - 1;
            zd = mixf(mixf(box[pd], box[pd + BOX_SX], t.fx), mixf(box[pd + BOX_SY], box[pd + BOX_SX + BOX_SY], t.fx), t.fy);
        }
        const float p = mixf(up ? c.zh : c.zl, up ? zu : c.zh, t.fzp);
        const float m = mixf(dn ? zd : c.zl, dn ? c.zl : c.zh, t.fzm);
        dz = p - m;
    }
}

// Tap coordinates of a sample at (sm.px, sm.py, sm.pz), in two halves: the centre tap (returns whether the sample's cell
// lies in this brick), and the six normal taps -- needed only where the sample is lit (forward) or always (backward).
template <typename VT>
__device__ __forceinline__ bool sample_centre_coords_at(const VolView<VT> &vol, const BrickCtx &c, const Sample &sm, TapCoords &t) {
    int x0, y0, z0;
    axis_coord(sm.px, vol.scx, x0, t.fx);
    axis_coord(sm.py, vol.scy, y0, t.fy);
    axis_coord(sm.pz, vol.scz, z0, t.fz);
    t.lx = x0 - c.ox; t.ly = y0 - c.oy; t.lz = z0 - c.oz;
    // the brick's cells are box elements 1 .. BRK along each axis (element 0 is the voxel below the brick)
    return (unsigned)(t.lx - 1) < (unsigned)BRK && (unsigned)(t.ly - 1) < (unsigned)BRK && (unsigned)(t.lz - 1) < (unsigned)BRK;
}
template <typename VT>
__device__ __forceinline__ void sample_normal_coords_at(const VolView<VT> &vol, const BrickCtx &c, const Sample &sm, TapCoords &t) {
    const float delta = 1e-3f;
    int k;
    axis_coord(sm.px + delta, vol.scx, k, t.fxp); t.lxp = k - c.ox;
    axis_coord(sm.px - delta, vol.scx, k, t.fxm); t.lxm = k - c.ox;
    axis_coord(sm.py + delta, vol.scy, k, t.fyp); t.lyp = k - c.oy;
    axis_coord(sm.py - delta, vol.scy, k, t.fym); t.lym = k - c.oy;
    axis_coord(sm.pz + delta, vol.scz, k, t.fzp); t.lzp = k - c.oz;
    axis_coord(sm.pz - delta, vol.scz, k, t.fzm); t.lzm = k - c.oz;
}
template <typename VT>
__device__ __forceinline__ bool sample_coords_at(const VolView<VT> &vol, const BrickCtx &c, Sample &sm, TapCoords &t) {
    if (!sample_centre_coords_at(vol, c, sm, t)) return false;
    sample_normal_coords_at(vol, c, sm, t);
    return true;
}
__device__ __forceinline__ void load_ray(const float *entry, const float *exit_, const float *rays, const int32_t *nsamp,
                                         size_t p, RayGeom &rg) {
    rg.n = nsamp[p]; rg.entry = entry[p]; rg.exit_ = exit_[p];
    rg.vx = rays[3 * p]; rg.vy = rays[3 * p + 1]; rg.vz = rays[3 * p + 2];
    rg.t0 = rg.entry + 0.5f * (rg.exit_ - rg.entry) / (float)rg.n;
}

// ---- LDS gradient accumulators --------------------------------------------------------------------------
// ds_add_f32 is serialised on gfx950 (193 cycles per wave-instruction whatever the addresses), ds_add_u64 takes 9-12 and
// ds_add_f64 16-20, twice that per conflicting address (tools/microbench/lds_atomic_bench,
// profiles/r02_microbench_lds_atomics.txt). Every accumulator is one 64-bit word, used in one of two formats:
//   FIXED   64-bit fixed point with one scale PER BRICK (2^40 / the largest finite |grad_out| among the brick's candidate
//           pixels) and block-floating-point addends per wave (below). The fast format (d_volume at 512^3: 6.6 ms against
//           7.5 ms with doubles); a contribution below 2^-41 of the brick's largest upstream gradient is lost.
//   DOUBLE  the f32 contribution is widened (v_cvt_f64_f32: the cost of the float -> int conversion) and added with
//           ds_add_f64: no scale, no dynamic-range limit.
// d_tf always accumulates in DOUBLE (few conflicts after the run sums: it is the faster format there). d_volume uses
// FIXED unless the brick's candidate pixels span more than 2^DR_MIXED_BITS in |grad_out| (smallest non-zero against
// largest): then a voxel touched only by the small-gradient rays would lose precision, and the brick switches to
// DOUBLE (brick-uniform branch; stats[ST_F64_BRICKS] counts them). A stray 1e30 or an infinity in grad_out is just such a
// range: its brick goes to DOUBLE, where its ray's contributions are clamped and every other ray's stay exact.
#ifndef DR_MIXED_BITS
#define DR_MIXED_BITS 6
#endif
constexpr float ACC_LIM = 1.0e30f;  // DOUBLE: adjoints beyond this (an overflowed loss) are clamped, NaN adjoints dropped
__device__ __forceinline__ float acc_sanitise(float x) { return (x == x) ? fminf(fmaxf(x, -ACC_LIM), ACC_LIM) : 0.0f; }
__device__ __forceinline__ void acc_add_f64(unsigned long long *p, float x) {
    atomicAdd(reinterpret_cast<double *>(p), (double)x);  // ds_add_f64
}
// a workgroup's double d_tf sum (bit pattern, as acc_add_f64 leaves it in LDS) into the call's table in the workspace
__device__ __forceinline__ void dtf64_add(double *table, int view, int k, unsigned long long raw) {
    unsafeAtomicAdd(table + (size_t)view * 4 * DTF64_R + k, __longlong_as_double((long long)raw));   // global_atomic_add_f64
}
__device__ __forceinline__ float acc_f64_to_float(unsigned long long v) {
    return fminf(fmaxf((float)__longlong_as_double((long long)v), -3.0e38f), 3.0e38f);
}

// FIXED format --------------------------------------------------------------------------------------------
// value = x * 2^shift, stored as a two's-complement int64; shift = DR_FIX_BITS - exponent(gmax), gmax = the largest
// |grad_out| component among the brick's candidate pixels (every brick has its own scale). The unit 2^-shift is 2^-40 of
// gmax; sums stay inside 63 bits as long as no addend exceeds lim = 2^12 * gmax (2^10 such addends fit) -- the rare
// sample beyond that goes straight to global memory (B1). Addends are BLOCK FLOATING POINT per wave: the 64 samples of
// a pass share the exponent of the largest adjoint among them, each addend is rounded to nearest to 31 bits below that
// (v_cvt_rpi_i32_f32) and shifted up into the 64-bit word -- three VALU instructions per add, a resolution of 2^-31 of
// the wave's own largest adjoint (f32 atomics: 2^-24 of the running SUM), no bias, deterministic.
#ifndef DR_FIX_BITS
#define DR_FIX_BITS 40
#endif
#define DR_FIX_LIM_BITS 12
struct FixScale {
    int shift;   // value = x * 2^shift
    float lim;   // adjoints beyond +-lim = 2^12 * gmax do not go through the box
    double inv;  // 2^-shift
    int sh;      // wave-uniform, set per pass (fix_wave_scale): addend = cvt(x * 2^(shift - sh)) << sh, 1 <= sh <= 24
    int pw;      // 2^sh
};
__device__ __forceinline__ FixScale make_fix_scale(float gmax) {
    if (!(gmax > 0.0f) || !(gmax < 3.0e38f)) gmax = 1.0f;  // all-zero upstream gradient
    int e;
    frexpf(gmax, &e);            // gmax < 2^e
    e = e < -80 ? -80 : e;       // (a unit of 2^-120 is below anything an f32 adjoint resolves; keeps 2^(shift - sh) finite)
    FixScale f;
    f.shift = DR_FIX_BITS - e;   // gmax * 2^shift < 2^DR_FIX_BITS
    f.lim = ldexpf(1.0f, e + DR_FIX_LIM_BITS);
    f.inv = ldexp(1.0, -f.shift);
    f.sh = 1; f.pw = 2;
    return f;
}
// A NaN adjoint (NaN pixel in grad_out, NaN voxel) contributes nothing: the reference lets it poison every voxel and
// texel the ray touches and then zeroes those with nan_to_num (VR.py:463-475); here only the bad ray is dropped.
__device__ __forceinline__ float fix_clamp(float x, const FixScale &f) {
    return (x == x) ? fminf(fmaxf(x, -f.lim), f.lim) : 0.0f;
}
// round to nearest (ties up): one instruction, like the truncating conversion, but without its bias toward zero
__device__ __forceinline__ int cvt_rn_i32(float x) {
    int q;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(q) : "v"(x));
    return q;
}
// The pass's scale from the largest adjoint magnitude of the wave (bmax >= every |addend| of the pass, wave-uniform, finite,
// <= 4 lim): returns the factor 2^(shift - sh) the adjoints are multiplied with once, and sets f.sh.
__device__ __forceinline__ float fix_wave_scale(float bmax, FixScale &f) {
    // bmax < 2^eb; addends < 2^30 after scaling by 2^(30 - eb): sh = shift - (30 - eb), kept in [1, 31]
    const int eb = (bmax > 0.0f) ? (int)((__float_as_uint(bmax) >> 23) & 0xff) - 126 : -126;  // (a denormal bmax counts as 2^-126)
    int sh = f.shift - 30 + eb;
    static_assert(DR_FIX_BITS + DR_FIX_LIM_BITS + 2 - 30 <= 30, "2^sh must fit a positive int32 (fix_add_scaled)");
    sh = sh < 1 ? 1 : (sh > 30 ? 30 : sh);  // sh > 24 cannot happen below 4 lim (DR_FIX_BITS + DR_FIX_LIM_BITS + 2 - 30)
    f.sh = sh; f.pw = 1 << sh;
    return __uint_as_float((unsigned int)(127 + f.shift - sh) << 23);  // 2^(shift - sh): the exponent stays within [-119, 119]
}
// x already carries the factor 2^(shift - sh) and |x| < 2^31
__device__ __forceinline__ void fix_add_scaled(unsigned long long *p, float x, const FixScale &f) {
    const int q = cvt_rn_i32(x);
    // q << sh as a 64-bit integer: ONE v_mad_i64_i32 with the wave-uniform 2^sh (4.4 issue cycles) instead of a 32-bit left
    // shift + an arithmetic right shift for the sign-extended high word (4.5 + 2.6)
    // (spelled in assembly: from C the compiler turns the multiplication by 2^sh back into a sign extension + a 64-bit shift)
    unsigned long long v, carry_unused;
    asm("v_mad_i64_i32 %0, %1, %2, %3, 0" : "=v"(v), "=s"(carry_unused) : "v"(q), "s"(f.pw));
    atomicAdd(p, v);  // ds_add_u64
}
__device__ __forceinline__ float fix_to_float(unsigned long long v, const FixScale &f) {
    return (float)((double)(long long)v * f.inv);
}

// Zeroes the segment counts between the alpha pre-pass and the forward march -- only if the pre-pass ran at all (some
// view's TF can terminate rays); otherwise nothing was written since the forward's first memset.
static __global__ __launch_bounds__(256) void clear_counts_if_prepass_kernel(uint4 *cnt, size_t n16, const unsigned int *vflags, int n_views) {
    bool any = false;
    for (int v = 0; v < n_views; ++v) any = any || vflags[v] != 0u;  // uniform
    if (!any) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) cnt[i] = make_uint4(0u, 0u, 0u, 0u);
}

// ------------------------------------------------------------------------------------------------ host
struct Workspace {
    float4 *seg_rgba; uint16_t *seg_cnt; int32_t *ws_steps; uint8_t *rayflag; unsigned int *stats, *vflags, *n_items;
    float4 *fin; unsigned int *exact_list; uint16_t *seg_tiny; float2 *tape; double *dtf64;
    unsigned long long *unlit; size_t unlit_bytes; int lm_words;   // right behind seg_cnt: one memset clears the counts and the masks
    BrickCtxRec *ctx;
    BrickItem *items;
    size_t cnt_bytes;
};
// tape_stride > 0: the DR_TAPE_TF tape follows the ordinary workspace
static inline size_t ws_layout(void *base, int n_views, int NP, const BrickGrid &g, Workspace *w, int tape_stride = 0) {
    size_t o = 0;
    const int NL = g.NL;
    const size_t nseg = (size_t)n_views * NL * NP;
    unsigned char *b = static_cast<unsigned char *>(base);
    if (w) w->stats = reinterpret_cast<unsigned int *>(b + o);
    o += ST_WORDS * 4;
    if (w) w->vflags = reinterpret_cast<unsigned int *>(b + o);
    o += align16((size_t)n_views * 8);
    if (w) w->seg_rgba = reinterpret_cast<float4 *>(b + o);
    o += nseg * 16;
    // 16 bytes in front of seg_cnt hold the count of overflow work items: the forward's first memset zeroes both
    if (w) w->n_items = reinterpret_cast<unsigned int *>(b + o);
    o += 16;
    if (w) { w->seg_cnt = reinterpret_cast<uint16_t *>(b + o); w->cnt_bytes = nseg * 2; }
    o += align16(nseg * 2);
    {   // "all unlit" masks of the non-differentiable render: one bit per (ray, layer), up to 128 layers (volumes up to ~770^3)
        const int lm = NL <= 64 ? 1 : (NL <= 128 ? 2 : 0);
        if (w) { w->unlit = reinterpret_cast<unsigned long long *>(b + o); w->lm_words = lm; w->unlit_bytes = (size_t)n_views * NP * 8 * lm; }
        o += align16((size_t)n_views * NP * 8 * lm);
    }
    if (w) w->ws_steps = reinterpret_cast<int32_t *>(b + o);
    o += align16((size_t)n_views * NP * 4);
    if (w) w->rayflag = reinterpret_cast<uint8_t *>(b + o);
    o += align16((size_t)n_views * NP);
    if (w) w->ctx = reinterpret_cast<BrickCtxRec *>(b + o);
    o += (size_t)n_views * g.NBx * g.NBy * g.NBz * sizeof(BrickCtxRec);
    if (w) w->items = reinterpret_cast<BrickItem *>(b + o);
    o += (size_t)ITEM_CAP * sizeof(BrickItem);
    if (w) w->fin = reinterpret_cast<float4 *>(b + o);
    o += (size_t)n_views * NP * 16;
    if (w) w->exact_list = reinterpret_cast<unsigned int *>(b + o);
    o += align16((size_t)n_views * NP * 4);
    if (w) w->seg_tiny = reinterpret_cast<uint16_t *>(b + o);
    o += align16(nseg * 2);
    if (w) w->dtf64 = reinterpret_cast<double *>(b + o);
    o += (size_t)n_views * DTF64_R * 4 * 8;
    o = (o + 255) & ~(size_t)255;
    if (w) w->tape = tape_stride > 0 ? reinterpret_cast<float2 *>(b + o) : nullptr;
    o += (size_t)n_views * NP * (size_t)tape_stride * 8;
    return o;
}



template <typename VT>
static inline BrickParams<VT> make_brick_params(const MarchArgs &a, const Workspace &w) {
    BrickParams<VT> P;
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
    P.imgW = a.img_W > 0 ? a.img_W : a.W; P.row0 = a.img_W > 0 ? a.row0 : 0;
    const double near_h = 2.0 * tan(a.fov_rad) * a.near_plane;
    P.near_ = (float)a.near_plane; P.near_h = (float)near_h; P.near_w = (float)(near_h * ((double)P.imgW / (double)a.H));
    P.g = make_brick_grid(a.VX, a.VY, a.VZ);

    P.seg_rgba = w.seg_rgba; P.seg_cnt = w.seg_cnt; P.rayflag = w.rayflag; P.stats = w.stats; P.vflags = w.vflags; P.n_views = a.n_views; P.ws_steps = w.ws_steps; P.fin = w.fin; P.exact_list = w.exact_list; P.seg_tiny = w.seg_tiny; P.dtf64 = w.dtf64; P.tape = nullptr; P.tape_stride = 0; P.unlit = w.unlit; P.lm_words = 0;
    P.hint_noterm = (a.hints & DR_HINT_NO_EARLY_TERMINATION) ? 1 : 0;
    P.count_eval = (a.hints & DR_COUNT_EVALUATED) ? 1 : 0;
    P.nondiff = a.mode == DR_MODE_NONDIFF ? 1 : 0;
    P.use_live = a.use_live; P.ctx = w.ctx; P.items = w.items; P.n_items = w.n_items;
    P.mark = ws_fingerprint(a);
    P.pp_l0 = a.pp_l0; P.pp_l1 = a.pp_l1; P.pp_first = a.pp_first;
    P.out = a.out; P.steps = a.steps;
    P.grad_out = a.grad_out; P.out_fwd = a.out_fwd;
    P.dvol.p = a.d_vol; P.dvol.sx = a.dsx; P.dvol.sy = a.dsy; P.dvol.sz = a.dsz; P.dvol_vs = a.dvol_vs;
    P.d_tf = a.d_tf; P.dtf_vs = a.dtf_vs / 4;
    return P;
}

// Dynamic LDS above 64 KB needs an explicit opt-in per kernel (and device); asked for once, not on every launch.
hipError_t allow_lds_impl(const void *kernel, size_t bytes);
template <typename K>
static inline hipError_t allow_lds(K kernel, size_t bytes) {
    if (bytes <= 64 * 1024) return hipSuccess;
    return allow_lds_impl(reinterpret_cast<const void *>(kernel), bytes);
}


}  // namespace dr
