// march_flat.hip -- brick-centric march with ONE LANE PER SAMPLE (DR_VARIANT_AUTO fast path, gfx950).
//
// Why bricks: one marched sample needs 56 voxel fetches (7 trilinear taps, VR.py:153-203). Served as global gathers
// they cost >= 25 cycles per wave-load on a CU (measured, tools/microbench); from LDS they cost 2-6. So the volume is
// processed brick by brick: a workgroup stages one 12^3-cell brick (+apron: 15^3 voxels) in LDS with coalesced reads
// and marches every ray segment that crosses it, leaving one partial composite per (ray, layer) for the per-ray
// composition (F2, ray_passes.hip); the backward accumulates d_volume / d_tf in fixed-point LDS boxes and flushes
// each once per brick. The work items of a brick are the SAMPLES of all ray segments that cross it, laid out back
// to back ("flat" index): a wave takes 64 consecutive lanes' worth.
//   * every lane is busy whatever the lengths of the individual segments (a brick holds only ~170 segments
//     but ~7800 samples);
//   * the 64 lanes of a load walk along one or two rays: few distinct cells per instruction on odd LDS
//     strides, i.e. mostly conflict-free reads, and neighbouring lanes that share a cell are LDS broadcasts;
//   * front-to-back compositing inside a segment becomes a segmented wave scan of the associative "over"
//     operator (DPP, no LDS traffic); the running composite of a segment that spans several chunks is carried
//     in registers by the wave that owns it. The forward composites K (2 or 4) consecutive samples per lane in
//     registers first, so scan and bookkeeping are paid once per K*64 samples.
// Tuned against the measured VALU cost table of gfx950 (profiles/r01_microbench_oprate.txt): only f32
// add/mul/fma, v_mov and v_add_u32 issue in 2 cycles, everything else in 4 (rcp/rsq 8).
// Reference functions replaced: VR.py:261-372 and the autodiff twins VR.py:460-461,470-471.
#include "dr_brick_common.h"
#include "dr_wave.h"
#include "dr_tuning.h"

namespace dr {

// Workgroup shapes, samples per lane and launch grids: dr_tuning.h (every value is the measured optimum, each with its A/B row).
// Workgroup configuration of a brick kernel: forward / alpha pre-pass / backward with a gradient box / backward w.r.t. the TF only.
// ALPHA: 0 = not the pre-pass; 1 = the pre-pass in the forward's shape (256-entry tables, five workgroups per CU); 2 = the pre-pass
// of HIGH sampling rates (>= 3): 192-entry tables, an alpha-only TF table (4 B per entry) and 80 VGPRs put SIX workgroups on a CU.
template <bool BWD, bool WANT_VOL, int ALPHA = 0>
struct FlatCfg {
    static constexpr int FNT = BWD ? (WANT_VOL ? DR_FNT_BWD : DR_FNT_BWDTF) : DR_FNT_FWD;      // threads per workgroup
    static constexpr int EC = BWD ? (WANT_VOL ? DR_FEC_BWD : DR_FEC_BWDTF) : (ALPHA == 2 ? DR_FEC_ALPHA_HI : DR_FEC_FWD);   // segment-table entries (= candidates per round)
    static constexpr int WAVES = BWD ? (WANT_VOL ? DR_BWD_WAVES : DR_BWDTF_WAVES) : (ALPHA == 2 ? DR_ALPHA_WAVES_HI : DR_FWD_WAVES);  // waves per SIMD the registers must allow
    static constexpr int UNEVEN = (BWD && WANT_VOL) ? DR_BWD_UNEVEN : 0;                      // uneven dealing (cand_load)
    static constexpr int FNW = FNT / 64, CW = EC / FNW;                                       // waves; candidates per wave and round
    static_assert(CW <= 64 && CW * FNW == EC, "one candidate per lane and round");
};

constexpr int VALID_LIT = 1 << 30;   // FlatLds::valid, alpha pre-pass of a non-differentiable render: a piece of the segment held a lit sample
// FlatLds::valid packs two counters of a segment: bits 0-15 its in-brick samples, bits 16-29 those among them whose opacity is TINY
// (0 < op < DR_D4_TINY_OP: dr_brick_common.h, "Sequential float32 compositing") -- one LDS atomic per piece adds both
constexpr int VALID_CNT_MASK = 0xffff, VALID_TINY_SHIFT = 16, VALID_TINY_MASK = 0x3fff;
struct FlatLds {
    float4 *tf; float *box; unsigned long long *dbox; unsigned long long *dtf;
    float4 *ray0;   // (t0, exit, (float)(n-1), RN(1/(n-1)))
    float4 *ray1;   // (vx, vy, vz, bits(pixel index))
    int *segi;      // workspace slot of the segment within its view: ray's layer of this brick * NP + pixel
    int *s_rel;     // first sample index of the segment minus its flat offset
    int *offs;      // per wave: flat index of each segment's first sample, then the wave's total (index entry + wave)
    int *valid;     // in-brick samples of each segment (forward)
    int *slen;      // true length of the segment (its flat extent is padded to a multiple of the samples per lane)
    int *live;      // backward: live sample count of the ray
    float *gmax;    // backward: per wave, the largest |grad_out| among the brick's candidate pixels
    float *tfa;     // alpha pre-pass: the TF table holds the R alphas only (4 B per entry instead of 16)
    float *mm;      // forward: per wave, smallest / largest staged voxel and a NaN flag (brick_empty_test)
};
// LDS layout: everything of compile-time size first (so every address below is an immediate), then the two
// tables whose size depends on the run-time TF resolution R.
template <bool BWD>
__host__ __device__ constexpr size_t flat_fixed_bytes(bool want_vol, int alpha = 0) {
    const int EC = want_vol ? FlatCfg<BWD, true>::EC : (alpha == 2 ? FlatCfg<BWD, false, 2>::EC : FlatCfg<BWD, false>::EC);
    size_t s = ((size_t)BOX_LDS * 4 + 15) / 16 * 16;
    if (BWD && want_vol) s += ((size_t)BOX_LDS * 8 + 15) / 16 * 16;
    s += (size_t)EC * 32;
    s += (size_t)EC * 4 + (((size_t)EC + 8) * 4 + 15) / 16 * 16 + (size_t)EC * 4 + (size_t)EC * 4;  // s_rel, offs, valid, slen
    s += (size_t)EC * 4;  // segi
    if (BWD) s += (size_t)EC * 4 + 64;  // live; gmax, gmin per wave
    else s += 48;                       // mm: min, max, NaN flag per wave (brick_empty_test)
    return s;
}
template <bool BWD>
__host__ __device__ inline size_t flat_lds_bytes(int R, bool want_vol, bool want_tf, int alpha = 0) {
    return flat_fixed_bytes<BWD>(want_vol, alpha) + (alpha ? align16((size_t)R * 4) : (size_t)R * 16) + ((BWD && want_tf) ? (size_t)R * 32 : 0);
}
template <bool BWD, bool WANT_VOL, bool WANT_TF, int ALPHA = 0>
__device__ __forceinline__ FlatLds flat_carve(unsigned char *smem, int R) {
    constexpr int EC = FlatCfg<BWD, WANT_VOL, ALPHA>::EC;
    FlatLds L;
    size_t o = 0;
    L.box = reinterpret_cast<float *>(smem + o); o += align16(BOX_LDS * 4);
    L.dbox = nullptr; L.dtf = nullptr; L.live = nullptr;
    if (BWD && WANT_VOL) { L.dbox = reinterpret_cast<unsigned long long *>(smem + o); o += align16(BOX_LDS * 8); }
    L.ray0 = reinterpret_cast<float4 *>(smem + o); o += (size_t)EC * 16;
    L.ray1 = reinterpret_cast<float4 *>(smem + o); o += (size_t)EC * 16;
    L.s_rel = reinterpret_cast<int *>(smem + o); o += (size_t)EC * 4;
    L.segi = reinterpret_cast<int *>(smem + o); o += (size_t)EC * 4;
    L.offs = reinterpret_cast<int *>(smem + o); o += align16((EC + 8) * 4);  // per wave: its entries' offsets + end marker
    L.valid = reinterpret_cast<int *>(smem + o); o += (size_t)EC * 4;
    L.gmax = nullptr;
    L.slen = reinterpret_cast<int *>(smem + o); o += (size_t)EC * 4;
    L.mm = nullptr;
    if (BWD) { L.live = reinterpret_cast<int *>(smem + o); o += (size_t)EC * 4; L.gmax = reinterpret_cast<float *>(smem + o); o += 64; }
    else { L.mm = reinterpret_cast<float *>(smem + o); o += 48; }
    L.tf = reinterpret_cast<float4 *>(smem + o); o += ALPHA ? align16((size_t)R * 4) : (size_t)R * 16;   // (the pre-pass: R alphas, read as L.tfa)
    L.tfa = reinterpret_cast<float *>(L.tf);
    if (BWD && WANT_TF) L.dtf = reinterpret_cast<unsigned long long *>(smem + o);
    return L;
}

// Walk of the brick's BOX^3 voxel box by a workgroup: 16 lanes per row of BOX (= 15) elements along the axis
// with the smallest global stride ("a"), FNT/16 rows per pass; rows are numbered r = b + BOX*d over the other two
// axes. All index arithmetic is 32-bit: the brick origin is folded into a uniform base pointer, in-box offsets
// stay below 2^31 (strides are checked on the host, flat_strides_ok).
struct BoxWalk {
    int la, lb, ld;        // LDS strides of (a, b, d)
    int ga, gb, gd;        // global strides of (a, b, d), elements
    int oa, ob, od;        // global index of box element 0 along (a, b, d)
    int Va, Vb, Vd;        // volume extents along (a, b, d)
    long long base;        // element offset of box element (0,0,0) from the volume pointer (may be negative)
};
__device__ __forceinline__ BoxWalk make_box_walk(long long sx, long long sy, long long sz, int VX, int VY, int VZ,
                                                 const BrickCtx &c) {
    const int fast = (sx <= sy && sx <= sz) ? 0 : ((sy <= sz) ? 1 : 2);
    BoxWalk w;
    if (fast == 0) { w.la = BOX_SX; w.lb = BOX_SY; w.ld = 1; w.ga = (int)sx; w.gb = (int)sy; w.gd = (int)sz;
                     w.oa = c.ox; w.ob = c.oy; w.od = c.oz; w.Va = VX; w.Vb = VY; w.Vd = VZ; }
    else if (fast == 1) { w.la = BOX_SY; w.lb = BOX_SX; w.ld = 1; w.ga = (int)sy; w.gb = (int)sx; w.gd = (int)sz;
                          w.oa = c.oy; w.ob = c.ox; w.od = c.oz; w.Va = VY; w.Vb = VX; w.Vd = VZ; }
    else { w.la = 1; w.lb = BOX_SY; w.ld = BOX_SX; w.ga = (int)sz; w.gb = (int)sy; w.gd = (int)sx;
           w.oa = c.oz; w.ob = c.oy; w.od = c.ox; w.Va = VZ; w.Vb = VY; w.Vd = VX; }
    w.base = (long long)c.ox * sx + (long long)c.oy * sy + (long long)c.oz * sz;
    return w;
}
// row r (0 .. BOX*BOX-1) -> (b, d): r / BOX == (r * (65536 / BOX + 1)) >> 16 for r < 4000 (checked exhaustively for
// BOX = 9, 11, 13, 15, 19, 23)
__device__ __forceinline__ void box_row(int r, int &b, int &d) {
    static_assert(BOX == 9 || BOX == 11 || BOX == 13 || BOX == 15 || BOX == 19 || BOX == 23, "magic divisor checked for these box edges only");
    d = (r * (65536 / BOX + 1)) >> 16; b = r - d * BOX;
}

// Staging of the brick's voxel box (and the TF) in LDS, in two halves: box_issue() puts all global loads in flight
// (one register per pass), box_commit() stores them to LDS. The caller lists the brick's ray segments in between,
// which needs neither: the memory latency of the box is covered by that work. (16-byte loads of 4 voxels per lane
// were tried instead of 15 x 4-byte passes: no gain in the forward, -2.5 % in the backward.)
template <int FNT>
struct BoxStage {
    static constexpr int RP = FNT / 16, NPASS = (BOX * BOX + RP - 1) / RP;
    float bv[NPASS];
    float4 tfv;
};
template <typename VT, int FNT, bool ALPHA_TF = false>
__device__ __forceinline__ void box_issue(const BrickParams<VT> &P, const VolView<VT> &vol, const BrickCtx &c,
                                          const float4 *tfg, BoxStage<FNT> &st) {
    using S = BoxStage<FNT>;
    const BoxWalk w = make_box_walk(vol.sx, vol.sy, vol.sz, vol.VX, vol.VY, vol.VZ, c);
    const VT *base = vol.p + w.base;
    const int a = threadIdx.x & 15, row = threadIdx.x >> 4;
    const bool a_ok = a < BOX && (unsigned)(w.oa + a) < (unsigned)w.Va;
    const int a_off = a * w.ga;
#pragma unroll
    for (int k = 0; k < S::NPASS; ++k) {
        const int r = row + k * S::RP;
        int b, d;
        box_row(r, b, d);
        st.bv[k] = 0.0f;
        if (a_ok && r < BOX * BOX && (unsigned)(w.ob + b) < (unsigned)w.Vb && (unsigned)(w.od + d) < (unsigned)w.Vd)
            st.bv[k] = ld_voxel(base + (a_off + b * w.gb + d * w.gd));
    }
    st.tfv = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.R <= FNT) {   // uniform: the usual one texel per thread
        if ((int)threadIdx.x < P.R) { if (ALPHA_TF) st.tfv.w = tfg[threadIdx.x].w; else st.tfv = tfg[threadIdx.x]; }
    }
}
template <typename VT, int FNT, bool ALPHA_TF = false>
__device__ __forceinline__ void box_commit(const BrickParams<VT> &P, const VolView<VT> &vol, const BrickCtx &c,
                                           const float4 *tfg, const BoxStage<FNT> &st, FlatLds &L) {
    using S = BoxStage<FNT>;
    const BoxWalk w = make_box_walk(vol.sx, vol.sy, vol.sz, vol.VX, vol.VY, vol.VZ, c);
    const int a = threadIdx.x & 15, row = threadIdx.x >> 4;
    const int a_lds = a * w.la;
    if (a < BOX) {
#pragma unroll
        for (int k = 0; k < S::NPASS; ++k) {
            const int r = row + k * S::RP;
            int b, d;
            box_row(r, b, d);
            if (r < BOX * BOX) L.box[a_lds + b * w.lb + d * w.ld] = st.bv[k];
        }
    }
    if (ALPHA_TF) {
        if (P.R <= FNT) { if ((int)threadIdx.x < P.R) L.tfa[threadIdx.x] = st.tfv.w; }
        else { for (int k = threadIdx.x; k < P.R; k += FNT) L.tfa[k] = tfg[k].w; }
    } else {
        if (P.R <= FNT) { if ((int)threadIdx.x < P.R) L.tf[threadIdx.x] = st.tfv; }
        else { for (int k = threadIdx.x; k < P.R; k += FNT) L.tf[k] = tfg[k]; }
    }
}

// ---- Empty bricks (round 5; see brick_probe_empty in dr_brick_common.h) -------------------------------------------------------
// Part 1, with the staged values still in registers: every wave leaves the smallest and the largest voxel it staged (padding
// outside the volume is staged as 0 and counts: conservative) and whether it saw a NaN. Before the workgroup's barrier.
template <int FNT>
__device__ __forceinline__ void brick_empty_publish(const BoxStage<FNT> &st, FlatLds &L) {
    using S = BoxStage<FNT>;
    const int a = threadIdx.x & 15, row = threadIdx.x >> 4;
    float mn = 3.0e38f, mx = -3.0e38f;
    bool nanv = false;
    if (a < BOX) {
#pragma unroll
        for (int k = 0; k < S::NPASS; ++k) {
            if (row + k * S::RP < BOX * BOX) { const float v = st.bv[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); nanv = nanv || (v != v); }
        }
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o)); mx = fmaxf(mx, __shfl_xor(mx, o)); }
    const bool wnan = __any(nanv);
    if ((threadIdx.x & 63) == 0) { const int w = threadIdx.x >> 6; L.mm[w] = mn; L.mm[4 + w] = mx; L.mm[8 + w] = wnan ? 1.0f : 0.0f; }
}
// Part 2, after the barrier (box and TF are in LDS): every intensity a sample of this brick can take lies between the smallest and
// the largest staged voxel (trilinear taps are convex combinations; a few ulps of rounding are absorbed by the texel of slack on
// either side), so it indexes TF texels [lo, hi]; if none of those composites, no sample of the brick does. One more barrier
// (only for bricks whose nine probes said "maybe"). Workgroup-uniform result.
template <typename VT, int FNT, bool ALPHA_TF>
__device__ __forceinline__ bool brick_empty_decide(const BrickParams<VT> &P, const FlatLds &L) {
    static_assert(FNT / 64 <= 4, "mm holds four waves");
    float mn = 3.0e38f, mx = -3.0e38f, nn = 0.0f;
#pragma unroll
    for (int w = 0; w < FNT / 64; ++w) { mn = fminf(mn, L.mm[w]); mx = fmaxf(mx, L.mm[4 + w]); nn += L.mm[8 + w]; }
    // (a negative intensity indexes texel 0, an infinite one the last texel -- as low_high_frac and the clamp of tf_lookup_from_I do)
    const int lo = min(max((int)fminf(fmaxf(mn, 0.0f) * P.tf_len, (float)P.R) - 1, 0), P.R - 1);
    const int hi = min((int)fminf(fmaxf(mx, 0.0f) * P.tf_len, (float)P.R) + 2, P.R - 1);
    bool bad = nn != 0.0f;   // a NaN voxel: leave the brick to the ordinary path
    for (int k = lo + (int)threadIdx.x; k <= hi; k += FNT) bad = bad || texel_composites(ALPHA_TF ? L.tfa[k] : L.tf[k].w, P.nondiff);
    return !__syncthreads_or(bad);
}
// Part 3: the brick is empty -- the wave's segments need their exact in-brick sample counts and nothing else. One lane per
// segment: the in-brick samples of a ray are one run (every coordinate is monotone along a line) inside the listed conservative
// range [s0, s1), so the run's ends are found by testing a few samples at either end of the range with the sample loop's own
// position and cell arithmetic. A lane that does not find both ends within four probes reports failure, and its wave marches
// the round the ordinary way (the box is staged). Returns the count (0: no sample of the ray in this brick).
template <typename VT>
__device__ __forceinline__ bool empty_segment_count(const VolView<VT> &vol, const BrickCtx &c, f3 cam, float4 r0, float4 r1, int s0, int s1, int &count) {
    auto inside = [&](int s) {
        float px, py, pz, fr;
        int x0, y0, z0;
        sample_pos_rcp(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, cam.x, cam.y, cam.z, s, px, py, pz);
        axis_coord(px, vol.scx, x0, fr); axis_coord(py, vol.scy, y0, fr); axis_coord(pz, vol.scz, z0, fr);
        return (unsigned)(x0 - c.ox - 1) < (unsigned)BRK && (unsigned)(y0 - c.oy - 1) < (unsigned)BRK && (unsigned)(z0 - c.oz - 1) < (unsigned)BRK;
    };
    int first = -1, last = -1;
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int s = s0 + k; if (first < 0 && s < s1 && inside(s)) first = s; }
#pragma unroll
    for (int k = 0; k < 4; ++k) { const int s = s1 - 1 - k; if (last < 0 && s >= s0 && inside(s)) last = s; }
    count = 0;
    if (first >= 0 && last >= first) { count = last - first + 1; return true; }
    return first < 0 && last < 0 && s1 - s0 <= 4;   // every listed sample was tested: none is in the brick
}

// What a thread reads from global memory for its candidate pixel of a round -- issued as ONE batch of loads (the
// ray buffers, and for the backward the coarse tape and the gradients), before any of it is looked at.
struct CandData {
    bool have;
    int pl; size_t p;
    int n; float entry, exit_, vx, vy, vz;
    int live; unsigned char rflag;
    unsigned long long um0, um1;   // the ray's "all unlit" layer masks (colour march of a non-differentiable render)
};
constexpr int MAIN_CAND = CTX_MAIN_CAND;  // candidates of a brick the main launch handles (4 listing rounds); the rest become items
constexpr int ITEM_MAX_CAND = CTX_ITEM_MAX_CAND;  // candidates per overflow item at most (its list of hits lives in LDS: 2 B each)
constexpr size_t ITEM_EXTRA_LDS = (size_t)ITEM_MAX_CAND * 2 + 16;

// Pinhole geometry of a view, for the geometric pre-test of the candidates of HEAVY bricks (more candidate pixels than the
// main launch takes: the bounding rectangle of a brick next to the eye is mostly pixels whose lines miss it): the line
// of pixel (i, j) runs along dn + u * rw + v * uh.
struct CamBasis { f3 dn, rw, uh; bool heavy; };
template <typename VT>
__device__ __forceinline__ CamBasis make_cam_basis(const BrickParams<VT> &P, f3 cam, bool heavy) {
    CamBasis b;
    b.heavy = heavy;
    if (!heavy) { b.dn = b.rw = b.uh = make_f3(0.f, 0.f, 0.f); return b; }
    const f3 vdir = normalized3(make_f3(-cam.x, -cam.y, -cam.z));
    const f3 right = normalized3(cross3b(vdir, make_f3(0.f, 1.f, 0.f)));
    const f3 up = normalized3(cross3b(right, vdir));
    b.dn = make_f3(P.near_ * vdir.x, P.near_ * vdir.y, P.near_ * vdir.z);
    b.rw = make_f3(P.near_w * right.x, P.near_w * right.y, P.near_w * right.z);
    b.uh = make_f3(P.near_h * up.x, P.near_h * up.y, P.near_h * up.z);
    return b;
}
// does the LINE cam + t * d (any t) meet the brick's (slack-widened) box?
__device__ __forceinline__ bool line_meets_brick(const BrickCtx &c, f3 cam, f3 d) {
    const float o[3] = {cam.x, cam.y, cam.z}, dd[3] = {d.x, d.y, d.z};
    float ta = -3.0e38f, tb = 3.0e38f;
    for (int k = 0; k < 3; ++k) {
        if (fabsf(dd[k]) < 1e-12f) {
            if (o[k] < c.lo[k] || o[k] > c.hi[k]) return false;
        } else {
            const float inv = __builtin_amdgcn_rcpf(dd[k]);
            const float t1 = (c.lo[k] - o[k]) * inv, t2 = (c.hi[k] - o[k]) * inv;
            ta = fmaxf(ta, fminf(t1, t2)); tb = fminf(tb, fmaxf(t1, t2));
        }
    }
    return ta <= tb;  // (the box carries BRICK_EPS of slack: far more than the rounding of the stored ray directions)
}
// `hits` (overflow items of heavy bricks): the item's candidates that passed the geometric pre-test, as offsets from c_lo;
// the rounds then run over [0, number of hits) instead of over the raw candidate range.
template <typename VT, int MODE, bool BWD, int ALPHA, bool WANT_VOL>
__device__ __forceinline__ void cand_load(const BrickParams<VT> &P, const BrickCtx &c, int view, int cbase, int ncand,
                                          const unsigned short *hits, int c_lo, int ncand_all, CandData &d, int lmw) {
    using Cfg = FlatCfg<BWD, WANT_VOL, ALPHA>;
    constexpr int FNW = Cfg::FNW, CW = Cfg::CW;  // candidates per wave and round
    const int NP = P.W * P.H;
    const int nj = c.j1 - c.j0 + 1;
    // the candidates are dealt to the waves like cards, back and forth: neighbouring pixels (similar segment lengths)
    // go to different waves, so the waves of a workgroup get nearly equal shares of the brick's samples (the
    // slowest wave has 1.03 x the mean)
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    int cc = cbase + lane_ * FNW + ((lane_ & 1) ? FNW - 1 - wave_ : wave_);  // dealt back and forth: -1.3 %
    d.have = lane_ < CW && cc < ncand;
    if constexpr (Cfg::UNEVEN != 0) {
        // The SIMDs serve a workgroup's four earlier waves before its four later ones (1.19 x the loop time for the same
        // samples): deal 8 candidates to each earlier wave for every NL (< 8) of a later one. A round = 4 groups of
        // 8 NL + 4 (8 - NL) candidates: NL back-and-forth deals of 8, then 8 - NL deals of 4 to waves 0-3 only.
        static_assert(FNW == 8 && CW == 32, "uneven dealing is laid out for 8 waves of 32 slots");
        constexpr int NL = Cfg::UNEVEN, SG = 8 * NL + 4 * (8 - NL);
        const bool early = wave_ < 4;
        const int per = early ? 8 : NL;
        const int sg = early ? (lane_ >> 3) : lane_ / NL;
        const int k = lane_ - sg * per;
        const int pos = (k < NL) ? 8 * k + ((k & 1) ? 7 - wave_ : wave_) : 8 * NL + 4 * (k - NL) + wave_;
        cc = cbase + SG * sg + pos;
        d.have = lane_ < 4 * per && cc < ncand;
    }
    d.pl = 0; d.p = 0; d.n = 0; d.entry = -1.0f; d.exit_ = 0.f; d.vx = d.vy = d.vz = 0.f;
    d.live = 0; d.rflag = 0; d.um0 = d.um1 = 0ull;
    if (!d.have) return;
    if (hits) cc = c_lo + (int)hits[cc];
    // cc / nj without the ~25-instruction integer division while the rectangle is small (always, unless the camera sits
    // inside the volume of a > 2-megapixel image): the float quotient of cc + 0.5 stays >= 0.5/nj away from an integer,
    // its error is <= 2^-22 * ni, so ni * nj < 2^21 is safe (checked exhaustively up to 1100 x 1100 with rcp +- 1 ulp)
    const int qi = (ncand_all < (1 << 21)) ? (int)(((float)cc + 0.5f) * __builtin_amdgcn_rcpf((float)nj)) : cc / nj;
    const int i = c.i0 + qi, j = c.j0 + (cc - qi * nj);
    d.pl = i * P.H + j;
    d.p = (size_t)view * NP + d.pl;
    d.n = P.nsamp[d.p];
    d.entry = P.entry[d.p];
    d.exit_ = P.exit_[d.p];
    d.vx = P.rays[3 * d.p]; d.vy = P.rays[3 * d.p + 1]; d.vz = P.rays[3 * d.p + 2];
    if (BWD || (!ALPHA && P.use_live) || (ALPHA && !P.pp_first)) d.live = P.ws_steps[d.p];
    if constexpr (!ALPHA && (!BWD || WANT_VOL)) {
        if (lmw > 0) {   // uniform; same batch of loads as the ray buffers
            const unsigned long long *um = P.unlit + ((size_t)view * lmw * NP + d.pl);
            d.um0 = um[0];
            if (lmw > 1) d.um1 = um[NP];
        }
    }
    if (BWD) d.rflag = P.rayflag[d.p];  // (the three float4 of the coarse tape / gradients are fetched per chunk: the
                                        //  backward kernel is register-bound and they would be held across the box staging)
}

// List the ray segments of this WAVE's candidates of the round and their flat offsets: every wave keeps its own
// segment table (entries [wave*CW, wave*CW + nE), offsets at index entry + wave so that each table has its end
// marker) and its own flat sample space [0, M). No workgroup barrier, no serial phase: the compaction is a ballot,
// the offsets a DPP scan. Slots follow the candidate order, so the flat sample order -- and with it every
// rounding -- is reproducible.
template <typename VT, int MODE, bool BWD, bool WANT_VOL, int ALPHA = 0, int KS = 1>
__device__ __forceinline__ void flat_build_entries(const BrickParams<VT> &P, const BrickCtx &c, f3 cam, int view,
                                                   const CandData &d, FlatLds &L, int &nE, int &M, int *live_flag, bool count_stats, int lmw) {
    constexpr int CW = FlatCfg<BWD, WANT_VOL, ALPHA>::CW;
    bool has = false;
    int s0 = 0, s1 = 0;
    float t0 = 0.f, nm1 = 0.f;
    const int pl = d.pl, live = d.live;
    const float exit_ = d.exit_;
    const f3 vd = make_f3(d.vx, d.vy, d.vz);
    if (d.have) {
        const int n = d.n;
        bool ok = ray_is_regular(n);
        int nmarch = (MODE == DR_MODE_DIFF && n > P.S) ? P.S : n;
        if (!BWD && !ALPHA && ok && P.use_live && P.vflags[view] != 0u)
            nmarch = min(nmarch, live);  // exact live count from the alpha pre-pass: dead samples are not marched
        if (ALPHA && !P.pp_first && live == -1) ok = false;  // terminated in an earlier phase of the pre-pass
        if (BWD && ok) {
            ok = !d.rflag;  // a ray the per-ray pass marched whole (repaired by the count check): B2 handles it
            nmarch = min(nmarch, live);
        }
        if (ok) {
            t0 = d.entry + 0.5f * (exit_ - d.entry) / (float)n;
            nm1 = (float)(n - 1);
            has = segment_range(c, cam, vd, t0, exit_, n, nmarch, s0, s1);
        }
    }
    // the ray's layer of this brick (dr_brick.h): distance from the brick of the ray's first sample
    int lay = 0;
    if (has) {
        int ebx, eby, ebz;
        entry_brick(P.vol.scx, P.vol.scy, P.vol.scz, cam, vd, t0, ebx, eby, ebz);
        lay = ray_layer(c.bx, c.by, c.bz, ebx, eby, ebz);
    }
    if constexpr (!ALPHA && (!BWD || WANT_VOL)) {
        // (B1 gets here only as the d_volume-ONLY backward, lmw = 0 otherwise: an unlit segment's samples have opacity 0 and a flat
        //  alpha, nothing of theirs reaches d_volume -- see the pre-pass's definition of "lit")
        // Colour march after an alpha pre-pass (round 5): the pre-pass has marched this very segment, counted its samples and
        // found none that composites (alpha <= 1e-3 in a non-differentiable render, VR.py:334; opacity exactly 0 in a
        // differentiable one). Its count and its all-zero partial are in the workspace already -- nothing to march, nothing to
        // write. The brick still holds live samples of the view: the backward (alpha = 0 has a slope) must not skip it.
        const bool skip = has && lmw > 0 && (((lay < 64 ? d.um0 : d.um1) >> (lay & 63)) & 1ull);
        if (skip) has = false;
        const unsigned long long skm = __ballot(skip);
        if (skm != 0ull && (threadIdx.x & 63) == 0) {
            // (diagnostics, workspace_stats()[12]: every 64th workgroup reports -- atomics of EVERY wave on one word cost the
            //  CT-like 512^3 forward 3.3 ms)
            if (!BWD && count_stats) atomicAdd(&P.stats[ST_UNLIT_SKIPPED], (unsigned int)__popcll(skm));
            if constexpr (MODE == DR_MODE_DIFF && !BWD) *live_flag = 1;
        }
    }
    const unsigned long long hm = __ballot(has);
    const int lane_ = threadIdx.x & 63, wave_ = threadIdx.x >> 6;
    const int flen = has ? (s1 - s0 + KS - 1) / KS * KS : 0;  // flat length: a multiple of KS
    const int incl = wave_incl_sum(flen);
    nE = __popcll(hm);
    M = __builtin_amdgcn_readlane(incl, 63);
    if (has) {
        const int slot = wave_ * CW + __popcll(hm & ((1ull << lane_) - 1ull));
        const int off = incl - flen;
        L.ray0[slot] = make_float4(t0, exit_, nm1, 1.0f / nm1);  // n >= 2 (ray_is_regular)
        L.ray1[slot] = make_float4(vd.x, vd.y, vd.z, __int_as_float(pl));
        L.segi[slot] = lay * (P.W * P.H) + pl;
        L.s_rel[slot] = s0 - off;
        L.offs[slot + wave_] = off;
        L.slen[slot] = s1 - s0;
        L.valid[slot] = 0;
        if (BWD) L.live[slot] = live;
    }
    if (lane_ == 0) L.offs[wave_ * CW + nE + wave_] = M;  // end marker of this wave's table
    // the table is read back by other lanes of the SAME wave only: LDS operations of a wave complete in order
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// How a d_volume contribution reaches its LDS accumulator (dr_brick_common.h, "LDS gradient accumulators")
enum { ACC_FIX = 0,       // FIXED, value already carries the wave's factor (fix_wave_scale)
       ACC_F64 = 2,       // DOUBLE
       ACC_GLOBAL = 3 };  // not the LDS box at all: float atomics on the gradient volume (oversized adjoints, see B1)
template <int ACC>
__device__ __forceinline__ void acc_add(unsigned long long *p, float x, const FixScale &f) {
    if (ACC == ACC_F64) acc_add_f64(p, x);
    else fix_add_scaled(p, x, f);
}
// straight to the gradient volume in global memory (float atomics, as the baseline kernels and the reference do): the
// few samples whose adjoints exceed the fixed-point clamp (normalisation of a nearly vanishing gradient)
template <int ACC>
__device__ __forceinline__ void acc_add(float *p, float x, const FixScale &) { unsafeAtomicAdd(p, x); }
// SX, SY: element strides of the destination along x and y (z is contiguous in the LDS box; SZ for the global volume)
template <int ACC, typename PT, typename ST>
__device__ __forceinline__ void scatter8(PT *dst, ST SX, ST SY, ST SZ, const float (&w)[8], const FixScale &f) {
    acc_add<ACC>(dst, w[0], f);
    acc_add<ACC>(dst + SX, w[1], f);
    acc_add<ACC>(dst + SY, w[2], f);
    acc_add<ACC>(dst + SX + SY, w[3], f);
    acc_add<ACC>(dst + SZ, w[4], f);
    acc_add<ACC>(dst + SX + SZ, w[5], f);
    acc_add<ACC>(dst + SY + SZ, w[6], f);
    acc_add<ACC>(dst + SX + SY + SZ, w[7], f);
}
// the four voxels dst + {0, SA, SB, SA+SB} receive c * w[0..3]
template <int ACC, typename PT, typename ST>
__device__ __forceinline__ void scatter4(PT *dst, ST SA, ST SB, float c, const float (&w)[4], const FixScale &f) {
    acc_add<ACC>(dst, c * w[0], f);
    acc_add<ACC>(dst + SA, c * w[1], f);
    acc_add<ACC>(dst + SB, c * w[2], f);
    acc_add<ACC>(dst + SA + SB, c * w[3], f);
}
// Adjoint of the two central-difference taps of one axis, reduced to coefficients along that axis over the box
// planes l0-1 .. l0+2 (l0 = centre cell). The +delta tap sits in cell l0 or l0+1, the -delta tap in l0-1 or l0
// (delta < 1 voxel, brick_path_supported); a tap in cell l spreads (1-f, f) over planes l, l+1. Times g.
__device__ __forceinline__ void tap_line(int l0, int lp, int lm, float fp, float fm, float g, float (&cf)[4]) {
    const bool in_p = lp == l0, in_m = lm == l0;
    const float gp = 1.0f - fp, gm = 1.0f - fm;
    cf[0] = in_m ? 0.0f : -gm * g;
    cf[1] = ((in_p ? gp : 0.0f) - (in_m ? gm : fm)) * g;
    cf[2] = ((in_p ? fp : gp) - (in_m ? fm : 0.0f)) * g;
    cf[3] = in_p ? 0.0f : fp * g;
}

// d_volume scatter of one sample. Every tap is a product of per-axis weights, and the six normal taps differ from
// the centre tap along ONE axis only, so their adjoint factors into a 4-plane line along that axis (tap_line)
// times the centre's weights of the other two axes. Planes l0, l0+1 are the centre cell's own: everything that
// lands there is summed into the centre's 8 corners first (8 LDS adds); what remains per axis is one outside
// plane of 4 voxels (l0+2 or l0-1; both only when delta >= 0.5 voxel, i.e. dim > 1000: rare uniform branch).
// 8 + 3*4 = 20 LDS adds per sample instead of 8 per tap.
template <int ACC, typename PT, typename ST>
__device__ __forceinline__ void scatter_sample(PT *dst, ST SX, ST SY, ST SZ, const TapCoords &t, bool valid,
                                               float I_bar, const float (&gq)[3], const FixScale &fs) {
    const float X[2] = {1.0f - t.fx, t.fx}, Y[2] = {1.0f - t.fy, t.fy}, Z[2] = {1.0f - t.fz, t.fz};
    const float YZ[4] = {Y[0] * Z[0], Y[1] * Z[0], Y[0] * Z[1], Y[1] * Z[1]};  // offsets {0, SY, SZ, SY+SZ}
    const float XZ[4] = {X[0] * Z[0], X[1] * Z[0], X[0] * Z[1], X[1] * Z[1]};  // offsets {0, SX, SZ, SX+SZ}
    const float XY[4] = {X[0] * Y[0], X[1] * Y[0], X[0] * Y[1], X[1] * Y[1]};  // offsets {0, SX, SY, SX+SY}
    float cx[4], cy[4], cz[4];
    tap_line(t.lx, t.lxp, t.lxm, t.fxp, t.fxm, gq[0], cx);
    tap_line(t.ly, t.lyp, t.lym, t.fyp, t.fym, gq[1], cy);
    tap_line(t.lz, t.lzp, t.lzm, t.fzp, t.fzm, gq[2], cz);
    const float ax[2] = {fmaf(I_bar, X[0], cx[1]), fmaf(I_bar, X[1], cx[2])};
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const int ix = q & 1, iy = (q >> 1) & 1, iz = q >> 2;
        acc[q] = fmaf(ax[ix], YZ[iy + 2 * iz], fmaf(cy[1 + iy], XZ[ix + 2 * iz], cz[1 + iz] * XY[ix + 2 * iy]));
    }
    {   // x: outside plane l0+2 (coefficient cx[3]) or l0-1 (cx[0])
        const bool hi = cx[3] != 0.0f, lo = cx[0] != 0.0f;
        if (valid && (hi || lo)) scatter4<ACC>(dst + (hi ? 2 * SX : -SX), SY, SZ, hi ? cx[3] : cx[0], YZ, fs);
        if (__any(valid && hi && lo)) { if (valid && hi && lo) scatter4<ACC>(dst - SX, SY, SZ, cx[0], YZ, fs); }
    }
    {   // y
        const bool hi = cy[3] != 0.0f, lo = cy[0] != 0.0f;
        if (valid && (hi || lo)) scatter4<ACC>(dst + (hi ? 2 * SY : -SY), SX, SZ, hi ? cy[3] : cy[0], XZ, fs);
        if (__any(valid && hi && lo)) { if (valid && hi && lo) scatter4<ACC>(dst - SY, SX, SZ, cy[0], XZ, fs); }
    }
    {   // z
        const bool hi = cz[3] != 0.0f, lo = cz[0] != 0.0f;
        if (valid && (hi || lo)) scatter4<ACC>(dst + (hi ? 2 * SZ : -SZ), SX, SY, hi ? cz[3] : cz[0], XY, fs);
        if (__any(valid && hi && lo)) { if (valid && hi && lo) scatter4<ACC>(dst - SZ, SX, SY, cz[0], XY, fs); }
    }
    // (Consecutive lanes are consecutive samples of a ray, ~3.5 per cell: these eight adds collide in the LDS, ~7 cycles
    // per duplicate address. Summing the runs across lanes first was tried twice: +1.0 ms of VALU for 0.4 ms of LDS.)
    if (valid) scatter8<ACC>(dst, SX, SY, SZ, acc, fs);
}
// the common destination: the brick's LDS gradient box, centre cell at element cbase_i
template <int ACC>
__device__ __forceinline__ void scatter_sample(unsigned long long *dbox, const TapCoords &t, bool valid, int cbase_i,
                                               float I_bar, const float (&gq)[3], const FixScale &fs) {
    scatter_sample<ACC>(dbox + cbase_i, (int)BOX_SX, (int)BOX_SY, 1, t, valid, I_bar, gq, fs);
}

// Backward: the largest and the smallest non-zero |grad_out| (per pixel: its largest component) over the brick's
// candidate pixels whose rays hit the volume, per thread (the caller reduces them over the workgroup). A NaN component is
// ignored (that ray's adjoints are dropped); an infinite or absurd one simply makes the range huge, which sends the brick to
// double accumulators. All loads of a thread are independent and issued together with the box staging.
template <typename VT, int FNT>
__device__ __forceinline__ void cand_grad_range(const BrickParams<VT> &P, const BrickCtx &c, int view, int r_lo, int r_hi,
                                                const unsigned short *hits, int c_lo, int ncand_all, float &gmax, float &gmin) {
    const size_t vb = (size_t)view * P.W * P.H;
    const float4 *go4 = reinterpret_cast<const float4 *>(P.grad_out) + vb;
    const int nj = c.j1 - c.j0 + 1;
    const float rnj = __builtin_amdgcn_rcpf((float)nj);
    gmax = 0.0f; gmin = 3.0e38f;
    for (int idx = r_lo + threadIdx.x; idx < r_hi; idx += FNT) {
        const int cc = hits ? c_lo + (int)hits[idx] : idx;
        const int qi = (ncand_all < (1 << 21)) ? (int)(((float)cc + 0.5f) * rnj) : cc / nj;  // see cand_load
        const int pl = (c.i0 + qi) * P.H + c.j0 + (cc - qi * nj);
        const float4 g = go4[pl];
        const int n = P.nsamp[vb + pl];
        const float a = fmaxf(fmaxf(fabsf(g.x), fabsf(g.y)), fmaxf(fabsf(g.z), fabsf(g.w)));  // (NaN-suppressing)
        if (n > 0 && a > 0.0f) { gmax = fmaxf(gmax, a); gmin = fminf(gmin, a); }
    }
}
__device__ __forceinline__ float wave_max_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_min_f(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o));
    return v;
}

// ALPHA (forward only): the alpha pre-pass -- centre tap + TF only, the partial of a segment is its accumulated alpha.
// One work item: the candidate pixels [c_lo, c_hi) of one brick of one view (c_hi is clamped to the brick's candidate
// count). The main launch gives every brick its first MAIN_CAND candidates; bricks with more -- a camera close to or
// inside the volume: the brick around the eye is a candidate of every pixel -- had the rest cut into items of ITEM_CAND
// candidates by brick_ctx_kernel, which a second, small launch works off (brick_flat_items_kernel). Without that, the few
// bricks next to the camera would each be one workgroup's job: 61 ms instead of 6 for a 512^2 view from inside a 512^3 volume.
// TAPE (colour march of a differentiable render only): DR_TAPE_TF -- every marched sample is shaded (the TF gradient of a transparent
// sample needs its lighting term too) and leaves (intensity, lighting) on the per-sample tape; nothing is skipped.
template <typename VT, int MODE, bool BWD, bool WANT_VOL, bool WANT_TF, int ALPHA, int KF, bool HEAVY, bool NARROW, bool TAPE = false>
__device__ __forceinline__ void brick_flat_body(const BrickParams<VT> &P, unsigned char *smem, const int slot, const int view,
                                                const int c_lo, const int c_hi, int &box_valid) {   // box_valid: 0 no box staged, 1 staged, 2 staged and EMPTY
    if (ALPHA && P.vflags[view] == 0u) return;  // uniform: no ray of this view can terminate early
    using Cfg = FlatCfg<BWD, WANT_VOL, ALPHA>;
    constexpr int EC = Cfg::EC;
    constexpr int ROUND = Cfg::UNEVEN ? 4 * (8 * Cfg::UNEVEN + 4 * (8 - Cfg::UNEVEN)) : EC;  // candidates consumed per round (cand_load)
    constexpr int FNT = Cfg::FNT;
    constexpr int FNW = FNT / 64;
    constexpr bool BWD_TF = BWD && !WANT_VOL;             // backward w.r.t. the TF only
    constexpr int KS = BWD ? (BWD_TF ? DR_BWDTF_K : 1) : KF;  // consecutive samples per lane
    const int nbricks = P.g.NBx * P.g.NBy * P.g.NBz;
    int *live_flag = &const_cast<BrickCtxRec *>(P.ctx)[(size_t)view * nbricks + slot].live;   // "this brick holds live samples of the view"
    const bool count_stats = (slot & 63) == 0;   // diagnostics counters (ST_UNLIT_SKIPPED, ST_EMPTY_BRICKS) are SAMPLED: one workgroup in 64
    // words of the rays' "unlit" layer masks this pass may use (uniform): the forward's own; for the d_volume-only backward what the
    // forward LEFT (header word ST_MASKS, written with the fingerprint checked above), for every other backward none
    const int lmw = BWD ? ((WANT_VOL && !WANT_TF && P.lm_words > 0 && (int)P.stats[ST_MASKS] == P.lm_words) ? P.lm_words : 0) : P.lm_words;
    // backward: records, flags, items and coarse tape are only touched if THIS call's forward wrote them (ws_fingerprint)
    if (BWD && P.stats[ST_MARK] != P.mark) return;  // uniform; B2 then marches every ray
    const f3 cam = make_f3(P.cam[3 * view], P.cam[3 * view + 1], P.cam[3 * view + 2]);
    BrickCtx c;
    // the records are stored in dispatch order (near-first, brick_ctx_kernel): the address does not wait for the camera
    brick_ctx_load(P.ctx + (size_t)view * nbricks + slot, c);  // uniform address: scalar loads
    if (c.i0 > c.i1 || c.j0 > c.j1) return;  // uniform: the brick projects outside the image
    if (ALPHA) {  // uniform: is this brick part of this phase of the pre-pass? (camera inside the volume: one phase, all bricks)
        if (P.vflags[P.n_views + view] ? !P.pp_first : (c.layer < P.pp_l0 || c.layer >= P.pp_l1)) return;
    }
    // backward after a flat forward: bricks in which the forward marched nothing (rays terminated before them) have no work
    if (BWD && c.live == 0) return;  // uniform
    // ... and bricks the forward found EMPTY (no sample composites: every TF texel their voxels can index has alpha exactly 0) give
    // d_volume nothing: a sample's intensity adjoint is r_bar (r_hi - r_lo) + ... + a_bar (a_hi - a_lo) with r/g/b_bar = L op T go = 0
    // and a_hi = a_lo = 0, its normal-path adjoint carries the factor op = 0. (d_tf is another matter: alpha = 0 has a slope, the
    // air's texels collect a_bar from every such sample -- with a TF gradient wanted the brick is marched like any other.)
    if (BWD && WANT_VOL && !WANT_TF && (c.maybe_empty & 2) != 0) return;  // uniform

    __builtin_amdgcn_s_setprio(3);  // the staging / listing prologue is short and latency-bound: let it overtake sample loops
    FlatLds L = flat_carve<BWD, WANT_VOL, WANT_TF, ALPHA>(smem, P.R);
    VolView<VT> vol = P.vol;
    vol.p += view * P.vol_vs;
    const int NP = P.W * P.H;
    const size_t seg_view = (size_t)view * P.g.NL * NP;  // this view's [layer][pixel] slots
    const int ncand_all = (c.i1 - c.i0 + 1) * (c.j1 - c.j0 + 1);
    const int ncand = min(ncand_all, c_hi);  // this item's candidates: [c_lo, ncand)
    if (c_lo >= ncand) return;               // uniform
    // The rounds run over candidate indices [r_lo, r_hi). An overflow item of a heavy brick first puts its candidates through
    // the geometric pre-test (most of the bounding rectangle of a brick next to the eye are pixels whose lines miss it) and
    // runs its rounds over the compacted list of hits, in LDS behind the regular layout; no hit, no work.
    const unsigned short *hits = nullptr;
    int r_lo = c_lo, r_hi = ncand;
    if (HEAVY) {
        unsigned short *hl = reinterpret_cast<unsigned short *>(smem + align16(flat_lds_bytes<BWD>(P.R, WANT_VOL, WANT_TF, ALPHA)));
        int *nhit = reinterpret_cast<int *>(hl + ITEM_MAX_CAND);
        if (threadIdx.x == 0) *nhit = 0;
        __syncthreads();
        const CamBasis cbasis = make_cam_basis(P, cam, true);
        const int nj = c.j1 - c.j0 + 1;
        const float rnj = __builtin_amdgcn_rcpf((float)nj);
        for (int c0 = c_lo; c0 < ncand; c0 += FNT) {  // uniform
            const int cc = c0 + (int)threadIdx.x;
            bool hit = false;
            if (cc < ncand) {
                const int qi = (ncand_all < (1 << 21)) ? (int)(((float)cc + 0.5f) * rnj) : cc / nj;
                const int i = c.i0 + qi, j = c.j0 + (cc - qi * nj);
                const float u = ((float)(i + P.row0) + 0.5f) / (float)P.imgW - 0.5f, v = ((float)j + 0.5f) / (float)P.H - 0.5f;
                const f3 dir = make_f3(fmaf(v, cbasis.uh.x, fmaf(u, cbasis.rw.x, cbasis.dn.x)), fmaf(v, cbasis.uh.y, fmaf(u, cbasis.rw.y, cbasis.dn.y)),
                                       fmaf(v, cbasis.uh.z, fmaf(u, cbasis.rw.z, cbasis.dn.z)));
                hit = line_meets_brick(c, cam, dir);
            }
            const unsigned long long hm = __ballot(hit);
            int base = 0;
            if ((threadIdx.x & 63) == 0 && hm) base = atomicAdd(nhit, __popcll(hm));
            base = __builtin_amdgcn_readfirstlane(base);
            if (hit) hl[base + __popcll(hm & ((1ull << (threadIdx.x & 63)) - 1ull))] = (unsigned short)(cc - c_lo);
        }
        __syncthreads();
        hits = hl;
        r_lo = 0; r_hi = *nhit;
        if (r_hi == 0) return;  // uniform
    }
    // A work item that follows another item of the same brick in its workgroup finds the voxel box (and the TF) in LDS already.
    // (Forward and pre-pass only: the backward's gradient box is flushed per item.)
    const bool reuse_box = HEAVY && !BWD && box_valid != 0;  // uniform
    CandData cd;
    cand_load<VT, MODE, BWD, ALPHA, WANT_VOL>(P, c, view, r_lo, r_hi, hits, c_lo, ncand_all, cd, lmw);  // ray buffers of the first round's candidates
    BoxStage<FNT> stage;
    FixScale fs;
    int nE0, M0;
    // Forward after an alpha pre-pass that found terminating rays: many bricks lie entirely behind the termination
    // points (likewise in the later groups of the pre-pass itself). List the segments first and leave without staging
    // anything when there are none.
    const bool lazy = (!BWD && !ALPHA && P.use_live && P.vflags[view] != 0u) || (ALPHA && !P.pp_first);  // uniform
    if (lazy) {
        flat_build_entries<VT, MODE, BWD, WANT_VOL, ALPHA, KS>(P, c, cam, view, cd, L, nE0, M0, live_flag, count_stats, lmw);
        if (!__syncthreads_or(nE0 > 0) && r_hi - r_lo <= ROUND) return;  // uniform: no wave found a segment
        if (!reuse_box) box_issue<VT, FNT, ALPHA != 0>(P, vol, c, P.tf + view * P.tf_vs, stage);
    } else {
        float gm = 0.0f, gn = 3.0e38f;
        if (BWD && WANT_VOL) cand_grad_range<VT, FNT>(P, c, view, r_lo, r_hi, hits, c_lo, ncand_all, gm, gn);  // upstream gradients of the candidates,
        if (!reuse_box) box_issue<VT, FNT, ALPHA != 0>(P, vol, c, P.tf + view * P.tf_vs, stage);      // voxel box + TF: in flight ...
        if (BWD) {
            if (WANT_VOL) for (int k = threadIdx.x; k < BOX_LDS; k += FNT) L.dbox[k] = 0ull;
            if (WANT_TF) for (int k = threadIdx.x; k < 4 * P.R; k += FNT) L.dtf[k] = 0ull;
            if (WANT_VOL) {
                gm = wave_max_f(gm); gn = wave_min_f(gn);
                if ((threadIdx.x & 63) == 0) { L.gmax[threadIdx.x >> 6] = gm; L.gmax[8 + (threadIdx.x >> 6)] = gn; }
            }
        }
        flat_build_entries<VT, MODE, BWD, WANT_VOL, ALPHA, KS>(P, c, cam, view, cd, L, nE0, M0, live_flag, count_stats, lmw);  // ... while the segments are listed
    }
    if (!reuse_box) box_commit<VT, FNT, ALPHA != 0>(P, vol, c, P.tf + view * P.tf_vs, stage, L);
    const bool test_empty = !BWD && !TAPE && !reuse_box && (c.maybe_empty & 1) != 0;  // uniform
    if constexpr (!BWD) { if (test_empty) brick_empty_publish<FNT>(stage, L); }
    if (!reuse_box) box_valid = 1;
    __syncthreads();
    bool brick_empty = false;  // uniform: no sample of this brick composites anything (forward passes only)
    if constexpr (!BWD) {
        if (test_empty) {
            if (brick_empty_decide<VT, FNT, ALPHA != 0>(P, L)) {
                box_valid = 2;
                // the record remembers it (bit 1): a backward that only wants d_volume has nothing to do in this brick -- opacity 0
                // and a flat alpha (both texels exactly 0) make every d_volume term of its samples vanish (see B1 below)
                if (MODE == DR_MODE_DIFF && threadIdx.x == 0)
                    const_cast<BrickCtxRec *>(P.ctx)[(size_t)view * nbricks + slot].maybe_empty = 3;
            }
        }
        brick_empty = box_valid == 2;
        if (brick_empty && count_stats && threadIdx.x == 0) atomicAdd(&P.stats[ST_EMPTY_BRICKS], 1u);   // (diagnostics: workspace_stats()[13])
    }
    bool acc64 = false;  // brick-uniform: d_volume accumulates in double
    if (BWD && WANT_VOL) {
        // this brick's fixed-point scale, or doubles if its candidates' |grad_out| span more than 2^DR_MIXED_BITS
        // (every thread derives the same answer)
        float gm = 0.0f, gn = 3.0e38f;
#pragma unroll
        for (int k = 0; k < FNW; ++k) { gm = fmaxf(gm, L.gmax[k]); gn = fminf(gn, L.gmax[8 + k]); }
        acc64 = __builtin_amdgcn_readfirstlane((int)(gn * (float)(1 << DR_MIXED_BITS) < gm)) != 0;
        fs = make_fix_scale(gm);
        if (acc64 && threadIdx.x == 0) atomicAdd(&P.stats[ST_F64_BRICKS], 1u);
    }
    __builtin_amdgcn_s_setprio(0);
    const f3 light = make_f3(cam.x + 0.0f, cam.y + 1.0f, cam.z + 0.0f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    bool any = false;
    unsigned int n_eval = 0u;   // backward: samples this wave evaluated (wave-uniform; reported under DR_COUNT_EVALUATED)
    unsigned int n_eval_lane = 0u;   // forward passes: the same from the segment counts (per lane: one entry each)

    for (int cbase = r_lo; cbase < r_hi; cbase += ROUND) {
        int nE = nE0, M = M0;
        if (cbase > r_lo) {
            cand_load<VT, MODE, BWD, ALPHA, WANT_VOL>(P, c, view, cbase, r_hi, hits, c_lo, ncand_all, cd, lmw);
            flat_build_entries<VT, MODE, BWD, WANT_VOL, ALPHA, KS>(P, c, cam, view, cd, L, nE, M, live_flag, count_stats, lmw);  // syncs inside
        }
        any = any || nE > 0;
        // this wave's own segment table: entries [ea, eb), flat samples [0, M); offsets live at index entry + wave
        const int ea = wave * (EC / FNW), eb = ea + nE;
        const int *offs = L.offs + wave;
        bool skip_loop = false;  // wave-uniform
        if constexpr (!BWD) {
            if (brick_empty) {  // uniform: counts and all-zero partials, no sample is evaluated (empty_segment_count)
                const int e = ea + lane;
                int count = 0;
                bool resolved = true;
                float4 r1e = make_float4(0.f, 0.f, 0.f, 0.f);
                if (e < eb) {
                    const float4 r0e = L.ray0[e];
                    r1e = L.ray1[e];
                    const int s0e = L.s_rel[e] + offs[e];
                    resolved = empty_segment_count(vol, c, cam, r0e, r1e, s0e, s0e + L.slen[e], count);
                }
                if (!__any(!resolved)) {
                    skip_loop = true;
                    if (e < eb && count > 0) {
                        const int sgi = L.segi[e];
                        L.valid[e] = count;   // (written to seg_cnt with the other counts below)
                        P.seg_rgba[seg_view + sgi] = make_float4(0.f, 0.f, 0.f, 0.f);
                        if constexpr (ALPHA != 0) {
                            if (P.lm_words > 0) {   // tell the colour march that this segment is done (see flat_build_entries)
                                const int plq = __float_as_int(r1e.w);
                                const int lay = (int)((float)(sgi - plq) * __builtin_amdgcn_rcpf((float)NP) + 0.5f);
                                atomicOr(P.unlit + ((size_t)view * P.lm_words + (lay >> 6)) * NP + plq, 1ull << (lay & 63));
                            }
                        }
                    }
                }
            }
        }
        const int fa = 0, fb = skip_loop ? 0 : M;
        Over carry = {0.f, 0.f, 0.f, 0.f};
        Over2 carry2 = {0.f, 0.f};  // the backward's pair (gC . C, A)
        int carry_e = -1;  // entry whose composite so far is in `carry` (continues into the next chunk)
        int e_cur = ea;
        for (int f0 = fa, ks = 1; f0 < fb; f0 += 64 * ks) {
            // samples per lane in this pass: KS, but the tail of the wave's samples takes the smallest power of two
            // that still covers it with 64 lanes (a half-empty last pass would cost all KS sub-samples)
            const int rem = fb - f0;
            const int ksh = (KS >= 4 && rem > 128) ? 2 : ((KS >= 2 && rem > 64) ? 1 : 0);  // uniform
            ks = 1 << ksh;
            const int f = f0 + ks * lane;
            const bool act = f < fb;
            // entry of this lane: advance from the previous chunk's entry (flat order is entry order)
            if (act) { while (f >= offs[e_cur + 1]) ++e_cur; }
            const int e = act ? e_cur : eb - 1;
            const int eoff = offs[e];
            const int sl = max(lane - ((f - eoff) >> ksh), 0);  // first lane of this lane's segment within the chunk
            const float4 r0 = L.ray0[e], r1 = L.ray1[e];
            const int s = f + L.s_rel[e];
            const f3 vd = make_f3(r1.x, r1.y, r1.z);
            // backward: the per-ray inputs of the adjoint are requested now and consumed ~800 issue cycles later
            float4 pf_pre = make_float4(0.f, 0.f, 0.f, 0.f), pf_go = pf_pre, pf_of = pf_pre;
            if (BWD && act) {
                const int plq = __float_as_int(r1.w);
                pf_pre = P.seg_rgba[seg_view + L.segi[e]];
                pf_go = reinterpret_cast<const float4 *>(P.grad_out)[(size_t)view * NP + plq];
                pf_of = P.fin[(size_t)view * NP + plq];   // (F2's own final composite, not the image: ray_exact_kernel may have rewritten that)
            }
            if (ALPHA) {
                // alpha pre-pass: position, centre cell, one tap, TF -> transmittance; nothing else.
                // KS consecutive samples per lane, multiplied up in registers before the cross-lane product scan.
                const int slen = L.slen[e];
                float Tl = 1.0f;
                // in-brick samples of this lane, and in the bits from ALPHA_TINY_SHIFT up those of them with a tiny opacity (D4): one number,
                // one scan -- a piece has at most 64 * KS samples, so neither field of the piece's sum leaves its bits nor the float's 24
                constexpr int ALPHA_TINY_SHIFT = 10;
                static_assert(64 * KS < (1 << ALPHA_TINY_SHIFT) && 64 * KS <= (1 << (24 - ALPHA_TINY_SHIFT)), "packed piece counts");
                int cnt_lane = 0;
                bool lit_lane = false;  // some sample of this lane composites (alpha > 1e-3 / opacity != 0)
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    if (j >= ks) continue;  // uniform
                    Sample sa;
                    int x0 = 0, y0 = 0, z0 = 0;
                    float fx = 0.f, fy = 0.f, fz = 0.f;
                    bool va = act && (f + j - eoff) < slen;
                    if (va) {
                        sample_pos_rcp(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, cam.x, cam.y, cam.z, s + j, sa.px, sa.py, sa.pz);
                        axis_coord(sa.px, vol.scx, x0, fx); axis_coord(sa.py, vol.scy, y0, fy); axis_coord(sa.pz, vol.scz, z0, fz);
                        va = (unsigned)(x0 - c.ox - 1) < (unsigned)BRK && (unsigned)(y0 - c.oy - 1) < (unsigned)BRK &&
                             (unsigned)(z0 - c.oz - 1) < (unsigned)BRK;
                    }
                    if (va) {
                        sa.I = tri_lds(L.box, (x0 - c.ox) * BOX_SX + (y0 - c.oy) * BOX_SY + (z0 - c.oz), fx, fy, fz);
                        tf_alpha_from_I(L.tfa, P.R, P.tf_len, sa);
                        if constexpr (MODE != DR_MODE_NONDIFF) {
                            const float opj = opacity_of_alpha(sa.a, P.inv_sr);
                            // "lit" for the masks: the colour march's own test (c = L * rgb * op is exactly 0 for op == 0) AND a flat
                            // alpha -- both texels exactly 0 -- so that the same bit also tells the d_volume-only backward that the
                            // sample's intensity adjoint vanishes (it carries the factor a_hi - a_lo next to terms with the factor op)
                            lit_lane = lit_lane || opj != 0.0f || L.tfa[sa.lo] != 0.0f || L.tfa[sa.hi] != 0.0f;
                            Tl *= 1.0f - opj;
                            cnt_lane += (opj != 0.0f && opj < DR_D4_TINY_OP) ? (1 << ALPHA_TINY_SHIFT) : 0;
                        }
                        ++cnt_lane;
                    }
                    // the opacity (a power at sampling rates other than 1) only where it counts: the nondiff march skips
                    // alpha <= 1e-3 (VR.py:334), and transparent stretches are wave-uniform (lanes = consecutive samples)
                    if constexpr (MODE == DR_MODE_NONDIFF) {
                        const bool vis = va && sa.a > 1e-3f;
                        lit_lane = lit_lane || vis;
                        if (__any(vis)) {
                            if (vis) {
                                const float opj = opacity_of_alpha(sa.a, P.inv_sr);
                                Tl *= 1.0f - opj;
                                cnt_lane += (opj < DR_D4_TINY_OP) ? (1 << ALPHA_TINY_SHIFT) : 0;
                            }
                        }
                    }
                }
                Tl = seg_scan_prod(Tl, lane, sl);
                const int e_first = __builtin_amdgcn_readfirstlane(e);
                if (carry_e == e_first && e == e_first) Tl *= carry.a;  // carry.a holds the transmittance so far
                {
                    const int e_last = __builtin_amdgcn_readlane(e, 63);
                    const float lastT = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(Tl), 63));
                    const bool more = (f0 + 64 * ks < fb) && (offs[e_last + 1] > f0 + 64 * ks);
                    carry.a = lastT; carry_e = more ? e_last : -1;
                }
                const bool seg_end = act && (f + ks >= offs[e + 1]);
                // in-brick samples of the piece [sl, lane]: inclusive lane-sum of the per-lane counts
                float cf[1] = {(float)cnt_lane};
                seg_scan_sum<1>(cf, lane, sl);
                const bool piece_end = act && (seg_end || lane == 63 || f + ks >= fb);
                // nondiff: does the piece [sl, lane] hold a lit sample? (bit VALID_LIT of the segment's counter collects the pieces)
                const unsigned long long litm = __ballot(lit_lane);
                if (piece_end) {
                    const int cpk = (int)cf[0];   // (the tiny count rides in the upper bits, here and in L.valid)
                    const int cntp = (cpk & ((1 << ALPHA_TINY_SHIFT) - 1)) + ((cpk >> ALPHA_TINY_SHIFT) << VALID_TINY_SHIFT);
                    int before = cntp ? atomicAdd(&L.valid[e], cntp) : L.valid[e];
                    bool seg_lit = true;
                    {
                        const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                        const bool piece_lit = (litm & upto & ~((1ull << sl) - 1ull)) != 0ull;
                        seg_lit = piece_lit || (before & VALID_LIT);
                        if (piece_lit && !(before & VALID_LIT)) atomicOr(&L.valid[e], VALID_LIT);
                        before &= VALID_LIT - 1;
                    }
                    if (seg_end && ((before + cntp) & VALID_CNT_MASK) > 0) {
                        const int sgi = L.segi[e];
                        P.seg_rgba[seg_view + sgi] = make_float4(0.f, 0.f, 0.f, 1.0f - Tl);
                        {
                            // every sample of the ray in this brick was marched (the pre-pass has no live limit) and none is lit: tell the
                            // colour march (flat_build_entries). Layer = (slot - pixel) / NP through the float reciprocal: exact, the
                            // quotient is an integer below 128 and the error of the product below 1e-4.
                            if (!seg_lit && P.lm_words > 0) {
                                const int plq = __float_as_int(r1.w);
                                const int lay = (int)((float)(sgi - plq) * __builtin_amdgcn_rcpf((float)NP) + 0.5f);
                                atomicOr(P.unlit + ((size_t)view * P.lm_words + (lay >> 6)) * NP + plq, 1ull << (lay & 63));
                            }
                        }
                    }
                }
                continue;
            }
            if constexpr (BWD_TF && KS == 2) {
                // ---- Backward w.r.t. the TF only (C3): TWO consecutive samples per lane, like the forward. Without a gradient
                // box to feed, what a sample must carry across the scan is six registers (L, op, gC.rgb, alpha, TF cell and
                // fraction), so the cross-lane work -- the scan of (gC.C, A), the run sums of d_tf, the three per-ray loads,
                // the entry walk -- is paid once per 128 samples. d_tf is a continuous function of the lighting term
                // (min(1, L) enters, not its switch), so the kink re-shading of D7 is not needed here.
                const int slen = L.slen[e];
                const int live_e = L.live[e];
                const float4 go = pf_go;
                const float gpre = go.x * pf_pre.x + go.y * pf_pre.y + go.z * pf_pre.z, apre = pf_pre.w;  // stored prefix as (gC . C, A)
                const float gfin = go.x * pf_of.x + go.y * pf_of.y + go.z * pf_of.z, afin = pf_of.w;      // the final composite likewise
                float kL[KS], kop[KS], krd[KS], kfr[KS], ka[KS];
                int klo[KS], khi[KS];
                bool kval[KS];
                Over2 el2 = {0.f, 0.f};
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    kL[j] = kop[j] = krd[j] = kfr[j] = ka[j] = 0.f; klo[j] = khi[j] = 0; kval[j] = false;
                    if (j >= ks) continue;  // uniform
                    Sample sm; TapCoords t;
                    const bool actj = act && (f + j - eoff) < slen;
                    bool vj = false;
                    if (actj) {
                        sample_pos_rcp(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, cam.x, cam.y, cam.z, s + j, sm.px, sm.py, sm.pz);
                        vj = sample_coords_at(vol, c, sm, t);
                    }
                    Over2 ej = {0.f, 0.f};
                    n_eval += (unsigned int)__popcll(__ballot(vj));
                    if (vj) {
                        float dx, dy, dz;
                        CentreLerps cl;
                        sm.I = sample_centre_lds_keep(L.box, t, cl);
                        classify_from_I(L.tf, P.R, P.tf_len, P.inv_sr, sm);
                        sample_normal_taps_shared_lds<NARROW>(L.box, t, cl, dx, dy, dz);
                        shade_from_grad<true>(dx, dy, dz, light, vd, true, sm);
                        const float rd = go.x * sm.r + go.y * sm.g + go.z * sm.b;
                        kL[j] = sm.L; kop[j] = sm.op; krd[j] = rd; kfr[j] = sm.fr; ka[j] = sm.a; klo[j] = sm.lo; khi[j] = sm.hi;
                        kval[j] = true;
                        ej.w = (sm.L * sm.op) * rd; ej.a = sm.op;
                    }
                    el2 = (j == 0) ? ej : over2(el2, ej);  // over2(x, 0) == x exactly
                }
                // segmented scan of the lanes' composites; the first segment may continue an entry begun in an earlier chunk
                const int e_first = __builtin_amdgcn_readfirstlane(e);
                const bool cont = (carry_e == e_first);
                const int e_last = __builtin_amdgcn_readlane(e, 63);
                const bool more = (f0 + 64 * ks < fb) && (offs[e_last + 1] > f0 + 64 * ks);
                Over2 inc2 = seg_scan_over2(el2, lane, sl), exc2;
                exc2.w = wave_up1(inc2.w, 0.f); exc2.a = wave_up1(inc2.a, 0.f);
                if (lane == sl) { exc2.w = 0.f; exc2.a = 0.f; }
                if (cont && e == e_first) { inc2 = over2(carry2, inc2); exc2 = over2(carry2, exc2); }
                carry2.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.w), 63));
                carry2.a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.a), 63));
                carry_e = more ? e_last : -1;
                // adjoints of the lane's samples, front to back (tape-free identity, as sample_adjoint)
                float tv[KS][4];
                {
                    Over2 cur = exc2;  // composite before the sample, within this brick's part of the segment
                    const float Tpre = 1.0f - apre;
#pragma unroll
                    for (int j = 0; j < KS; ++j) {
                        tv[j][0] = tv[j][1] = tv[j][2] = tv[j][3] = 0.f;
                        if (kval[j]) {
                            Over2 ej; ej.w = (kL[j] * kop[j]) * krd[j]; ej.a = kop[j];
                            const Over2 after = over2(cur, ej);
                            const float absw = fmaf(Tpre, after.w, gpre), absa = fmaf(Tpre, after.a, apre);
                            const float T = Tpre * (1.0f - cur.a);  // transmittance before the sample
                            const bool last = (s + j == live_e - 1);
                            const float suffix = (gfin - absw) + go.w * (afin - absa);
                            const float qs = kL[j] * krd[j] + go.w;
                            const float sfx = suffix * __builtin_amdgcn_rcpf(1.0f - kop[j]);
                            const float op_bar = T * qs - (last ? 0.0f : sfx);
                            const float Lop = kL[j] * kop[j] * T;
                            const float a_bar = op_bar * ((P.inv_sr == 1.0f) ? 1.0f : P.inv_sr * powf(1.0f - ka[j], P.inv_sr - 1.0f));
                            const float w0 = 1.0f - kfr[j], w1 = kfr[j];
                            tv[j][0] = w0 * Lop; tv[j][1] = w1 * Lop; tv[j][2] = w0 * a_bar; tv[j][3] = w1 * a_bar;
                            cur = after;
                        }
                    }
                }
                // d_tf: runs of consecutive samples between the same two texels are summed across lanes (DPP) and added once.
                // A lane whose two samples fall into different TF cells ("split") closes the incoming run with its first sample
                // and opens a new one with its second. Runs stop at segment boundaries (the upstream colour gradient go.xyz is
                // per ray) and at the ends of the pass.
                const bool vA = kval[0], vB = kval[1];
                const bool split = vA && vB && klo[0] != klo[1];
                const int key_in = vA ? klo[0] : (vB ? klo[1] : -1 - lane);
                const int key_out = vB ? klo[1] : (vA ? klo[0] : -1 - lane);
                const int hi_out = vB ? khi[1] : khi[0];
                const int prev_out = wave_up1(key_out, key_out);
                const bool contl = lane != 0 && lane != sl && key_in == prev_out;  // continues the run of the lane before
                const bool start = !contl || split;                                // the lane's outgoing value starts a run
                float V[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) V[q] = split ? tv[1][q] : tv[0][q] + tv[1][q];
                const unsigned long long starts = __ballot(start);
                const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                const int rs = 63 - __clzll((long long)(starts & upto));  // first lane of this lane's outgoing run (lane 0 starts one)
                seg_scan_sum<4>(V, lane, rs);
                const unsigned long long contm = __ballot(contl);
                const bool run_end = lane == 63 || !((contm >> ((lane + 1) & 63)) & 1ull);
                auto emit = [&](bool em, const float (&tq)[4], int lo, int hi) {
                    const float v8[8] = {tq[0] * go.x, tq[0] * go.y, tq[0] * go.z, tq[2], tq[1] * go.x, tq[1] * go.y, tq[1] * go.z, tq[3]};
                    const float vmax = ((fabsf(v8[0]) + fabsf(v8[1])) + (fabsf(v8[2]) + fabsf(v8[3]))) +
                                       ((fabsf(v8[4]) + fabsf(v8[5])) + (fabsf(v8[6]) + fabsf(v8[7])));
                    unsigned long long *d0 = L.dtf + 4 * lo, *d1 = L.dtf + 4 * hi;
                    if (__any(em && !(vmax <= ACC_LIM))) {  // a NaN or an absurd run total: the sanitising path
                        if (em) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) { acc_add_f64(d0 + q, acc_sanitise(v8[q])); acc_add_f64(d1 + q, acc_sanitise(v8[4 + q])); }
                        }
                    } else if (em) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) { acc_add_f64(d0 + q, v8[q]); acc_add_f64(d1 + q, v8[4 + q]); }
                    }
                };
                emit((vA || vB) && run_end, V, key_out, hi_out);
                if (__any(split)) {  // uniform
                    // (the DPP moves are pinned in front of the select: the compiler otherwise turns `contl ? dpp : 0` into an
                    // exec-masked region around the v_mov_dpp, and a DPP move whose SOURCE lane is masked off writes nothing)
                    float Pv[4], H[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q) Pv[q] = wave_up1(V[q], 0.f);   // inclusive sum of the previous lane's outgoing run
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(Pv[q]));
#pragma unroll
                    for (int q = 0; q < 4; ++q) H[q] = (contl ? Pv[q] : 0.0f) + tv[0][q];
                    emit(split, H, klo[0], khi[0]);
                }
                continue;
            }
            Sample sm; TapCoords t;
            bool valid = false;
            float dx = 0.f, dy = 0.f, dz = 0.f;
            Over el = {0.f, 0.f, 0.f, 0.f};
            Over2 el2 = {0.f, 0.f};  // backward
            unsigned long long vm_fwd[KS];  // forward: which lanes hold an in-brick sample, per sub-sample
            unsigned int tbits = 0u;        // ... bit j: this lane's sub-sample j is lit with a tiny opacity (D4; kept per lane: masks held in
                                            // scalar registers across the sample loop cost spills)
            // Lighting only matters where the sample has opacity: c = L*rgb*op is exactly 0 for op == 0 whatever L
            // is (the nondiff path skips alpha <= 1e-3 by definition, VR.py:334). Lanes are consecutive samples of a
            // ray, so empty stretches of the transfer function are wave-uniform: skip the six normal taps (48 of the
            // 56 LDS reads) and the shading for the whole wave. The backward always needs L (d/d alpha).
            CentreLerps cl;
            cl.a0 = cl.b0 = cl.a1 = cl.b1 = cl.zl = cl.zh = 0.f;
            if (KS == 1) {
                if (act) {
                    sample_pos_rcp(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, cam.x, cam.y, cam.z, s, sm.px, sm.py, sm.pz);
                    valid = BWD ? sample_coords_at(vol, c, sm, t) : sample_centre_coords_at(vol, c, sm, t);
                }
                if (valid) {
                    sm.I = sample_centre_lds_keep(L.box, t, cl);
                    classify_from_I(L.tf, P.R, P.tf_len, P.inv_sr, sm);
                }
                const bool lit = !ALPHA && valid && (BWD || (MODE == DR_MODE_NONDIFF ? (sm.a > 1e-3f) : (sm.op != 0.0f)));
                if (BWD || __any(lit)) {
                    if (lit) {
                        if (!BWD) sample_normal_coords_at(vol, c, sm, t);
                        sample_normal_taps_shared_lds<NARROW>(L.box, t, cl, dx, dy, dz);
                        shade_from_grad<true>(dx, dy, dz, light, vd, MODE == DR_MODE_DIFF, sm);
                    }
                    // Kinks of the lighting model: the adjoint switches on 1 < Lraw (the clamp of VR.py:298),
                    // 0 < n.l and 0 < r.v (the max(., 0) of :291,:294). A sample within rounding distance of one of them is
                    // shaded again with the oracle's exact normalisations, so that the switch falls as it does there.
                    if (BWD && __any(lit && near_lighting_kink(sm))) {
                        if (lit && near_lighting_kink(sm)) shade_from_grad<false>(dx, dy, dz, light, vd, MODE == DR_MODE_DIFF, sm);
                    }
                    if (lit) {
                        if (BWD) { el2.w = (sm.L * sm.op) * (pf_go.x * sm.r + pf_go.y * sm.g + pf_go.z * sm.b); el2.a = sm.op; }
                        else { el.c0 = sm.L * sm.r * sm.op; el.c1 = sm.L * sm.g * sm.op; el.c2 = sm.L * sm.b * sm.op; el.a = sm.op; }
                    }
                }
                {
                    const unsigned long long vmask = __ballot(valid);
                    if (BWD) n_eval += (unsigned int)__popcll(vmask);   // (the forward passes count from their segment counts, below)
                    vm_fwd[0] = BWD ? 0ull : vmask;
                    if (!BWD && lit && sm.op < DR_D4_TINY_OP) tbits = 1u;
                }
            } else {
                // forward, KS consecutive samples per lane: composited in registers, so that the cross-lane scan and
                // the per-chunk bookkeeping below are paid once per KS*64 samples
                const int slen = L.slen[e];
                float2 tape_v[KS];   // TAPE: what the lane's samples leave on the per-sample tape ...
                bool tape_ok[KS];    // ... those that are marched at all
#pragma unroll
                for (int j = 0; j < KS; ++j) { tape_v[j] = make_float2(0.f, 0.f); tape_ok[j] = false; }
#pragma unroll
                for (int j = 0; j < KS; ++j) {
                    if (j >= ks) { vm_fwd[j] = 0ull; continue; }  // uniform
                    const bool actj = act && (f + j - eoff) < slen;
                    bool vj = false;
                    if (actj) {
                        sample_pos_rcp(r0.x, r0.y, r0.z, r0.w, r1.x, r1.y, r1.z, cam.x, cam.y, cam.z, s + j, sm.px, sm.py, sm.pz);
                        vj = sample_centre_coords_at(vol, c, sm, t);
                    }
                    if (vj) {
                        sm.I = sample_centre_lds_keep(L.box, t, cl);
                        tf_lookup_from_I(L.tf, P.R, P.tf_len, sm);
                        if (MODE != DR_MODE_NONDIFF) sm.op = opacity_of_alpha(sm.a, P.inv_sr);
                    }
                    const bool lit = vj && (MODE == DR_MODE_NONDIFF ? (sm.a > 1e-3f) : (TAPE || sm.op != 0.0f));
                    Over ej = {0.f, 0.f, 0.f, 0.f};
                    if (__any(lit)) {
                        if (lit) {
                            if (MODE == DR_MODE_NONDIFF) sm.op = opacity_of_alpha(sm.a, P.inv_sr);  // (a power: only where lit)
                            sample_normal_coords_at(vol, c, sm, t);  // (not before: transparent stretches never need them)
                            sample_normal_taps_shared_lds<NARROW>(L.box, t, cl, dx, dy, dz);
                            shade_from_grad<true>(dx, dy, dz, light, vd, MODE == DR_MODE_DIFF, sm);
                            ej.c0 = sm.L * sm.r * sm.op; ej.c1 = sm.L * sm.g * sm.op; ej.c2 = sm.L * sm.b * sm.op; ej.a = sm.op;
                            if (sm.op < DR_D4_TINY_OP && (!TAPE || sm.op != 0.0f)) tbits |= 1u << j;
                            if constexpr (TAPE) {
                                // (a ray longer than the tape's stride -- ray buffers made for another sampling rate than this call's -- stays
                                //  inside its slot: F2 hands such a ray to the per-ray kernels, forward and backward)
                                tape_v[j] = make_float2(sm.I, sm.L); tape_ok[j] = s + j < P.tape_stride;
                            }
                        }
                    }
                    vm_fwd[j] = __ballot(vj);
                    el = (j == 0) ? ej : over(el, ej);  // over(x, 0) == x exactly
                }
                if constexpr (TAPE) {
                    // lanes are consecutive samples of a ray, a lane's own samples neighbours on the tape: one 16-byte store per pair
                    // (1 KB contiguous per wave and ray), single samples -- the ends of a segment -- on their own
                    float2 *tq = P.tape + ((size_t)view * NP + (size_t)__float_as_int(r1.w)) * (size_t)P.tape_stride + (size_t)s;
#pragma unroll
                    for (int h = 0; h < KS / 2; ++h) {
                        if (tape_ok[2 * h] && tape_ok[2 * h + 1]) {
                            // (only 8-byte aligned where the ray's first sample in this brick has an odd index: global memory takes that)
                            *reinterpret_cast<float4 *>(tq + 2 * h) = make_float4(tape_v[2 * h].x, tape_v[2 * h].y, tape_v[2 * h + 1].x, tape_v[2 * h + 1].y);
                        } else {
                            if (tape_ok[2 * h]) tq[2 * h] = tape_v[2 * h];
                            if (tape_ok[2 * h + 1]) tq[2 * h + 1] = tape_v[2 * h + 1];
                        }
                    }
                }
            }
            // segmented inclusive scan of "over" across the wave (segments = entries)
            // the first segment may continue an entry begun in an earlier chunk of this wave
            const int e_first = __builtin_amdgcn_readfirstlane(e);
            const bool cont = (carry_e == e_first);
            const int e_last = __builtin_amdgcn_readlane(e, 63);
            const bool more = (f0 + 64 * ks < fb) && (offs[e_last + 1] > f0 + 64 * ks);
            Over inc = {0.f, 0.f, 0.f, 0.f};
            Over2 inc2 = {0.f, 0.f}, exc2 = {0.f, 0.f};   // backward: composite up to and including / before this sample, within the chunk
            if (!BWD) {
                inc = seg_scan_over(el, lane, sl);
                if (cont && e == e_first) inc = over(carry, inc);
                // carry out: composite of the last lane's entry if it continues past this chunk
                carry = readlane_over(inc, 63);
            } else {
                inc2 = seg_scan_over2(el2, lane, sl);
                exc2.w = wave_up1(inc2.w, 0.f); exc2.a = wave_up1(inc2.a, 0.f);
                if (lane == sl) { exc2.w = 0.f; exc2.a = 0.f; }
                if (cont && e == e_first) { inc2 = over2(carry2, inc2); exc2 = over2(carry2, exc2); }
                carry2.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.w), 63));
                carry2.a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.a), 63));
            }
            carry_e = more ? e_last : -1;
            const bool seg_end = act && (f + ks >= offs[e + 1]);
            if (!BWD) {
                // count the in-brick samples of each segment piece, store finished segments
                const bool piece_end = act && (seg_end || lane == 63 || f + ks >= fb);
                const unsigned long long below = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                const unsigned long long from = ~((1ull << sl) - 1ull);
                int tiny_piece = 0;   // samples of tiny opacity in the piece [sl, lane]
                if (__ballot(tbits != 0u) != 0ull) {   // wave-uniform, rare: the masks are made here, where all lanes are
#pragma unroll
                    for (int j = 0; j < KS; ++j) tiny_piece += __popcll(__ballot(((tbits >> j) & 1u) != 0u) & below & from);
                }
                if (piece_end) {
                    int cntp = tiny_piece << VALID_TINY_SHIFT;
#pragma unroll
                    for (int j = 0; j < KS; ++j) cntp += __popcll(vm_fwd[j] & below & from);
                    const int before = cntp ? atomicAdd(&L.valid[e], cntp) : L.valid[e];
                    // a (ray, layer) slot belongs to the one brick that holds samples of the ray: a candidate
                    // segment without any in-brick sample must not touch it
                    if (seg_end && ((before + cntp) & VALID_CNT_MASK) > 0)
                        P.seg_rgba[seg_view + L.segi[e]] = make_float4(inc.c0, inc.c1, inc.c2, inc.a);
                }
            } else {
                SampleAdj ad;
                ad.r_bar = ad.g_bar = ad.b_bar = ad.a_bar = 0.f; ad.gx = ad.gy = ad.gz = 0.f; ad.Lop = 0.f;
                float4 go = make_float4(0.f, 0.f, 0.f, 0.f);
                if (valid) {
                    const float4 pre = pf_pre, of = pf_of;  // lanes of a chunk share a few rays: broadcast-like cached loads
                    go = pf_go;
                    // composite up to and including s, as (gC . C, A): the segment's stored prefix, then the chunk's scan
                    const float Tpre = 1.0f - pre.w;
                    const float absw = fmaf(Tpre, inc2.w, go.x * pre.x + go.y * pre.y + go.z * pre.z);
                    const float absa = fmaf(Tpre, inc2.a, pre.w);
                    const float T = Tpre * (1.0f - exc2.a);                  // transmittance before s
                    const bool last = (s == L.live[e] - 1);
                    const float suffix = ((go.x * of.x + go.y * of.y + go.z * of.z) - absw) + go.w * (of.w - absa);
                    sample_adjoint<true>(sm, vd, T, suffix, last, go, P.inv_sr, ad);
                }
                if (WANT_TF) {
                    // neighbouring lanes are consecutive samples of a ray: long runs fall between the same two
                    // texels. Sum each run across lanes (DPP) and let its last lane do the eight LDS adds. Runs stop
                    // at segment boundaries, so the upstream colour gradient go.xyz is constant over a run and
                    // (r,g,b)_bar = (L*op*T) * go.xyz is applied to the run total: 4 scanned values, not 8.
                    const int key = valid ? sm.lo : -1 - lane;
                    const int key_prev = wave_up1(key, key);
                    const bool run_start = lane == 0 || key != key_prev || lane == sl;
                    // first lane of this lane's run = highest run-start bit at or below the lane (lane 0 always starts one)
                    const unsigned long long starts = __ballot(run_start);
                    const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
                    const int rs = 63 - __clzll((long long)(starts & upto));
                    const float w0 = 1.0f - sm.fr, w1 = sm.fr;
                    float tv[4] = {w0 * ad.Lop, w1 * ad.Lop, w0 * ad.a_bar, w1 * ad.a_bar};
                    seg_scan_sum<4>(tv, lane, rs);
                    const bool run_end = lane == 63 || ((starts >> (lane + 1)) & 1ull);
                    const bool emit = valid && run_end;
                    const float v8[8] = {tv[0] * go.x, tv[0] * go.y, tv[0] * go.z, tv[2],
                                         tv[1] * go.x, tv[1] * go.y, tv[1] * go.z, tv[3]};
                    // sum of magnitudes (>= the largest one; full-rate adds, and a NaN propagates into the test)
                    const float vmax = ((fabsf(v8[0]) + fabsf(v8[1])) + (fabsf(v8[2]) + fabsf(v8[3]))) +
                                       ((fabsf(v8[4]) + fabsf(v8[5])) + (fabsf(v8[6]) + fabsf(v8[7])));
                    // d_tf accumulates in double; a NaN or an absurd run total takes the sanitising path
                    if (__any(emit && !(vmax <= ACC_LIM))) {
                        if (emit) {
                            unsigned long long *d0 = L.dtf + 4 * sm.lo, *d1 = L.dtf + 4 * sm.hi;
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                acc_add_f64(d0 + q, acc_sanitise(v8[q]));
                                acc_add_f64(d1 + q, acc_sanitise(v8[4 + q]));
                            }
                        }
                    } else if (emit) {
                        unsigned long long *d0 = L.dtf + 4 * sm.lo, *d1 = L.dtf + 4 * sm.hi;
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            acc_add_f64(d0 + q, v8[q]);
                            acc_add_f64(d1 + q, v8[4 + q]);
                        }
                    }
                }
                if (WANT_VOL) {
                    // d_volume: fixed-point adds into the LDS gradient box (dr_brick_common.h). A 32-bit addend
                    // suffices unless some adjoint of the wave exceeds 2^31 / 2^shift (then: exact wide path).
                    // the 24 tap coordinates are cheap to rebuild from the position (45 VALU) and expensive to keep
                    // alive across shading and the adjoint (the kernel is register-bound: spills go to scratch)
                    const int cbase_i = valid ? (t.lx * BOX_SX + t.ly * BOX_SY + t.lz) : 0;
                    float I_bar = 0.f;
                    float gq[3] = {0.f, 0.f, 0.f};
                    if (valid) {
                        I_bar = intensity_adjoint(sm, L.tf[sm.lo], L.tf[sm.hi], ad, P.tf_len);
                        if (!sm.flat) { gq[0] = ad.gx; gq[1] = ad.gy; gq[2] = ad.gz; }
                    }
                    // (a NaN or an overflow makes the magnitude test fail: the exact path clamps, the common one need not)
                    const float bound = fabsf(I_bar) + (fabsf(gq[0]) + fabsf(gq[1]) + fabsf(gq[2]));
                    if (acc64) {  // brick-uniform
                        if (__any(!(bound <= ACC_LIM))) {
                            I_bar = acc_sanitise(I_bar);
                            gq[0] = acc_sanitise(gq[0]); gq[1] = acc_sanitise(gq[1]); gq[2] = acc_sanitise(gq[2]);
                        }
                        scatter_sample<ACC_F64>(L.dbox, t, valid, cbase_i, I_bar, gq, fs);
                    } else {
                        // Beyond the range of the fixed-point box (2^12 x the brick's largest upstream gradient: the
                        // normalisation of a nearly vanishing volume gradient amplifies by 1/|grad|, 1e7 and more where
                        // only rounding noise is left of it) a sample goes straight to the gradient volume with float
                        // atomics, exactly as in the baseline kernels and in the reference; NaN / absurd values are
                        // dropped or clamped.
                        bool in_box = valid;
                        float bnd = valid ? bound : 0.0f;
                        if (__any(valid && !(bound <= fs.lim))) {
                            const bool over = valid && bound > fs.lim && bound <= ACC_LIM;
                            if (__any(over)) {
                                const GradView dv = P.dvol;
                                float *gdst = dv.p + ((long long)view * P.dvol_vs + (long long)(c.ox + t.lx) * dv.sx +
                                                      (long long)(c.oy + t.ly) * dv.sy + (long long)(c.oz + t.lz) * dv.sz);
                                scatter_sample<ACC_GLOBAL>(gdst, (long long)dv.sx, (long long)dv.sy, (long long)dv.sz, t, over, I_bar, gq, fs);
                            }
                            in_box = valid && !over;
                            I_bar = fix_clamp(I_bar, fs);
                            gq[0] = fix_clamp(gq[0], fs); gq[1] = fix_clamp(gq[1], fs); gq[2] = fix_clamp(gq[2], fs);
                            bnd = in_box ? fabsf(I_bar) + (fabsf(gq[0]) + fabsf(gq[1]) + fabsf(gq[2])) : 0.0f;
                        }
                        // block floating point: the pass's addends share the exponent of its largest adjoint
                        // (a pass none of whose samples has an adjoint -- a stretch of the ray where the TF is transparent and
                        // flat: opacity 0, alpha slope 0 -- would add twenty zeros per lane: it adds nothing instead)
                        const float wmax = wave_max_nonneg(bnd);
                        if (wmax > 0.0f) {  // uniform
                            FixScale fw = fs;
                            const float mul = fix_wave_scale(wmax, fw);
                            const float gs[3] = {gq[0] * mul, gq[1] * mul, gq[2] * mul};
                            scatter_sample<ACC_FIX>(L.dbox, t, in_box, cbase_i, I_bar * mul, gs, fw);
                        }
                    }
                }
            }
        }
        if (!BWD) {
            // sample counts of the segments this wave owns (only this wave added to them)
            bool some = false;
            for (int e = ea + lane; e < eb; e += 64) {
                const int vv = L.valid[e];
                const int v = vv & VALID_CNT_MASK, tn = (vv >> VALID_TINY_SHIFT) & VALID_TINY_MASK;
                // (15 bits of count -- a longer in-brick run fails F2's count check and the ray is marched whole -- and the flag "some of
                //  them have a tiny opacity: their number is in seg_tiny")
                if (v > 0) {
                    P.seg_cnt[seg_view + L.segi[e]] = (uint16_t)(min(v, SEG_CNT_MAX) | (tn ? SEG_CNT_TINY : 0));
                    if (tn) P.seg_tiny[seg_view + L.segi[e]] = (uint16_t)tn;
                    some = true;
                    if (!skip_loop) n_eval_lane += (unsigned int)v;   // (an empty brick's segments are counted, not evaluated)
                }
            }
            if (!ALPHA && __any(some) && lane == 0)  // tell the backward that this brick holds live samples of the view
                *live_flag = 1;
        }
    }
    if (P.count_eval) {   // uniform (measurement only: bench.py's evaluated_voxel_steps)
        if (!BWD) { n_eval = n_eval_lane; for (int o = 32; o > 0; o >>= 1) n_eval += __shfl_xor(n_eval, o); }
        if (n_eval != 0u && lane == 0)
            atomicAdd(reinterpret_cast<unsigned long long *>(P.stats + (BWD ? ST_EVAL_BWD : (ALPHA ? ST_EVAL_PRE : ST_EVAL_FWD))), (unsigned long long)n_eval);
    }
    if (!BWD) return;
    if (!__syncthreads_or(any)) return;  // uniform; also: every wave's LDS adds are done before the flush
    // flush: one pass of global float atomics per brick, walking the gradient's fastest axis
    if (WANT_VOL) {
        GradView dv = P.dvol;
        dv.p += view * P.dvol_vs;
        const BoxWalk w = make_box_walk(dv.sx, dv.sy, dv.sz, vol.VX, vol.VY, vol.VZ, c);
        float *base = dv.p + w.base;
        const int a = threadIdx.x & 15, row = threadIdx.x >> 4;
        if (a < BOX) {
            for (int r = row; r < BOX * BOX; r += FNT / 16) {
                int b, d;
                box_row(r, b, d);
                const unsigned long long raw = L.dbox[a * w.la + b * w.lb + d * w.ld];
                if (raw != 0ull)  // in range whenever raw != 0
                    atomic_add_sat(base + (a * w.ga + b * w.gb + d * w.gd), acc64 ? acc_f64_to_float(raw) : fix_to_float(raw, fs));
            }
        }
    }
    if (WANT_TF) {   // (into the call's double table: dtf_commit_kernel hands the sums to the caller's tensor)
        for (int k = threadIdx.x; k < 4 * P.R; k += FNT) {
            const unsigned long long raw = L.dtf[k];
            if (raw != 0ull) dtf64_add(P.dtf64, view, k, raw);
        }
    }
}

// workgroups of the overflow launch (they loop over the items): what is resident at once on 256 CUs -- a workgroup that starts
// only after another has left would work off its first, statically assigned run of items at the very end
constexpr int ITEM_GRID_FWD = DR_ITEM_GRID_FWD, ITEM_GRID_BWD = DR_ITEM_GRID_BWD;
constexpr int ITEM_RUN = DR_ITEM_RUN;   // consecutive items a workgroup takes at a time

template <typename VT, int MODE, bool BWD, bool WANT_VOL, bool WANT_TF, int ALPHA = 0, int KF = 1, bool NARROW = true, bool TAPE = false>
__global__ __launch_bounds__((FlatCfg<BWD, WANT_VOL, ALPHA>::FNT), (FlatCfg<BWD, WANT_VOL, ALPHA>::WAVES)) void brick_flat_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    int bv = 0;
    brick_flat_body<VT, MODE, BWD, WANT_VOL, WANT_TF, ALPHA, KF, false, NARROW, TAPE>(P, smem, blockIdx.x, blockIdx.y,
                                                                                     0, MAIN_CAND, bv);
}
// the overflow items of heavy bricks (all views), worked off by a fixed, small grid
template <typename VT, int MODE, bool BWD, bool WANT_VOL, bool WANT_TF, int ALPHA = 0, int KF = 1, bool NARROW = true, bool TAPE = false>
__global__ __launch_bounds__((FlatCfg<BWD, WANT_VOL, ALPHA>::FNT), (FlatCfg<BWD, WANT_VOL, ALPHA>::WAVES)) void brick_flat_items_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    if (BWD && P.stats[ST_MARK] != P.mark) return;  // uniform over the grid: not this call's workspace, the item list is garbage
    const int n_items = min((int)*P.n_items, ITEM_CAP);
    if (blockIdx.x == 0 && threadIdx.x == 0) P.stats[ST_NITEMS] = *P.n_items;
    // Runs of ITEM_RUN consecutive items (consecutive items belong to one brick, brick_ctx_kernel) are handed out through a
    // ticket: the items differ widely in work, and within a run the brick's box stays staged.
    if (n_items == 0) return;  // uniform over the grid: nobody touches the ticket
    __shared__ int s_first;
    for (int round = 0;; ++round) {  // uniform
        // first run: by workgroup index; further runs: from the ticket (1024 returning atomics on one address take ~30 us)
        int first = blockIdx.x * ITEM_RUN;
        if (round > 0) {
            if (threadIdx.x == 0) s_first = (int)gridDim.x * ITEM_RUN + (int)atomicAdd(&P.stats[ST_TICKET], (unsigned int)ITEM_RUN);
            __syncthreads();
            first = s_first;
            __syncthreads();
        }
        if (first >= n_items) break;
        int box_valid = 0;
        int pv = -1, ps = -1;
        for (int it = first; it < min(first + ITEM_RUN, n_items); ++it) {
            const BrickItem item = P.items[it];
            if (item.view != pv || item.brick != ps) { box_valid = 0; pv = item.view; ps = item.brick; }
            brick_flat_body<VT, MODE, BWD, WANT_VOL, WANT_TF, ALPHA, KF, true, NARROW, TAPE>(P, smem, item.brick, item.view, item.c0, item.c1, box_valid);
            __syncthreads();  // the next item reuses the LDS
        }
    }
    // the last workgroup to leave resets the ticket for the next launch
    if (threadIdx.x == 0 && atomicAdd(&P.stats[ST_DONE], 1u) == gridDim.x - 1) {
        P.stats[ST_TICKET] = 0u; P.stats[ST_DONE] = 0u;
    }
}

// ------------------------------------------------------------------------------------------------ host
// This file is compiled TWICE (Makefile): as march_flat.o -- everything but the backward with a gradient box -- and, through
// march_flat_bwdvol.hip (which defines DR_FLAT_TU_BWDVOL and includes it), as the translation unit of that one kernel family,
// B1, built with -mllvm -amdgpu-sched-strategy=iterative-minreg: the scheduler strategy is a per-compilation switch, B1
// (128 VGPRs, its waves parked 40 % of their lifetime) gains 2.5 % from it, the forward and the TF-only backward lose 2-4 %
// (profiles/r04_ab_experiments.txt). Kernel templates are instantiated where they are launched, so each object holds its own.
#ifndef DR_FLAT_TU_BWDVOL
// in-box element offsets are computed in 32 bits: 3 * (BOX-1) * max|stride| must stay below 2^31
bool flat_strides_ok(int64_t sx, int64_t sy, int64_t sz) {
    const int64_t lim = ((int64_t)1 << 31) / (3 * BOX);
    return sx >= 0 && sy >= 0 && sz >= 0 && sx < lim && sy < lim && sz < lim;
}
bool brick_image_supported(int W, int H, int VX, int VY, int VZ) {
    const BrickGrid g = make_brick_grid(VX, VY, VZ);
    return (long long)g.NL * W * H < (1ll << 31);
}
bool brick_path_supported(int VX, int VY, int VZ, int R) {
    const int m = VX > VY ? (VX > VZ ? VX : VZ) : (VY > VZ ? VY : VZ);
    if (m - 1 >= 2000) return false;       // normal taps must stay within one voxel of the centre cell
    // The largest LDS request of the path is the backward's work-item kernel (its list of hits behind the regular layout). The
    // runtime does not grant the CU's full 160 KiB to one workgroup: 163 232 B launched, 163 616 B did not (R = 2040 / 2048,
    // tools/lds_limit_probe.py on the GPU box, round 5 -- until then R = 2041 .. 2223 passed this test and failed at the backward's
    // launch). 1 KiB of headroom: R <= 2030; a larger TF is served by the plain kernels, which have no limit.
    return align16(flat_lds_bytes<true>(R, true, true)) + ITEM_EXTRA_LDS + 64 <= 159 * 1024;
}

#endif  // !DR_FLAT_TU_BWDVOL

static inline bool taps_narrow(const MarchArgs &a) {
    const int m = a.VX > a.VY ? (a.VX > a.VZ ? a.VX : a.VZ) : (a.VY > a.VZ ? a.VY : a.VZ);
    return m - 1 <= 990;   // delta = 1e-3 world units = 1e-3 * (m - 1) / 2 voxels <= 0.495
}
// one pass over the bricks: the main launch (one workgroup per brick and view) + the overflow items of heavy bricks
#define DR_LAUNCH_BOTH_N(MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NT_, NARROW_)                                                             \
    {                                                                                                                                 \
        if ((e = allow_lds(brick_flat_kernel<VT, MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NARROW_>, lds)) != hipSuccess) return (int)e;    \
        if ((e = allow_lds(brick_flat_items_kernel<VT, MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NARROW_>, align16(lds) + ITEM_EXTRA_LDS)) != hipSuccess) return (int)e; \
        hipLaunchKernelGGL((brick_flat_kernel<VT, MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NARROW_>), grid1, dim3(NT_), lds, stream, P);         \
        hipLaunchKernelGGL((brick_flat_items_kernel<VT, MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NARROW_>), dim3(BWD_ ? (VOL_ ? ITEM_GRID_BWD : 2 * ITEM_GRID_BWD) : ITEM_GRID_FWD), dim3(NT_), align16(lds) + ITEM_EXTRA_LDS, stream, P); \
    }
// NARROW (delta below half a voxel: every edge <= 991 voxels) selects the cheaper shared-lerp taps (dr_brick_common.h); the alpha
// pre-pass takes no normal taps and exists in one flavour
#define DR_LAUNCH_BOTH(MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NT_)                                                                       \
    {                                                                                                                                 \
        if ((ALPHA_) || taps_narrow(a)) DR_LAUNCH_BOTH_N(MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NT_, true)                              \
        else DR_LAUNCH_BOTH_N(MODE_, BWD_, VOL_, TF_, ALPHA_, K_, NT_, false)                                                         \
    }

// B1 (the backward with a gradient box), launched from its own translation unit
template <typename VT>
int flat_bwd_vol_launch(const MarchArgs &a, BrickParams<VT> P, dim3 grid1, bool want_tf, hipStream_t stream);

#ifdef DR_FLAT_TU_BWDVOL
template <typename VT>
int flat_bwd_vol_launch(const MarchArgs &a, BrickParams<VT> P, dim3 grid1, bool want_tf, hipStream_t stream) {
    hipError_t e = hipSuccess;
    // (the LDS size is this translation unit's own: tuning / what-if switches may be given to it alone, tools/mkvariant.sh BWDVOL_EXTRA)
    const size_t lds = flat_lds_bytes<true>(a.R, true, want_tf);
    if (want_tf) DR_LAUNCH_BOTH(DR_MODE_DIFF, true, true, true, false, 1, (FlatCfg<true, true>::FNT))
    else DR_LAUNCH_BOTH(DR_MODE_DIFF, true, true, false, false, 1, (FlatCfg<true, true>::FNT))
    return (int)hipGetLastError();
}
template int flat_bwd_vol_launch<float>(const MarchArgs &, BrickParams<float>, dim3, bool, hipStream_t);
template int flat_bwd_vol_launch<__half>(const MarchArgs &, BrickParams<__half>, dim3, bool, hipStream_t);
#else

template <typename VT>
static int flat_fwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    // DR_TAPE_TF: a per-sample tape of (intensity, lighting) for the TF-only backward (tf_tape.hip) behind the ordinary workspace
    const bool tape = (a.hints & DR_TAPE_TF) && a.mode == DR_MODE_DIFF;
    const int tstride = tape ? tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S) : 0;
    const size_t need = ws_layout(a.workspace, a.n_views, NP, g, &w, tstride);
    if (!a.workspace || a.workspace_bytes < need) return DR_EINVAL;
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    P.tape = w.tape; P.tape_stride = tstride;
    hipError_t e;
    // Non-differentiable render with an alpha pre-pass: the pre-pass tells the colour march which (ray, layer) segments hold no
    // sample with alpha > 1e-3 ("unlit" masks behind seg_cnt, BrickParams::unlit); such segments keep the pre-pass's count and
    // zero partial, so seg_cnt is NOT cleared between the two passes (see below).
    // (not with a tape: a sample without opacity still has a TF gradient -- alpha = 0 has a slope -- and must leave its lighting term)
    const bool unlit_masks = !tape && (a.mode == DR_MODE_NONDIFF || DR_UNLIT_SKIP > 1) && !(a.hints & DR_HINT_NO_EARLY_TERMINATION) && w.lm_words > 0 && DR_UNLIT_SKIP;
    P.lm_words = unlit_masks ? w.lm_words : 0;
    // the item counter (brick_ctx_kernel appends) and seg_cnt behind it (and the masks behind that)
    e = hipMemsetAsync(w.n_items, 0, 16 + (unlit_masks ? align16(w.cnt_bytes) + w.unlit_bytes : w.cnt_bytes), stream);
    if (e != hipSuccess) return (int)e;
    const size_t lds = flat_lds_bytes<false>(a.R, false, false);
    const int nbricks = g.NBx * g.NBy * g.NBz;
    const dim3 grid1(nbricks, a.n_views);
    // Alpha pre-pass (early-termination culling): only if the TF can make some ray reach alpha >= 0.99 -- decided on
    // the device from max(alpha) by the first kernel (which also writes the brick records and resets the workspace
    // header); the kernels of the pre-pass return at once otherwise.
    // DR_HINT_NO_EARLY_TERMINATION: the caller knows the TF cannot make a ray opaque -- the (device-gated) launches of the
    // pre-pass are not issued at all. The device still evaluates may_terminate(); F2 repairs the view if the hint was wrong.
    const bool prepass = !(a.hints & DR_HINT_NO_EARLY_TERMINATION);
    double n_max = 0.0;
    {
        const double diag = sqrt((double)(a.VX - 1) * (a.VX - 1) + (double)(a.VY - 1) * (a.VY - 1) + (double)(a.VZ - 1) * (a.VZ - 1));
        n_max = floor((double)a.sr * 2.0 * sqrt(3.0) * diag) + 1.0;  // longest chord of the box (VR.py:251-253)
        if (a.mode == DR_MODE_DIFF && n_max > a.S) n_max = a.S;
        if (n_max < 1.0) n_max = 1.0;
    }
    hipLaunchKernelGGL(brick_ctx_kernel<VT>, dim3((nbricks + 255) / 256, a.n_views), dim3(256), 0, stream, P, w.ctx, nbricks, 1,
                       (float)n_max);
    // Samples per lane: the cross-lane scan and the chunk bookkeeping are paid once per K*64 samples, but lanes K
    // samples apart share fewer LDS words (more read cycles, more bank conflicts). Measured at 512^3: K = 2 wins up
    // at sampling rate 1 (-3 %), K = 4 from 2 on (-13 % at 2, -16 % at 4 and 8 vs K = 1: samples are closer together); K = 8 loses.
#define DR_LAUNCH_F1(MODE_, K_) DR_LAUNCH_BOTH(MODE_, false, false, false, false, K_, (FlatCfg<false, false>::FNT))
    const bool k_hi = a.sr >= 1.75f;
    if (prepass) {
        // The pre-pass runs front to back in G groups of brick layers; after each group the rays that have reached
        // alpha >= 0.99 are known and the next group does not march them (with the reference's tf1 preset three
        // quarters of all samples lie behind the termination point): ground-truth renders at sampling rate 8 take
        // 11.0 instead of 15.0 ms (8 views, 256^3). Each extra group costs the non-terminating case two empty
        // launches (~10 us: +0.7 % on the 512^3 headline at G = 2 for -2.5 % with tf1), so below sampling rate 3: G = 1.
        const int G = a.sr >= 3.0f ? DR_PP_GROUPS : ((a.hints & DR_HINT_EARLY_TERMINATION) ? DR_PP_GROUPS_LO : 1);
        MarchArgs pa = a;
        const bool alpha_hi = a.sr >= 3.0f;   // six workgroups per CU (FlatCfg<.., 2>)
        // (shadows the colour march's `lds` on purpose: the launch macros read that name)
        const size_t lds = flat_lds_bytes<false>(a.R, false, false, alpha_hi ? 2 : 1);   // the pre-pass's own table size
        for (int gi = 0; gi < G; ++gi) {
            pa.pp_l0 = g.NL * gi / G; pa.pp_l1 = g.NL * (gi + 1) / G; pa.pp_first = gi == 0;
            P.pp_l0 = pa.pp_l0; P.pp_l1 = pa.pp_l1; P.pp_first = pa.pp_first;
            // (several bricks per workgroup would quarter the cost of this launch when it is gated off -- 27 us of workgroup
            // exits at 512^3 -- but the looped kernel needs 96 VGPRs instead of 66 and is 3-5 % slower when it runs)
            if (a.mode == DR_MODE_DIFF) {
                if (alpha_hi) DR_LAUNCH_BOTH(DR_MODE_DIFF, false, false, false, 2, DR_ALPHA_K, (FlatCfg<false, false, 2>::FNT))
                else DR_LAUNCH_BOTH(DR_MODE_DIFF, false, false, false, 1, DR_ALPHA_K, (FlatCfg<false, false, 1>::FNT))
            } else {
                if (alpha_hi) DR_LAUNCH_BOTH(DR_MODE_NONDIFF, false, false, false, 2, DR_ALPHA_K, (FlatCfg<false, false, 2>::FNT))
                else DR_LAUNCH_BOTH(DR_MODE_NONDIFF, false, false, false, 1, DR_ALPHA_K, (FlatCfg<false, false, 1>::FNT))
            }
            if ((e = hipGetLastError()) != hipSuccess) return (int)e;
            const int rc = launch_ray_alpha(pa, stream);
            if (rc) return rc;
        }
        const int rc2 = launch_ray_cross(a, stream);
        if (rc2) return rc2;
        const size_t n16 = (w.cnt_bytes + 15) / 16;  // (seg_cnt is 16-byte aligned and padded)
        // With the unlit masks the counts stay: the colour march overwrites the count of every segment it marches; what it does
        // not touch is either an unlit segment in front of the ray's last live sample (count complete: a ray terminates on a
        // lit sample, and nondiff marches have no sample limit) or lies in a layer behind it, which F2 never reads.
        if (!unlit_masks)
        hipLaunchKernelGGL(clear_counts_if_prepass_kernel, dim3((unsigned)((n16 + 255) / 256 < 4096 ? (n16 + 255) / 256 : 4096)), dim3(256), 0,
                           stream, reinterpret_cast<uint4 *>(w.seg_cnt), n16, w.vflags, a.n_views);
    }
    MarchArgs b = a;
    b.use_live = prepass ? 1 : 0;
    P.use_live = b.use_live;
    if (tape) {
        // the same march with TAPE = true: main launch + work items, both tap flavours
#define DR_LAUNCH_F1_TAPE(K_, NARROW_)                                                                                                   \
        {                                                                                                                                \
            if ((e = allow_lds(brick_flat_kernel<VT, DR_MODE_DIFF, false, false, false, 0, K_, NARROW_, true>, lds)) != hipSuccess) return (int)e;   \
            if ((e = allow_lds(brick_flat_items_kernel<VT, DR_MODE_DIFF, false, false, false, 0, K_, NARROW_, true>, align16(lds) + ITEM_EXTRA_LDS)) != hipSuccess) return (int)e; \
            hipLaunchKernelGGL((brick_flat_kernel<VT, DR_MODE_DIFF, false, false, false, 0, K_, NARROW_, true>), grid1, dim3(FlatCfg<false, false>::FNT), lds, stream, P); \
            hipLaunchKernelGGL((brick_flat_items_kernel<VT, DR_MODE_DIFF, false, false, false, 0, K_, NARROW_, true>), dim3(ITEM_GRID_FWD), dim3(FlatCfg<false, false>::FNT), align16(lds) + ITEM_EXTRA_LDS, stream, P); \
        }
        const bool narrow = taps_narrow(a);
        if (k_hi) { if (narrow) DR_LAUNCH_F1_TAPE(DR_FWD_K_HI, true) else DR_LAUNCH_F1_TAPE(DR_FWD_K_HI, false) }
        else { if (narrow) DR_LAUNCH_F1_TAPE(DR_FWD_K, true) else DR_LAUNCH_F1_TAPE(DR_FWD_K, false) }
#undef DR_LAUNCH_F1_TAPE
    } else if (a.mode == DR_MODE_DIFF) { if (k_hi) DR_LAUNCH_F1(DR_MODE_DIFF, DR_FWD_K_HI) else DR_LAUNCH_F1(DR_MODE_DIFF, DR_FWD_K) }
    else { if (k_hi) DR_LAUNCH_F1(DR_MODE_NONDIFF, DR_FWD_K_HI) else DR_LAUNCH_F1(DR_MODE_NONDIFF, DR_FWD_K) }
#undef DR_LAUNCH_F1
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    const int rc3 = launch_ray_compose(b, stream);
    if (rc3) return rc3;
    return launch_ray_exact(b, stream);
}

hipError_t flat_invalidate_workspace(void *workspace, size_t workspace_bytes, hipStream_t stream) {
    if (!workspace || workspace_bytes < (size_t)ST_WORDS * 4) return hipSuccess;
    return hipMemsetAsync(static_cast<unsigned int *>(workspace) + ST_MARK, 0, 4, stream);
}

int launch_march_fwd_flat(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? flat_fwd_dispatch<__half>(a, stream) : flat_fwd_dispatch<float>(a, stream);
}

template <typename VT>
static int flat_bwd_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    Workspace w;
    const bool wv = a.d_vol != nullptr, wt = a.d_tf != nullptr;
    // DR_TAPE_TF: the TF-only backward as a per-ray pass over the forward's per-sample tape (tf_tape.hip), B2 for the flagged rays --
    // or for all of them, should the workspace turn out not to hold this call's tape (checked on the device)
    if ((a.hints & DR_TAPE_TF) && !wv && wt) {
        const int tstride = tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S);
        const size_t need_t = ws_layout(a.workspace, a.n_views, NP, g, &w, tstride);
        if (!a.workspace || a.workspace_bytes < need_t) return DR_EINVAL;
        const int rc = launch_tf_tape_bwd(a, stream);
        if (rc) return rc;
        MarchArgs b = a;
        b.only_flagged = w.rayflag;
        b.ws_mark = w.stats + ST_MARK; b.ws_mark_expect = ws_fingerprint(a);
        b.ws_aux = w.stats + ST_TAPE_STRIDE; b.ws_aux_expect = (unsigned int)tstride;
        const int rc2 = launch_march_bwd_baseline(b, stream);
        if (rc2) return rc2;
        const int rc3 = launch_ray_exact_bwd(a, stream);   // B3: the rays the forward recomputed sequentially
        if (rc3) return rc3;
        return launch_dtf_commit(a, stream);
    }
    const size_t need = ws_layout(a.workspace, a.n_views, NP, g, &w);
    if (!a.workspace || a.workspace_bytes < need) return DR_EINVAL;
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    P.lm_words = (wv && !wt && DR_UNLIT_SKIP > 1) ? w.lm_words : 0;   // the d_volume-only backward may skip unlit segments (if the forward left masks)
    const size_t lds = flat_lds_bytes<true>(a.R, wv, wt);
    const dim3 grid1(g.NBx * g.NBy * g.NBz, a.n_views);
    hipError_t e = hipSuccess;
    // (the brick records, live flags and work items are the forward's: same inputs, same workspace)
    if (wv) {   // B1: march_flat_bwdvol.o
        const int rc = flat_bwd_vol_launch<VT>(a, P, grid1, wt, stream);
        if (rc) return rc;
    } else DR_LAUNCH_BOTH(DR_MODE_DIFF, true, false, true, false, 1, (FlatCfg<true, false>::FNT))
    if ((e = hipGetLastError()) != hipSuccess) return (int)e;
    MarchArgs b = a;
    b.only_flagged = w.rayflag;  // B2: irregular rays through the baseline backward (every ray, if the workspace is not this call's)
    b.ws_mark = w.stats + ST_MARK; b.ws_mark_expect = P.mark;
    const int rc2 = launch_march_bwd_baseline(b, stream);
    if (rc2) return rc2;
    const int rc3 = launch_ray_exact_bwd(a, stream);   // B3: the rays the forward recomputed sequentially
    if (rc3 || !wt) return rc3;
    return launch_dtf_commit(a, stream);
}

int launch_march_bwd_flat(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? flat_bwd_dispatch<__half>(a, stream) : flat_bwd_dispatch<float>(a, stream);
}
#endif  // DR_FLAT_TU_BWDVOL

}  // namespace dr
