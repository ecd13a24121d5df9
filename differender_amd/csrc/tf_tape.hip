// tf_tape.hip -- the backward w.r.t. the transfer function ALONE (BASELINE config C3; the reference's TF optimisation) as a
// per-RAY pass over a per-sample tape (DR_TAPE_TF, gfx950).
//
// What the TF gradient needs of a sample from the volume is two numbers: the intensity I (which TF texels, which fraction) and
// the lighting term L. The brick-centric TF-only backward (march_flat.hip, B1-TF) re-derives them -- position, cell, seven
// trilinear taps from a re-staged brick, normalisations, Phong -- to throw everything else away: 2.9 ms for the 2.1 ms forward
// it re-marches. Under DR_TAPE_TF the forward leaves (I, L) of every marched sample on a tape, [view][pixel][tape_stride] float2,
// and the adjoint becomes a one-dimensional problem per ray: TF lookup (LDS), a wave scan of the composite as (gC . C, A), the
// tape-free identity of SURVEY 8(a)-bwd, run sums of d_tf over lanes that share a texel pair, double-precision LDS atomics. No
// brick, no listing, no tap; the loads are 16 contiguous bytes per lane.
//   One wave per ray, four consecutive samples per lane (256 samples per pass; the last 128 of a ray: two per lane), the composite
//   carried from pass to pass; four rays per workgroup in flight, a resident grid per view striding over the rays; TF + d_tf table
//   per workgroup, one flush.
// Rays the forward marched one by one (single-sample rays, repairs: rayflag) have no tape: the per-ray second pass (B2,
// march_baseline.hip) serves them, as after every fast backward -- and all rays, if the workspace does not hold this call's tape.
// Replaces, for d_tf: raycast.grad + get_final_image.grad (VR.py:460-461,470-471).
#include "dr_brick_common.h"
#include "dr_wave.h"
#include "dr_tuning.h"

namespace dr {


// What a pass needs of its workgroup and its ray.
struct TapeRay {
    const float4 *lds_tf;          // the view's TF
    unsigned long long *lds_dtf;   // its gradient, 64-bit fixed point (dr_device.h: acc_add_f64)
    const float2 *tp;              // the ray's tape
    float4 go;                     // d loss / d pixel
    float gfin, afin;              // the final composite as (gC . C, A)
    int live;                      // samples the forward marched
    int R, tf_len;
    float inv_sr;
};

// adds a finished run's sums tq = (w0 L op T, w1 L op T, w0 a_bar, w1 a_bar) to the texels lo / hi
__device__ __forceinline__ void tape_emit(const TapeRay &r, bool em, const float (&tq)[4], int lo, int hi) {
    const float4 go = r.go;
    const float v8[8] = {tq[0] * go.x, tq[0] * go.y, tq[0] * go.z, tq[2], tq[1] * go.x, tq[1] * go.y, tq[1] * go.z, tq[3]};
    const float vmax = ((fabsf(v8[0]) + fabsf(v8[1])) + (fabsf(v8[2]) + fabsf(v8[3]))) +
                       ((fabsf(v8[4]) + fabsf(v8[5])) + (fabsf(v8[6]) + fabsf(v8[7])));
    unsigned long long *d0 = r.lds_dtf + 4 * lo, *d1 = r.lds_dtf + 4 * hi;
    if (__any(em && !(vmax <= ACC_LIM))) {   // a NaN or an absurd run total: the sanitising path
        if (em) {
#pragma unroll
            for (int q = 0; q < 4; ++q) { acc_add_f64(d0 + q, acc_sanitise(v8[q])); acc_add_f64(d1 + q, acc_sanitise(v8[4 + q])); }
        }
    } else if (em) {
#pragma unroll
        for (int q = 0; q < 4; ++q) { acc_add_f64(d0 + q, v8[q]); acc_add_f64(d1 + q, v8[4 + q]); }
    }
}

// One pass of a wave over 64 * K consecutive samples of its ray, K per lane, from sample `base`; `carry` is the composite of the
// samples before the pass as (gC . C, A) and leaves as the composite including them. What is paid once per pass -- the scan of the
// composite, the run sums of d_tf, their ballots and the LDS adds -- is shared by K samples per lane: the kernel runs passes of
// four while more than 128 samples are left and ends a ray on a pass of two.
template <int K>
__device__ __forceinline__ void tape_pass(const TapeRay &r, const int base, const int lane, Over2 &carry) {
    static_assert(K == 2 || K == 4, "16-byte loads of two samples");
    const int s0 = base + K * lane;
    const float4 go = r.go;
    // K consecutive samples of the lane: 16-byte loads (the stride is even, so is s0; the second load of a lane whose third sample
    // exists stays inside the ray's slot of the tape: its stride is even and at least `live`)
    float kI[K], kL[K];
#pragma unroll
    for (int h = 0; h < K / 2; ++h) {
        float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
        if (s0 + 2 * h < r.live) t4 = *reinterpret_cast<const float4 *>(r.tp + s0 + 2 * h);
        kI[2 * h] = t4.x; kL[2 * h] = t4.y; kI[2 * h + 1] = t4.z; kL[2 * h + 1] = t4.w;
    }
    float kop[K], krd[K], kfr[K], ka[K];
    int klo[K], khi[K];
    Over2 el2 = {0.f, 0.f};
#pragma unroll
    for (int j = 0; j < K; ++j) {
        kop[j] = krd[j] = kfr[j] = ka[j] = 0.f; klo[j] = khi[j] = 0;
        Over2 ej = {0.f, 0.f};
        if (s0 + j < r.live) {
            Sample sm;
            sm.I = kI[j];
            classify_from_I(r.lds_tf, r.R, r.tf_len, r.inv_sr, sm);
            const float rd = go.x * sm.r + go.y * sm.g + go.z * sm.b;
            kop[j] = sm.op; krd[j] = rd; kfr[j] = sm.fr; ka[j] = sm.a; klo[j] = sm.lo; khi[j] = sm.hi;
            ej.w = (kL[j] * sm.op) * rd; ej.a = sm.op;
        }
        el2 = (j == 0) ? ej : over2(el2, ej);   // over2(x, 0) == x exactly
    }
    // inclusive wave scan of the lanes' composites (one segment: the ray), then the carry from the passes before
    Over2 inc2 = seg_scan_over2(el2, lane, 0), exc2;
    exc2.w = wave_up1(inc2.w, 0.f); exc2.a = wave_up1(inc2.a, 0.f);
    if (lane == 0) { exc2.w = 0.f; exc2.a = 0.f; }
    inc2 = over2(carry, inc2); exc2 = over2(carry, exc2);
    carry.w = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.w), 63));
    carry.a = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inc2.a), 63));
    // Adjoints of the lane's samples, front to back (tape-free identity, as sample_adjoint of dr_device.h), gathered into RUNS of
    // consecutive samples between the same two texels: run sums are added to d_tf once. A lane's samples form up to K runs: the
    // first continues the run of the lane before (if that ended in the same TF cell), the last is carried on to the next lane
    // (summed across lanes by DPP below); a lane whose samples fall into different TF cells ("split") closes the incoming run
    // with its first group -- and a group strictly inside a lane (K = 4: three cells within four samples) is added where it ends.
    // (the scheme of the brick-centric TF-only backward, march_flat.hip)
    const bool v0 = s0 < r.live;
    int key_in = v0 ? klo[0] : -1 - lane, hi_in = khi[0];
    int key_out = key_in, hi_out = hi_in;
    bool split = false;
    float V[4] = {0.f, 0.f, 0.f, 0.f};    // sums of the lane's last group
    float Hd[4] = {0.f, 0.f, 0.f, 0.f};   // ... of its first, once there are two
    {
        Over2 cur = exc2;   // composite before the sample
#pragma unroll
        for (int j = 0; j < K; ++j) {
            if (s0 + j < r.live) {
                Over2 ej; ej.w = (kL[j] * kop[j]) * krd[j]; ej.a = kop[j];
                const Over2 after = over2(cur, ej);
                const float T = 1.0f - cur.a;   // transmittance before the sample
                const bool last = (s0 + j == r.live - 1);
                const float suffix = (r.gfin - after.w) + go.w * (r.afin - after.a);
                const float qs = kL[j] * krd[j] + go.w;
                const float sfx = suffix * __builtin_amdgcn_rcpf(1.0f - kop[j]);
                const float op_bar = T * qs - (last ? 0.0f : sfx);
                const float Lop = kL[j] * kop[j] * T;
                const float a_bar = op_bar * ((r.inv_sr == 1.0f) ? 1.0f : r.inv_sr * powf(1.0f - ka[j], r.inv_sr - 1.0f));
                const float w0 = 1.0f - kfr[j], w1 = kfr[j];
                const float t[4] = {w0 * Lop, w1 * Lop, w0 * a_bar, w1 * a_bar};
                cur = after;
                if (j == 0 || klo[j] == key_out) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) V[q] += t[q];
                } else {
                    if (!split) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) Hd[q] = V[q];
                        split = true;
                    } else if (K > 2) {
                        tape_emit(r, true, V, key_out, hi_out);   // (among the lanes that are here)
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) V[q] = t[q];
                    key_out = klo[j]; hi_out = khi[j];
                }
            }
        }
    }
    const int prev_out = wave_up1(key_out, key_out);
    const bool contl = lane != 0 && key_in == prev_out;   // continues the run of the lane before
    const bool start = !contl || split;                   // the lane's outgoing value starts a run
    const unsigned long long starts = __ballot(start);
    const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
    const int rs = 63 - __clzll((long long)(starts & upto));   // first lane of this lane's outgoing run
    seg_scan_sum<4>(V, lane, rs);
    const unsigned long long contm = __ballot(contl);
    const bool run_end = lane == 63 || !((contm >> ((lane + 1) & 63)) & 1ull);
    tape_emit(r, v0 && run_end, V, key_out, hi_out);
    if (__any(split)) {   // uniform
        // (the DPP moves are pinned in front of the select: `contl ? dpp : 0` otherwise becomes an exec-masked region around
        //  the v_mov_dpp, and a DPP move whose SOURCE lane is masked off writes nothing -- march_flat.hip, round 3)
        float Pv[4], H[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) Pv[q] = wave_up1(V[q], 0.f);   // inclusive sum of the previous lane's outgoing run
#pragma unroll
        for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(Pv[q]));
#pragma unroll
        for (int q = 0; q < 4; ++q) H[q] = (contl ? Pv[q] : 0.0f) + Hd[q];
        tape_emit(r, split, H, key_in, hi_in);
    }
}

template <typename VT>
__global__ __launch_bounds__(256) void tf_tape_bwd_kernel(BrickParams<VT> P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float4 *lds_tf = reinterpret_cast<float4 *>(smem);
    unsigned long long *lds_dtf = reinterpret_cast<unsigned long long *>(smem + (size_t)P.R * 16);
    // the forward of THIS call left its tape here? (else: nothing -- B2 marches every ray)
    if (P.stats[ST_MARK] != P.mark || P.stats[ST_TAPE_STRIDE] != (unsigned int)P.tape_stride) return;   // uniform over the grid
    const int view = blockIdx.y;
    const float4 *tfg = P.tf + view * P.tf_vs;
    for (int k = threadIdx.x; k < P.R; k += 256) lds_tf[k] = tfg[k];
    for (int k = threadIdx.x; k < 4 * P.R; k += 256) lds_dtf[k] = 0ull;
    __syncthreads();
    const int NP = P.W * P.H;
    const int lane = threadIdx.x & 63;
    bool any = false;
    TapeRay r;
    r.lds_tf = lds_tf; r.lds_dtf = lds_dtf; r.R = P.R; r.tf_len = P.tf_len; r.inv_sr = P.inv_sr;
    for (int pl = blockIdx.x * 4 + (threadIdx.x >> 6); pl < NP; pl += 4 * (int)gridDim.x) {   // wave-uniform
        const size_t p = (size_t)view * NP + pl;
        const int live = P.ws_steps[p];
        const unsigned char rflag = P.rayflag[p];
        const float4 go = reinterpret_cast<const float4 *>(P.grad_out)[p];
        const float4 of = P.fin[p];
        if (rflag || live <= 0) continue;   // wave-uniform: no samples, or a ray of the per-ray pass
        any = true;
        r.tp = P.tape + p * (size_t)P.tape_stride;
        r.go = go; r.live = live;
        r.gfin = go.x * of.x + go.y * of.y + go.z * of.z; r.afin = of.w;
        Over2 carry = {0.f, 0.f};   // composite of the samples before this pass, as (gC . C, A)
        int base = 0;
        for (; live - base > 128; base += 256) tape_pass<4>(r, base, lane, carry);   // uniform
        if (base < live) tape_pass<2>(r, base, lane, carry);
    }
    if (!__syncthreads_or(any)) return;   // uniform; also: every wave's LDS adds are done before the flush
    for (int k = threadIdx.x; k < 4 * P.R; k += 256) {   // (into the call's double table: dtf_commit_kernel)
        const unsigned long long raw = lds_dtf[k];
        if (raw != 0ull) dtf64_add(P.dtf64, view, k, raw);
    }
}

// The double d_tf table of this backward call -> the caller's float tensor (added: the caller zero-filled it, the per-ray pass B2 may have
// put its few rays' share there already), clamped into the finite floats; the table is zero again afterwards. Only if
// the workspace holds this call's forward (else no kernel has added anything and the table is not ours to read).
template <typename VT>
__global__ __launch_bounds__(256) void dtf_commit_kernel(BrickParams<VT> P) {
    if (P.stats[ST_MARK] != P.mark) return;   // uniform
    if (P.tape_stride > 0 && P.stats[ST_TAPE_STRIDE] != (unsigned int)P.tape_stride) return;
    const int k = blockIdx.x * 256 + threadIdx.x, view = blockIdx.y;
    if (k >= 4 * P.R) return;
    double *t = P.dtf64 + (size_t)view * 4 * DTF64_R + k;
    const double v = *t;
    if (v != 0.0) {
        *t = 0.0;
        atomic_add_sat(P.d_tf + view * P.dtf_vs * 4 + k, fminf(fmaxf((float)v, -3.0e38f), 3.0e38f));
    }
}
template <typename VT>
static int dtf_commit_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    Workspace w;
    ws_layout(a.workspace, a.n_views, a.W * a.H, g, &w);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    if ((a.hints & DR_TAPE_TF) && !a.d_vol) P.tape_stride = tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S);
    hipLaunchKernelGGL((dtf_commit_kernel<VT>), dim3((4 * a.R + 255) / 256, a.n_views), dim3(256), 0, stream, P);
    return (int)hipGetLastError();
}
int launch_dtf_commit(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? dtf_commit_dispatch<__half>(a, stream) : dtf_commit_dispatch<float>(a, stream);
}

template <typename VT>
static int tf_tape_dispatch(const MarchArgs &a, hipStream_t stream) {
    const BrickGrid g = make_brick_grid(a.VX, a.VY, a.VZ);
    const int NP = a.W * a.H;
    const int tstride = tape_stride_for(a.VX, a.VY, a.VZ, a.sr, a.S);
    Workspace w;
    ws_layout(a.workspace, a.n_views, NP, g, &w, tstride);
    BrickParams<VT> P = make_brick_params<VT>(a, w);
    P.tape = w.tape; P.tape_stride = tstride;
    const size_t lds = (size_t)a.R * 48;
    const hipError_t e = allow_lds(tf_tape_bwd_kernel<VT>, lds);
    if (e != hipSuccess) return (int)e;
    int gx = (DR_TAPE_GRID + a.n_views - 1) / a.n_views;
    if (gx > (NP + 3) / 4) gx = (NP + 3) / 4;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((tf_tape_bwd_kernel<VT>), dim3(gx, a.n_views), dim3(256), lds, stream, P);
    return (int)hipGetLastError();
}

int launch_tf_tape_bwd(const MarchArgs &a, hipStream_t stream) {
    return a.vol_dtype == DR_F16 ? tf_tape_dispatch<__half>(a, stream) : tf_tape_dispatch<float>(a, stream);
}

}  // namespace dr
