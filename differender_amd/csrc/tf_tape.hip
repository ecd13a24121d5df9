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
//   One wave per ray, two consecutive samples per lane (128 samples per pass), the composite carried from pass to pass;
//   four rays per workgroup in flight, a resident grid per view striding over the rays; TF + d_tf table per workgroup, one flush.
// Rays the forward marched one by one (single-sample rays, repairs: rayflag) have no tape: the per-ray second pass (B2,
// march_baseline.hip) serves them, as after every fast backward -- and all rays, if the workspace does not hold this call's tape.
// Replaces, for d_tf: raycast.grad + get_final_image.grad (VR.py:460-461,470-471).
#include "dr_brick_common.h"
#include "dr_wave.h"
#include "dr_tuning.h"

namespace dr {


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
    for (int pl = blockIdx.x * 4 + (threadIdx.x >> 6); pl < NP; pl += 4 * (int)gridDim.x) {   // wave-uniform
        const size_t p = (size_t)view * NP + pl;
        const int live = P.ws_steps[p];
        const unsigned char rflag = P.rayflag[p];
        const float4 go = reinterpret_cast<const float4 *>(P.grad_out)[p];
        const float4 of = P.fin[p];
        if (rflag || live <= 0) continue;   // wave-uniform: no samples, or a ray of the per-ray pass
        any = true;
        const float2 *tp = P.tape + p * (size_t)P.tape_stride;
        const float gfin = go.x * of.x + go.y * of.y + go.z * of.z, afin = of.w;
        Over2 carry = {0.f, 0.f};   // composite of the samples before this pass, as (gC . C, A)
        for (int base = 0; base < live; base += 128) {   // uniform
            const int s0 = base + 2 * lane;
            // two consecutive samples of the lane: one 16-byte load (the stride is even, so is s0)
            float4 t4 = make_float4(0.f, 0.f, 0.f, 0.f);
            if (s0 < live) t4 = *reinterpret_cast<const float4 *>(tp + s0);
            const float kI[2] = {t4.x, t4.z}, kL[2] = {t4.y, t4.w};
            const bool kval[2] = {s0 < live, s0 + 1 < live};
            float kop[2], krd[2], kfr[2], ka[2];
            int klo[2], khi[2];
            Over2 el2 = {0.f, 0.f};
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                kop[j] = krd[j] = kfr[j] = ka[j] = 0.f; klo[j] = khi[j] = 0;
                Over2 ej = {0.f, 0.f};
                if (kval[j]) {
                    Sample sm;
                    sm.I = kI[j];
                    classify_from_I(lds_tf, P.R, P.tf_len, P.inv_sr, sm);
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
            // adjoints of the lane's samples, front to back (tape-free identity, as sample_adjoint of dr_device.h)
            float tv[2][4];
            {
                Over2 cur = exc2;   // composite before the sample
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    tv[j][0] = tv[j][1] = tv[j][2] = tv[j][3] = 0.f;
                    if (kval[j]) {
                        Over2 ej; ej.w = (kL[j] * kop[j]) * krd[j]; ej.a = kop[j];
                        const Over2 after = over2(cur, ej);
                        const float T = 1.0f - cur.a;   // transmittance before the sample
                        const bool last = (s0 + j == live - 1);
                        const float suffix = (gfin - after.w) + go.w * (afin - after.a);
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
            // d_tf: runs of consecutive samples between the same two texels are summed across lanes (DPP) and added once; a lane
            // whose two samples fall into different TF cells ("split") closes the incoming run with its first sample and opens a
            // new one with its second (the scheme of the brick-centric TF-only backward, march_flat.hip)
            const bool vA = kval[0], vB = kval[1];
            const bool split = vA && vB && klo[0] != klo[1];
            const int key_in = vA ? klo[0] : (vB ? klo[1] : -1 - lane);
            const int key_out = vB ? klo[1] : (vA ? klo[0] : -1 - lane);
            const int hi_out = vB ? khi[1] : khi[0];
            const int prev_out = wave_up1(key_out, key_out);
            const bool contl = lane != 0 && key_in == prev_out;   // continues the run of the lane before
            const bool start = !contl || split;                   // the lane's outgoing value starts a run
            float V[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) V[q] = split ? tv[1][q] : tv[0][q] + tv[1][q];
            const unsigned long long starts = __ballot(start);
            const unsigned long long upto = (lane == 63) ? ~0ull : ((2ull << lane) - 1ull);
            const int rs = 63 - __clzll((long long)(starts & upto));   // first lane of this lane's outgoing run
            seg_scan_sum<4>(V, lane, rs);
            const unsigned long long contm = __ballot(contl);
            const bool run_end = lane == 63 || !((contm >> ((lane + 1) & 63)) & 1ull);
            auto emit = [&](bool em, const float (&tq)[4], int lo, int hi) {
                const float v8[8] = {tq[0] * go.x, tq[0] * go.y, tq[0] * go.z, tq[2], tq[1] * go.x, tq[1] * go.y, tq[1] * go.z, tq[3]};
                const float vmax = ((fabsf(v8[0]) + fabsf(v8[1])) + (fabsf(v8[2]) + fabsf(v8[3]))) +
                                   ((fabsf(v8[4]) + fabsf(v8[5])) + (fabsf(v8[6]) + fabsf(v8[7])));
                unsigned long long *d0 = lds_dtf + 4 * lo, *d1 = lds_dtf + 4 * hi;
                if (__any(em && !(vmax <= ACC_LIM))) {   // a NaN or an absurd run total: the sanitising path
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
            if (__any(split)) {   // uniform
                // (the DPP moves are pinned in front of the select: `contl ? dpp : 0` otherwise becomes an exec-masked region around
                //  the v_mov_dpp, and a DPP move whose SOURCE lane is masked off writes nothing -- march_flat.hip, round 3)
                float Pv[4], H[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) Pv[q] = wave_up1(V[q], 0.f);   // inclusive sum of the previous lane's outgoing run
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" : "+v"(Pv[q]));
#pragma unroll
                for (int q = 0; q < 4; ++q) H[q] = (contl ? Pv[q] : 0.0f) + tv[0][q];
                emit(split, H, klo[0], khi[0]);
            }
        }
    }
    if (!__syncthreads_or(any)) return;   // uniform; also: every wave's LDS adds are done before the flush
    float *dtf = P.d_tf + view * P.dtf_vs * 4;
    for (int k = threadIdx.x; k < 4 * P.R; k += 256) {
        const unsigned long long raw = lds_dtf[k];
        if (raw != 0ull) atomic_add_sat(dtf + k, acc_f64_to_float(raw));
    }
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
