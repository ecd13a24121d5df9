/*
 * dr_oracle.c -- CPU oracle for the Differender volume_raycaster hot path (see dr_oracle_impl.inc).
 * TEST INFRASTRUCTURE ONLY (tests/, __graft_entry__.smoke(), bench.py cpu_baseline). PARITY UNPINNED.
 * Exports dro_*_f32 (parity oracle) and dro_*_f64 (finite-difference reference).
 * Build: make -C oracle   (gcc -O2 -ffp-contract=off -fopenmp -shared)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <stddef.h>

/* Opacity correction 1 - (1 - a)^(1/sr) (VR.py:285): `ti.pow` is CUDA's approximate __powf in the reference (Taichi's
 * fast_math default), i.e. not a pinned function, and at small a the result is ill-conditioned in f32: one ulp of the
 * power is a relative 1e-4 of the opacity at sr = 16, and whether a ray crosses alpha 0.99 at sample s or s + 1 can hinge
 * on it. Oracle and kernels therefore share one SPECIFIED single-precision power, reproducible bit for bit on any IEEE
 * platform (differender_amd/csrc/dr_device.h: pow_inv_sr, pow_spec -- written independently from this description):
 *   1/sr = 2^-k (sampling rates 2, 4, 8, 16): k correctly rounded square roots;
 *   otherwise exp2(y * log2(x)) in double precision from +, -, *, / only, rounded once to float:
 *     x = m * 2^e with m in [1/sqrt2, sqrt2];  z = (m-1)/(m+1);  ln m = z * (2 + 2/3 z^2 + ... + 2/13 z^12)  (Horner);
 *     t = y * (e + ln m * log2(e));  n = rint(t);  r = (t - n) * ln 2;  exp r = 1 + r + ... + r^11/11!  (Horner);
 *     result = (float)(exp r * 2^n).
 * Both are within 1 ulp of x^y (tests/test_oracle_kat.py checks that against the double-precision pow). */
static float dro_pow_spec_f32(float x, float y) {
    union { double d; int64_t i; } u;
    if (!(x > 0.0f)) return (x == 0.0f) ? 0.0f : NAN;
    u.d = (double)x;
    int e = (int)((u.i >> 52) & 0x7ff) - 1023;
    u.i = (u.i & INT64_C(0x000fffffffffffff)) | INT64_C(0x3ff0000000000000);
    double m = u.d;
    if (m > 1.4142135623730951) { m *= 0.5; e += 1; }
    const double z = (m - 1.0) / (m + 1.0), z2 = z * z;
    double p = 2.0 / 13.0;
    p = p * z2 + 2.0 / 11.0; p = p * z2 + 2.0 / 9.0; p = p * z2 + 2.0 / 7.0; p = p * z2 + 2.0 / 5.0;
    p = p * z2 + 2.0 / 3.0; p = p * z2 + 2.0;
    const double t = (double)y * ((double)e + (p * z) * 1.4426950408889634);
    if (!(t > -160.0)) return 0.0f;
    if (!(t < 128.0)) return INFINITY;
    const double n = rint(t), r = (t - n) * 0.6931471805599453;
    double q = 1.0 / 39916800.0;
    q = q * r + 1.0 / 3628800.0; q = q * r + 1.0 / 362880.0; q = q * r + 1.0 / 40320.0; q = q * r + 1.0 / 5040.0;
    q = q * r + 1.0 / 720.0; q = q * r + 1.0 / 120.0; q = q * r + 1.0 / 24.0; q = q * r + 1.0 / 6.0;
    q = q * r + 0.5; q = q * r + 1.0; q = q * r + 1.0;
    u.i = (int64_t)((int)n + 1023) << 52;
    return (float)(q * u.d);
}
static float dro_pow_inv_sr_f32(float base, float inv_sr) {
    if (inv_sr == 1.0f) return base;
    if (inv_sr == 0.5f) return sqrtf(base);
    if (inv_sr == 0.25f) return sqrtf(sqrtf(base));
    if (inv_sr == 0.125f) return sqrtf(sqrtf(sqrtf(base)));
    if (inv_sr == 0.0625f) return sqrtf(sqrtf(sqrtf(sqrtf(base))));
    return dro_pow_spec_f32(base, inv_sr);
}
float dro_pow_inv_sr(float base, float inv_sr) { return dro_pow_inv_sr_f32(base, inv_sr); }  /* exported for the KAT */

#define REAL float
#define SUF(x) x##_f32
#define R_SQRT sqrtf
#define R_FLOOR floorf
#define R_POW powf
#define R_POW_INV_SR dro_pow_inv_sr_f32
#define R_FMAX fmaxf
#define R_FMIN fminf
#define R_FMA fmaf
#include "dr_oracle_impl.inc"
#undef REAL
#undef SUF
#undef R_SQRT
#undef R_FLOOR
#undef R_POW
#undef R_POW_INV_SR
#undef R_FMAX
#undef R_FMIN
#undef R_FMA

#define REAL double
#define SUF(x) x##_f64
#define R_SQRT sqrt
#define R_FLOOR floor
#define R_POW pow
#define R_POW_INV_SR pow
#define R_FMAX fmax
#define R_FMIN fmin
#define R_FMA fma
#include "dr_oracle_impl.inc"

int dro_abi_version(void) { return 1; }

/* ---- loss / optimiser epilogue (f32 only; the f64 check is a numpy one-liner in the tests) ---- */

/* compute_loss (examples/taichi_volume_raycaster.py:368-373) and the mse_loss round trip of :439-443:
 * grad = (out - ref) * (2*inv_norm);  returns inv_norm * sum (out - ref)^2 (sum carried in double). */
double dro_mse_loss_grad_f32(const float *out, const float *ref, int64_t n, float inv_norm, float *grad) {
    double acc = 0.0;
    float two_inv = 2.0f * inv_norm;
    for (int64_t i = 0; i < n; ++i) {
        float d = out[i] - ref[i];
        if (grad) grad[i] = d * two_inv;
        acc += (double)d * (double)d;
    }
    return acc * (double)inv_norm;
}

/* apply_grad (examples/taichi_volume_raycaster.py:375-381): momentum step with gradient clipping, tf >= 0. */
void dro_tf_momentum_step_f32(float *tf, const float *g, float *mom, int n, float lr, float gamma, float max_grad) {
    for (int i = 0; i < n; ++i) {
        float c = fminf(max_grad, fmaxf(-max_grad, g[i]));
        float m = gamma * mom[i] + lr * c;
        mom[i] = m;
        tf[i] = fmaxf(tf[i] - m, 0.0f);
    }
}

/* Checks the identity the fast kernels rely on for sample positions (VR.py:279-280, s/(n-1)): with y = RN(1/d),
 * q0 = RN(s*y), r = fma(-q0, d, s), RN(q0 + r*y) equals the IEEE quotient s/d. Exhaustive over d <= dmax
 * (all 0 <= s <= d+3), plus `nrandom` pseudo-random pairs with d < 8e6. Returns the number of mismatches. */
long dro_check_rcp_division(int dmax, long nrandom) {
    long bad = 0;
    for (int d = 1; d <= dmax; ++d) {
        const float df = (float)d, y = 1.0f / df;
        for (int s = 0; s <= d + 3; ++s) {
            const float sf = (float)s;
            float q = sf * y;
            const float r = fmaf(-q, df, sf);
            q = fmaf(r, y, q);
            bad += (q != sf / df);
        }
    }
    uint64_t st = 0x9E3779B97F4A7C15ull;
    for (long k = 0; k < nrandom; ++k) {
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int d = 1 + (int)((st >> 33) % 8000000u);
        st = st * 6364136223846793005ull + 1442695040888963407ull;
        const int s = (int)((st >> 33) % (uint64_t)(d + 1));
        const float df = (float)d, y = 1.0f / df, sf = (float)s;
        float q = sf * y;
        const float r = fmaf(-q, df, sf);
        q = fmaf(r, y, q);
        bad += (q != sf / df);
    }
    return bad;
}
