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

#define REAL float
#define SUF(x) x##_f32
#define R_SQRT sqrtf
#define R_FLOOR floorf
#define R_POW powf
#define R_FMAX fmaxf
#define R_FMIN fminf
#define R_FMA fmaf
#include "dr_oracle_impl.inc"
#undef REAL
#undef SUF
#undef R_SQRT
#undef R_FLOOR
#undef R_POW
#undef R_FMAX
#undef R_FMIN
#undef R_FMA

#define REAL double
#define SUF(x) x##_f64
#define R_SQRT sqrt
#define R_FLOOR floor
#define R_POW pow
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
