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
