"""Randomised parity (tools/fuzz_parity.py) as a test: the seeds that exposed defects when the fuzzer first ran, plus a
block of fresh ones. Every case: ray setup and step counts bit-exact, RGBA within 1e-5, gradients within 1e-4 of the
tensor's largest magnitude (or, in ill-conditioned cases, within 3 x the baseline kernels' own distance from the oracle --
or, where the f32 oracle itself is more than 1 % of the maximum away from the same backward in float64, within 1 % of THAT distance).

What the listed seeds pinned down ("What the fuzzer found"):
  1, 603, 2552, 2571   adjoints beyond the range of the fixed-point LDS box (normalisation of a nearly vanishing gradient)
  28, 113, 1246        sampling rates 8 and 16: (1 - a)^(1/sr) must be the same function in oracle and kernels
  759, 1310, 2430      sampling rates 0.3 and 3: likewise for exponents that are not 2^-k
  826, 177             contributions far below the brick's largest upstream gradient (block-floating-point addends)
  20228, 23671, 26708  a sample on a kink of the lighting model (Lraw within 1e-5 of the clamp at 1): D7
  1108, 1116, 2440     a ray parallel to a slab it lies outside of: NaN sample count, defined as 0
  56, 90, 3255         alpha == 1 at a sampling rate != 1: infinite reference gradient, kernels stay finite
  603239               (round 4) one TF texel, sampling rate 16, the SEQUENTIAL kernels: their f32 LDS atomics put 1e-3 of summation noise on
                       d_tf (the oracle sums d_tf in double); they accumulate in double now
"""
import importlib.util
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REGRESSION_SEEDS = [1, 28, 56, 90, 113, 177, 603, 759, 826, 1108, 1116, 1246, 1310, 2430, 2440, 2552, 2571, 3255,
                    20228, 21150, 23557, 23646, 23671, 26708, 603239]


@pytest.fixture(scope="module")
def fuzz(hiplib, oracle):
    import torch
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    spec = importlib.util.spec_from_file_location("fuzz_parity", os.path.join(ROOT, "tools", "fuzz_parity.py"))
    mod = importlib.util.module_from_spec(spec)
    argv, sys.argv = sys.argv, sys.argv[:1]
    try:
        spec.loader.exec_module(mod)
    finally:
        sys.argv = argv
    return mod


@pytest.mark.parametrize("seed", REGRESSION_SEEDS)
def test_fuzz_regression_seed(fuzz, seed):
    case = fuzz.make_case(seed)
    fails = fuzz.run_case(case)
    assert not fails, (fails, fuzz.describe(case))


def test_fuzz_block_of_fresh_seeds(fuzz):
    bad, illcond, n = [], [], 0
    for seed in range(10000, 10120):
        case = fuzz.make_case(seed)
        fails = fuzz.run_case(case)
        n += 1
        if fails:
            bad.append((seed, fails, fuzz.describe(case)))
        if case.get("_illcond"):
            illcond.append(seed)
    assert not bad, bad
    # The "ill-conditioned" branch of the fuzzer (a gradient judged against the baseline kernels' own distance from the
    # oracle instead of the flat 1e-4) must stay the exception: a regression must not be able to hide there.
    assert len(illcond) <= 0.02 * n, f"{len(illcond)} of {n} cases took the ill-conditioned branch: seeds {illcond}"


def test_fuzz_block_under_the_d4_profile(fuzz, monkeypatch):
    """FUZZ_PROFILE=d4 (transparent ranges at alpha 1e-6 around an opaque spike, sampling rates 2-8, every kind of volume): sub-ulp
    contributions behind opaque structures. Until round 6 this profile had an escape hatch (up to 5e-5 from the float32 oracle
    where float64 sided with the fast path); it is gone -- every case holds the ordinary bars, termination decisions included.
    4000502 is the noise volume at rate 8 that sat at 1.09e-5."""
    monkeypatch.setattr(fuzz, "PROFILE", "d4")
    bad = []
    for seed in [4000502] + list(range(4000000, 4000090)):
        case = fuzz.make_case(seed)
        fails = fuzz.run_case(case)
        if fails:
            bad.append((seed, fails, fuzz.describe(case)))
    assert not bad, bad


def test_fuzz_scale3_ten_thousand_sample_rays(fuzz, monkeypatch):
    """FUZZ_SCALE=3 seed 90320 (round 6): a (185, 288, 82) volume rendered from INSIDE at sampling rate 16, non-differentiable -- rays
    of 10 000-12 000 samples. The reference's sequential float32 alpha is a random walk of roundings around the true value, 0.29 ulp per
    sample: after 10 600 samples it stood 3.5e-6 above the re-associated alpha of the crossing search and crossed 0.99 one sample
    early, outside the search's fixed 2e-6 band for "repeat the decision sequentially". The band grows with sqrt(samples) now
    (ray_passes.hip: cross_band). Plus a few neighbours of the seed."""
    monkeypatch.setattr(fuzz, "SCALE", 3)
    bad = []
    for seed in (90320, 90321, 90322):
        case = fuzz.make_case(seed)
        fails = fuzz.run_case(case)
        if fails:
            bad.append((seed, fails, fuzz.describe(case)))
    assert not bad, bad
