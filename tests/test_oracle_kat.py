"""Known-answer and finite-difference tests that pin the CPU oracle (oracle/dr_oracle.c).

The reference has no tests and cannot be imported, so these analytic facts -- each derived
from the reference's source text (VR.py = differender/volume_raycaster.py) -- are what the oracle is
pinned with ("parity unpinned" w.r.t. a running reference)."""
import math

import numpy as np
import pytest


def _scene(O, N=24, R=16, dtype=np.float64, alpha=0.05):
    vol = O.synth_volume(N, dtype=dtype)
    tf = O.bench_tf(R, alpha, dtype)
    cam = O.in_circles(0.3).astype(dtype)
    return vol, tf, cam


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_centre_ray_geometry(oracle, dtype):
    # camera on the +x axis: the centre ray (image centre falls between pixels; use odd-free check on t)
    # VR.py:233 view_dir = -cam/|cam|; slab test VR.py:41-50 gives tmin = d-1, tmax = d+1 on the axis.
    W = H = 9  # pixel 4 has centre (4.5/9) = 0.5 -> u = v = 0 -> ray == view_dir
    cam = np.array([3.0, 0.0, 0.0], dtype)
    e, x, r, n = oracle.ray_setup(cam, W, H, (32, 32, 32), sr=1.0, dtype=dtype)
    assert np.allclose(r[4, 4], [-1.0, 0.0, 0.0], atol=1e-6)
    assert abs(e[4, 4] - 2.0) < 1e-5 and abs(x[4, 4] - 4.0) < 1e-5
    diag = math.sqrt(3 * 31 ** 2)
    assert n[4, 4] == int(math.floor(1.0 * 2.0 * diag)) + 1  # VR.py:251-253
    # sampling_rate scales the count
    _, _, _, n2 = oracle.ray_setup(cam, W, H, (32, 32, 32), sr=2.5, dtype=dtype)
    assert abs(int(n2[4, 4]) - (int(math.floor(2.5 * 2.0 * diag)) + 1)) <= 1


def test_fov_is_tan_fov_not_half(oracle):
    # VR.py:146: near_h = 2*tan(fov)*near (not fov/2): top pixel row direction has tan(angle) = (v*2*tan(fov))
    W = H = 64
    cam = np.array([0.0, 0.0, 4.0], np.float64)
    e, x, r, n = oracle.ray_setup(cam, W, H, (16, 16, 16), dtype=np.float64)
    j = H - 1
    v = (j + 0.5) / H - 0.5
    d = r[W // 2, j]
    # ray = normalize(near*vd + u*near_w*right + v*near_h*up); here up = +y, vd = -z
    assert abs(d[1] / -d[2] - v * 2 * math.tan(math.radians(30.0))) < 1e-9


def test_miss_is_zero(oracle):
    vol, tf, cam = _scene(oracle)
    # camera far off-axis looking at origin with tiny volume coverage: corners of a wide image miss
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (48, 48))
    miss = n == 0
    assert miss.any() and (~miss).any()
    assert np.all(rgba[miss] == 0.0) and np.all(steps[miss] == 0)


def test_trilinear_reproduces_affine_volume(oracle):
    # VR.py:153-189: trilinear sampling of V = a + b.x is exact up to the 1e-4 scale fudge (VR.py:165)
    N = 20
    ax = np.linspace(0.0, 1.0, N)
    X, Y, Z = np.meshgrid(ax, ax, ax, indexing="ij")
    vol = 0.1 + 0.3 * X + 0.2 * Y + 0.25 * Z
    R = 64
    # TF whose red channel is the identity ramp and alpha constant: colour encodes the sampled intensity
    tf = np.zeros((R, 4)); tf[:, 0] = np.linspace(0, 1, R); tf[:, 3] = 1.0  # opaque: first sample decides
    cam = np.array([0.3, 0.2, 3.0])
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (8, 8))
    hit = n > 1
    # first sample position (VR.py:273-280, s=0): t0 = entry + 0.5*len/n
    t0 = e + 0.5 * (x - e) / np.maximum(n, 1)
    pos = cam[None, None, :] + t0[..., None] * r
    p01 = np.clip(0.5 * pos + 0.5, 0, 1) * (N - 1 - 1e-4) / (N - 1)
    expect_I = 0.1 + 0.3 * p01[..., 0] + 0.2 * p01[..., 1] + 0.25 * p01[..., 2]
    # opaque TF => op = 1 => C = L * I, A = 1; the normal of an affine field is constant: g/|g|
    g = np.array([0.3, 0.2, 0.25]); nrm = g / np.linalg.norm(g)
    light = cam + np.array([0.0, 1.0, 0.0])
    ld = pos - light; ld /= np.linalg.norm(ld, axis=-1, keepdims=True)
    m = ld @ nrm
    rf = ld - 2 * m[..., None] * nrm
    rdv = np.maximum(-(rf * r).sum(-1), 0)
    L = np.minimum(1.0, 0.8 * np.maximum(m, 0) + 0.3 * rdv ** 32 + 0.4)
    assert np.allclose(rgba[..., 3][hit], 1.0)
    assert np.allclose(rgba[..., 0][hit], (L * expect_I)[hit], atol=2e-6)
    assert np.all(steps[hit] == 1)  # early termination after the first opaque sample (VR.py:267)


def test_constant_alpha_accumulation(oracle):
    # constant-alpha TF: A_final = 1 - (1-o)^m with m marched samples (VR.py:300-302), o = 1-(1-a)^(1/sr)
    vol, tf, cam = _scene(oracle, alpha=0.01)
    for sr in (1.0, 2.0):
        rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (12, 12), sr=sr)
        o = 1.0 - (1.0 - 0.01) ** (1.0 / sr)
        expect = 1.0 - (1.0 - o) ** steps
        assert np.allclose(rgba[..., 3], expect, atol=1e-9)
        assert np.all(steps == n)  # no early termination with this alpha


def test_early_termination_and_max_samples(oracle):
    vol, tf, cam = _scene(oracle, alpha=0.2)
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (12, 12))
    hit = n > 30
    assert np.all(steps[hit] < n[hit])
    assert np.all(rgba[..., 3][hit] >= 0.99)
    # A just before the last marched sample was < 0.99
    o = 0.2
    assert np.allclose(rgba[..., 3][hit], 1 - (1 - o) ** steps[hit], atol=1e-9)
    assert np.all(1 - (1 - o) ** (steps[hit] - 1) < 0.99)
    # H2: max_samples bounds the marched samples of the differentiable path only
    vol, tf, cam = _scene(oracle, alpha=0.001)
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (12, 12), S=7)
    assert steps.max() == 7
    rgba1, steps1, _ = oracle.render(vol, tf, cam, (12, 12), S=7, mode=1)
    assert np.all(steps1 == n)


def test_nondiff_skips_low_alpha_and_clamps(oracle):
    vol, tf, cam = _scene(oracle)
    tf0 = tf.copy(); tf0[:, 3] = 1e-3  # not > 1e-3 -> every sample skipped (VR.py:334)
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf0, cam, (12, 12), mode=1)
    assert np.all(rgba == 0.0) and np.all(steps == n)
    tf1 = tf.copy(); tf1[:, 3] = 0.5; tf1[:, :3] = 1.0
    rgba, _, _ = oracle.render(vol, tf1, cam, (12, 12), mode=1)
    assert rgba.max() <= 1.0  # VR.py:358


def test_tf_lerp_at_texel_centres(oracle):
    # constant volume with intensity exactly on texel k/(R-1) -> TF value tf[k] (VR.py:215-219);
    # flat volume => normal undefined => ambient-only lighting L = 0.4
    R, k = 11, 4
    vol = np.full((8, 8, 8), k / (R - 1))
    tf = np.zeros((R, 4)); tf[k] = [0.9, 0.5, 0.25, 1.0]
    cam = np.array([0.0, 0.0, 3.0])
    rgba, steps, (e, x, r, n) = oracle.render(vol, tf, cam, (4, 4))
    hit = n > 0
    assert np.allclose(rgba[hit], [0.4 * 0.9, 0.4 * 0.5, 0.4 * 0.25, 1.0], atol=1e-12)


def test_jitter_is_deterministic_and_bounded(oracle):
    cam = oracle.in_circles(1.0)
    e0, x0, r0, n0 = oracle.ray_setup(cam, 16, 16, (32, 32, 32))
    e1, x1, r1, n1 = oracle.ray_setup(cam, 16, 16, (32, 32, 32), jitter_seed=7)
    e2, x2, r2, n2 = oracle.ray_setup(cam, 16, 16, (32, 32, 32), jitter_seed=7)
    e3, _, _, _ = oracle.ray_setup(cam, 16, 16, (32, 32, 32), jitter_seed=8)
    hit = n0 > 0
    assert np.array_equal(e1[hit], e2[hit]) and np.array_equal(n0, n1) and np.array_equal(x0, x1)
    assert not np.array_equal(e1[hit], e3[hit])
    u = (e1 - e0)[hit] / ((x0 - e0)[hit] / n0[hit])  # VR.py:255: tmin += U*len/n
    assert u.min() >= -1e-4 and u.max() < 1.0 + 1e-4 and 0.3 < u.mean() < 0.7


@pytest.mark.parametrize("sr", [1.0, 2.0])
def test_backward_matches_finite_differences_f64(oracle, sr):
    """The hand-derived adjoint against central differences of the f64 forward."""
    N, R, W = 14, 8, 10
    vol = oracle.synth_volume(N, dtype=np.float64)
    rng = np.random.RandomState(3)
    tf = rng.uniform(0.05, 0.6, size=(R, 4)); tf[:, 3] *= 0.3
    cam = oracle.in_circles(0.7).astype(np.float64)
    e, x, r, n = oracle.ray_setup(cam, W, W, vol.shape, sr=sr, dtype=np.float64)
    g = rng.randn(W, W, 4)
    S = 1 << 20
    dv, dtf = oracle.march_bwd(vol, tf, cam, e, x, r, n, S, sr, g)

    def loss(v, t):
        out, _ = oracle.march_fwd(v, t, cam, e, x, r, n, S, sr, 0)
        return float((out * g).sum())

    eps = 1e-6
    for idx in [(2, 0), (4, 3), (6, 1), (7, 3)]:
        tp, tm = tf.copy(), tf.copy(); tp[idx] += eps; tm[idx] -= eps
        fd = (loss(vol, tp) - loss(vol, tm)) / (2 * eps)
        assert abs(fd - dtf[idx]) <= 1e-6 * max(1.0, abs(fd))
    flat = np.argsort(-np.abs(dv).ravel())
    for fi in list(flat[:6]) + list(flat[200:1200:250]):
        idx = np.unravel_index(fi, vol.shape)
        vp, vm = vol.copy(), vol.copy(); vp[idx] += eps; vm[idx] -= eps
        fd = (loss(vp, tf) - loss(vm, tf)) / (2 * eps)
        assert abs(fd - dv[idx]) <= 2e-6 * max(1.0, abs(fd)), (idx, fd, dv[idx])


def test_backward_with_early_termination_f64(oracle):
    N, R, W = 14, 8, 8
    vol = oracle.synth_volume(N, dtype=np.float64)
    tf = oracle.peaks_tf(R, np.float64); tf[:, 3] = 0.25
    cam = oracle.in_circles(2.1).astype(np.float64)
    e, x, r, n = oracle.ray_setup(cam, W, W, vol.shape, dtype=np.float64)
    rng = np.random.RandomState(5)
    g = rng.randn(W, W, 4)
    dv, dtf = oracle.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g)
    out, steps = oracle.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, 0)
    assert (steps < n)[n > 20].all()

    def loss(t):
        o, _ = oracle.march_fwd(vol, t, cam, e, x, r, n, 1 << 20, 1.0, 0)
        return float((o * g).sum())

    eps = 1e-7
    for idx in [(3, 3), (4, 0), (5, 3)]:
        tp, tm = tf.copy(), tf.copy(); tp[idx] += eps; tm[idx] -= eps
        fd = (loss(tp) - loss(tm)) / (2 * eps)
        assert abs(fd - dtf[idx]) <= 1e-5 * max(1.0, abs(fd))


def test_f32_oracle_tracks_f64(oracle):
    vol, tf, cam = _scene(oracle, N=32, R=32, dtype=np.float64, alpha=0.02)
    e, x, r, n = oracle.ray_setup(cam, 24, 24, vol.shape, dtype=np.float64)
    o64, _ = oracle.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, 0)
    f = np.float32
    o32, _ = oracle.march_fwd(vol.astype(f), tf.astype(f), cam.astype(f), e.astype(f), x.astype(f), r.astype(f), n,
                              1 << 20, 1.0, 0)
    assert np.abs(o32 - o64).max() < 5e-5


def test_epilogue_loss_and_momentum_step(oracle):
    O = oracle
    """oracle epilogue vs. a float64 numpy statement of EX.py:368-381 / torch mse_loss."""
    rng = np.random.default_rng(5)
    out = rng.random((3, 8, 8, 4), dtype=np.float32)
    ref = rng.random((3, 8, 8, 4), dtype=np.float32)
    loss, grad = O.mse_loss_grad(out, ref)
    d = out.astype(np.float64) - ref
    assert abs(loss - (d * d).mean()) < 1e-7 * loss      # d and 1/n are rounded to f32
    np.testing.assert_allclose(grad, 2 * d / d.size, rtol=3e-7, atol=0)
    loss3, _ = O.mse_loss_grad(out[0], ref[0], inv_norm=1.0 / (3 * 64))   # EX.py:369-373 normalisation
    assert abs(loss3 - (d[0] * d[0]).sum() / (3 * 64)) < 1e-7 * loss3

    tf = rng.random((16, 4), dtype=np.float32) * 0.1
    g = (rng.standard_normal((16, 4)) * 3).astype(np.float32)
    mom = (rng.standard_normal((16, 4)) * 0.01).astype(np.float32)
    tf2, mom2 = O.tf_momentum_step(tf, g, mom, lr=0.05, gamma=0.9, max_grad=1.0)
    m_ref = 0.9 * mom.astype(np.float64) + 0.05 * np.clip(g.astype(np.float64), -1, 1)
    np.testing.assert_allclose(mom2, m_ref, rtol=1e-6, atol=1e-9)
    np.testing.assert_allclose(tf2, np.maximum(tf - m_ref, 0), rtol=1e-6, atol=1e-8)
    assert (tf2 >= 0).all() and (tf2 == 0).any()      # the clamp at zero is exercised


def test_specified_power_is_within_one_ulp(oracle):
    """(1 - a)^(1/sr), the one function oracle and kernels share bit for bit (oracle/dr_oracle.c, dr_device.h): against the
    double-precision power it must stay within 1 ulp at the square-root rates and be the correctly rounded float elsewhere."""
    import ctypes
    f = oracle.lib().dro_pow_inv_sr
    f.restype = ctypes.c_float; f.argtypes = [ctypes.c_float, ctypes.c_float]
    rng = np.random.default_rng(5)
    xs = np.concatenate([rng.random(4000, dtype=np.float32), 1 - rng.random(4000, dtype=np.float32) * np.float32(0.01),
                         rng.random(500, dtype=np.float32) * np.float32(1e-6), np.array([1.0, 0.5, 0.25, 1e-30, 1e-38], np.float32)])
    for sr in (2.0, 4.0, 8.0, 16.0, 0.3, 0.6, 1.5, 3.0, 5.0, 12.0):
        y = np.float32(1.0) / np.float32(sr)
        got = np.array([f(float(x), float(y)) for x in xs], np.float32)
        ref = np.power(xs.astype(np.float64), np.float64(y))
        ulp = np.spacing(ref.astype(np.float32)).astype(np.float64)
        err = np.abs(got.astype(np.float64) - ref) / ulp
        if sr in (2.0, 4.0, 8.0, 16.0):
            assert err.max() < 1.0, (sr, err.max())
        else:
            assert err.max() <= 0.5 + 1e-6, (sr, err.max())          # correctly rounded on this sample
            assert np.array_equal(got, ref.astype(np.float32))
    assert f(0.0, 0.3) == 0.0 and f(1.0, 0.3) == 1.0 and math.isnan(f(-0.1, 0.3))
    assert f(0.7, 1.0) == np.float32(0.7)


def test_ray_parallel_to_a_slab_it_misses_has_no_samples(oracle):
    """Both slab distances are the same infinity, the reference's hit test passes and ray_len = inf - inf: the sample count is
    NaN cast to int -- 0 on a GPU (CUDA cvt.rzi, AMD v_cvt_i32_f32), undefined in C. The oracle defines it as the GPUs do."""
    cam = np.array([3.5491052, -17.622072, 35.73327], np.float32)   # the fuzzer's find (seed 1116, view 1)
    e, x, r, n = oracle.ray_setup(cam, 47, 93, (3, 3, 95), sr=8.0)
    assert n.min() >= 0
    bad = ~np.isfinite(x - e)
    assert bad.any() and (n[bad] == 0).all()
