#!/usr/bin/env python3
"""Independent witness for ray setup and the non-differentiable march: tests/golden/setup_nondiff_*.npz.

A float64 numpy TRANSLITERATION, statement by statement and vectorised over the pixels, of
  * `compute_entry_exit` with its helpers `get_ray_direction` and `get_entry_exit_points`
    (differender/volume_raycaster.py:221-259, :127-151, :28-53), jitter off (`ti.random` cannot be reproduced), and
  * `raycast_nondiff` + `get_final_image_nondiff` (:308-361) with the helpers :7-21, :153-219,
written from the reference's text. It shares no code with oracle/ (C) or with the HIP kernels, and the march takes its
ray buffers from THIS file's ray setup, not from the oracle's: the oracle is no longer its own witness for these
functions (tests/test_setup_nondiff_golden.py compares the f64 oracle to 1e-12 / 1e-9, the f32 oracle and the HIP path
within the parity tolerances). Still "parity unpinned" w.r.t. a running reference -- Taichi is not installed.

taichi_glsl semantics assumed (from memory): mix(x, y, a) = x (1 - a) + y a; reflect(I, N) = I - 2 dot(N, I) N;
normalized() = v / |v|; min / max NaN-suppressing.

Pixels the comparison must leave out are flagged in the fixture (`ok_setup`, `ok_march`):
  * a sample count whose floor() argument lies within 1e-6 of an integer (f32 and f64 may legitimately differ by one);
  * single-sample rays (0/0 position, VR.py:279-280 / SURVEY H6);
  * rays whose accumulated alpha passes within 1e-5 of the 0.99 threshold (the f32 march may stop one sample apart).

Run:  python tests/golden/make_setup_nondiff_golden.py
"""
import math
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))


def normalized(v):
    return v / np.sqrt((v * v).sum(-1, keepdims=True))


def mix(x, y, a):
    return x * (1.0 - a) + y * a


# ---- VR.py:127-151
def get_ray_direction(orig, view_dir, x, y, fov_rad, near, aspect):
    u = x - 0.5
    v = y - 0.5
    up = np.array([0.0, 1.0, 0.0])
    right = normalized(np.cross(view_dir, up))
    up = normalized(np.cross(right, view_dir))
    near_h = 2.0 * math.tan(fov_rad) * near
    near_w = near_h * aspect
    near_m = orig + near * view_dir
    near_pos = near_m + u[..., None] * near_w * right + v[..., None] * near_h * up
    return normalized(near_pos - orig)


# ---- VR.py:28-53
def get_entry_exit_points(look_from, view_dir, bl, tr):
    dirfrac = 1.0 / view_dir
    t1 = (bl[0] - look_from[0]) * dirfrac[..., 0]
    t2 = (tr[0] - look_from[0]) * dirfrac[..., 0]
    t3 = (bl[1] - look_from[1]) * dirfrac[..., 1]
    t4 = (tr[1] - look_from[1]) * dirfrac[..., 1]
    t5 = (bl[2] - look_from[2]) * dirfrac[..., 2]
    t6 = (tr[2] - look_from[2]) * dirfrac[..., 2]
    tmin = np.maximum(np.maximum(np.minimum(t1, t2), np.minimum(t3, t4)), np.minimum(t5, t6))
    tmax = np.minimum(np.minimum(np.maximum(t1, t2), np.maximum(t3, t4)), np.maximum(t5, t6))
    hit = ~((tmax < 0.0) | (tmin > tmax))
    return tmin, tmax, hit


# ---- VR.py:221-259 (jitter = 0)
def compute_entry_exit(cam_pos, W, H, vol_shape, sampling_rate, fov_deg=30.0, near=0.1):
    fov_rad = np.radians(fov_deg)             # VR.py:77
    aspect = W / H                            # VR.py:75
    i, j = np.meshgrid(np.arange(W), np.arange(H), indexing="ij")
    max_x, max_y = float(W), float(H)
    look_from = np.asarray(cam_pos, np.float64)
    view_dir = normalized(-look_from)
    bb_bl = np.array([-1.0, -1.0, -1.0])
    bb_tr = np.array([1.0, 1.0, 1.0])
    x = (i.astype(np.float64) + 0.5) / max_x
    y = (j.astype(np.float64) + 0.5) / max_y
    vd = get_ray_direction(look_from, view_dir, x, y, fov_rad, near, aspect)
    tmin, tmax, hit = get_entry_exit_points(look_from, vd, bb_bl, bb_tr)
    vol_diag = float(np.sqrt(((np.asarray(vol_shape, np.float64) - 1.0) ** 2).sum()))
    ray_len = tmax - tmin
    arg = sampling_rate * ray_len * vol_diag
    n_samples = hit * (np.floor(arg) + 1.0)
    # sample counts that hinge on the rounding of `arg`
    unambiguous = ~hit | (np.abs(arg - np.round(arg)) > 1e-6 * np.maximum(np.abs(arg), 1.0))
    return tmin, tmax, vd, n_samples.astype(np.int32), unambiguous


# ---- VR.py:7-21
def low_high_frac(x):
    x = np.maximum(x, 0.0)
    low = np.floor(x)
    return low.astype(np.int64), low.astype(np.int64) + 1, x - low


# ---- VR.py:153-189
def sample_volume_trilinear(vol, pos):
    shape = np.asarray(vol.shape, np.float64)
    p = np.clip(0.5 * pos + 0.5, 0.0, 1.0) * (shape - 1.0 - 1e-4)
    xl, xh, xf = low_high_frac(p[:, 0])
    yl, yh, yf = low_high_frac(p[:, 1])
    zl, zh, zf = low_high_frac(p[:, 2])
    xh = np.minimum(xh, vol.shape[0] - 1); yh = np.minimum(yh, vol.shape[1] - 1); zh = np.minimum(zh, vol.shape[2] - 1)
    a = mix(vol[xl, yl, zl], vol[xh, yl, zl], xf)
    b = mix(vol[xl, yh, zl], vol[xh, yh, zl], xf)
    z_low = mix(a, b, yf)
    a = mix(vol[xl, yl, zh], vol[xh, yl, zh], xf)
    b = mix(vol[xl, yh, zh], vol[xh, yh, zh], xf)
    z_high = mix(a, b, yf)
    return mix(z_low, z_high, zf)


# ---- VR.py:191-203
def get_volume_normal(vol, pos):
    delta = 1e-3
    ex, ey, ez = np.array([delta, 0.0, 0.0]), np.array([0.0, delta, 0.0]), np.array([0.0, 0.0, delta])
    dx = sample_volume_trilinear(vol, pos + ex) - sample_volume_trilinear(vol, pos - ex)
    dy = sample_volume_trilinear(vol, pos + ey) - sample_volume_trilinear(vol, pos - ey)
    dz = sample_volume_trilinear(vol, pos + ez) - sample_volume_trilinear(vol, pos - ez)
    g = np.stack([dx, dy, dz], axis=1)
    norm = np.sqrt((g * g).sum(1, keepdims=True))
    flat = norm[:, 0] == 0.0
    # .normalized() of a vanishing gradient is NaN in the reference; its NaN-suppressing max() then yields n.l = r.v = 0
    # (DESIGN.md D1): the sample is lit by the ambient term only
    return np.where(flat[:, None], 0.0, g / np.where(norm == 0.0, 1.0, norm)), flat


# ---- VR.py:205-219
def apply_transfer_function(tf, intensity):
    R = tf.shape[0]
    lo, hi, fr = low_high_frac(intensity * float(R - 1))
    hi = np.minimum(hi, R - 1)
    lo = np.minimum(lo, R - 1)        # defined-domain guard for intensity > 1 (the reference reads out of bounds)
    return mix(tf[lo], tf[hi], fr[:, None])


# ---- VR.py:308-361
def raycast_nondiff(vol, tf, cam, entry, exit_, rays, n, sampling_rate):
    """All pixels at once. Returns output_rgba (P,4) = min(1, tape[..., 0]), the number of iterations each ray spent
    below alpha 0.99 (the kernels' `steps`), and the closest approach of the accumulated alpha to 0.99."""
    P = entry.shape[0]
    tape0 = np.zeros((P, 4))                       # render_tape[i, j, 0], cleared by clear_framebuffer
    live = np.zeros(P, np.int64)
    closest = np.full(P, np.inf)
    ambient, diffuse_k, specular_k, shininess = 0.4, 0.8, 0.3, 32.0
    light_pos = cam + np.array([0.0, 1.0, 0.0])
    nf = n.astype(np.float64)
    for cnt in range(int(n.max())):
        active = (cnt < n) & (tape0[:, 3] < 0.99)                                   # :317-318
        if not active.any():
            continue
        live += active
        ray_len = exit_ - entry
        tmin = entry + 0.5 * ray_len / np.where(nf > 0, nf, 1.0)
        frac = np.where(n > 1, float(cnt) / np.maximum(nf - 1.0, 1.0), 0.0)
        pos = cam[None, :] + mix(tmin, exit_, frac)[:, None] * rays
        pos = np.where(active[:, None], pos, 0.0)
        intensity = sample_volume_trilinear(vol, pos)
        sample_color = apply_transfer_function(tf, intensity)
        opacity = 1.0 - np.power(1.0 - sample_color[:, 3], 1.0 / sampling_rate)      # :332-333
        shade = active & (sample_color[:, 3] > 1e-3)                                # :334
        normal, flat = get_volume_normal(vol, pos)
        ld = pos - light_pos[None, :]
        light_dir = ld / np.sqrt((ld * ld).sum(1, keepdims=True))
        n_dot_l = np.maximum((normal * light_dir).sum(1), 0.0)
        diffuse = diffuse_k * n_dot_l
        r = light_dir - 2.0 * (normal * light_dir).sum(1, keepdims=True) * normal
        r_dot_v = np.where(flat, 0.0, np.maximum((r * (-rays)).sum(1), 0.0))
        specular = specular_k * np.power(r_dot_v, shininess)
        lighting = diffuse + specular + ambient                                     # NOT clamped (:345)
        shaded = np.concatenate([(lighting * opacity)[:, None] * sample_color[:, :3], opacity[:, None]], axis=1)
        new = (1.0 - tape0[:, 3:4]) * shaded + tape0
        tape0 = np.where(shade[:, None], new, tape0)
        closest = np.where(active, np.minimum(closest, np.abs(tape0[:, 3] - 0.99)), closest)
    return np.minimum(1.0, tape0), live, closest                                    # :358


def synth_volume(shape, seed):
    """Smooth blobs + ramp + a little noise, values in (0, 1) (this file's own generator)."""
    rng = np.random.RandomState(seed)
    ax = [np.linspace(-1.0, 1.0, s) for s in shape]
    X, Y, Z = np.meshgrid(*ax, indexing="ij")
    v = 0.25 + 0.02 * (X + 2 * Y - Z)
    for _ in range(5):
        c = rng.uniform(-0.6, 0.6, 3); s = rng.uniform(0.2, 0.5)
        v += rng.uniform(0.15, 0.4) * np.exp(-((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) / (2 * s * s))
    v += 0.01 * rng.standard_normal(shape)
    return np.clip(v, 0.02, 0.98)


CASES = {
    # name: (volume shape, (W, H), R, sampling rate, TF kind, camera)
    "a_orbit_sr1": ((16, 16, 16), (16, 16), 8, 1.0, "peaks", (2.5 * math.cos(0.8), 0.7, 2.5 * math.sin(0.8))),
    "b_aniso_sr4": ((20, 12, 28), (24, 16), 16, 4.0, "opaque", (-1.9, 1.1, 1.4)),
    "c_far_sr03": ((12, 24, 16), (16, 24), 12, 0.3, "peaks", (0.3, -2.2, 3.9)),
    "d_inside_sr2": ((16, 20, 16), (16, 16), 32, 2.0, "thin", (0.31, 0.12, -0.27)),
    "e_sr8_ert": ((24, 24, 24), (16, 16), 24, 8.0, "opaque", (2.5 * math.cos(4.0), 0.7, 2.5 * math.sin(4.0))),
}


def make_tf(R, kind, rng):
    tf = rng.uniform(0.05, 0.95, size=(R, 4))
    i = np.linspace(0.0, 1.0, R)
    if kind == "thin":
        tf[:, 3] = np.linspace(0.01, 0.06, R)
    elif kind == "opaque":
        tf[:, 3] = i ** 2 * 0.9 + 0.05
    else:   # peaks with empty stretches (alpha <= 1e-3 is skipped by the nondiff march)
        tf[:, 3] = 0.5 * np.exp(-((i - 0.55) / 0.12) ** 2) + 0.1 * np.exp(-((i - 0.8) / 0.05) ** 2)
        tf[i < 0.3, 3] = 0.0
    return tf


def make_case(name):
    vshape, (W, H), R, sr, kind, cam = CASES[name]
    seed = sorted(CASES).index(name) + 11
    rng = np.random.RandomState(seed)
    vol = synth_volume(vshape, seed)
    tf = make_tf(R, kind, rng)
    cam = np.asarray(cam, np.float64)
    entry, exit_, rays, n, ok_setup = compute_entry_exit(cam, W, H, vshape, sr)
    rgba, live, closest = raycast_nondiff(vol, tf, cam, entry.reshape(-1), exit_.reshape(-1), rays.reshape(-1, 3),
                                          n.reshape(-1).astype(np.int64), sr)
    ok_march = (n.reshape(-1) != 1) & (closest > 1e-5)
    return dict(vol=vol, tf=tf, cam=cam, sr=np.float64(sr), entry=entry, exit=exit_, rays=rays, n=n, ok_setup=ok_setup,
                rgba=rgba.reshape(W, H, 4), live=live.reshape(W, H).astype(np.int32), ok_march=ok_march.reshape(W, H))


def main():
    for name in CASES:
        c = make_case(name)
        path = os.path.join(HERE, f"setup_nondiff_{name}.npz")
        np.savez_compressed(path, **c)
        hit = c["n"] > 0
        print(f"wrote {path} ({os.path.getsize(path)} B): {int(hit.sum())}/{hit.size} rays hit, samples {int(c['n'].sum())}, "
              f"live {int(c['live'].sum())}, setup-ambiguous {int((~c['ok_setup']).sum())}, march-masked {int((~c['ok_march']).sum())}, "
              f"max alpha {c['rgba'][..., 3].max():.4f}, entry min {np.nanmin(np.where(hit, c['entry'], np.nan)):.3f}")


if __name__ == "__main__":
    main()
