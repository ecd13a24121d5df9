"""ctypes/numpy front-end of the CPU oracle (oracle/dr_oracle.c).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg, never by differender_amd. PARITY UNPINNED (the reference has no golden vectors and cannot be
imported here; see DESIGN.md section "Oracle").

Array conventions (same as the C file): volume (VX, VY, VZ) = the reference's field index (i, j, k)
(VR.py:481: (W, D, H) of the user's (1, D, H, W) tensor); tf (R, 4); image buffers (W, H[, C]).
"""
import ctypes
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libdr_oracle.so")
_lib = None


def build(force=False):
    """Compile the oracle with gcc (no GPU toolchain involved)."""
    src = [os.path.join(_HERE, f) for f in ("dr_oracle.c", "dr_oracle_impl.inc")]
    if (not force and os.path.exists(_LIB_PATH)
            and all(os.path.getmtime(_LIB_PATH) >= os.path.getmtime(s) for s in src)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-B", "libdr_oracle.so"], stdout=subprocess.DEVNULL)
    return _LIB_PATH


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = ctypes.CDLL(_LIB_PATH)
    return _lib


def _ct(dt):
    return ctypes.c_float if dt == np.float32 else ctypes.c_double


def _p(a):
    return a.ctypes.data_as(ctypes.c_void_p) if a is not None else None


def _suffix(dt):
    return "_f32" if np.dtype(dt) == np.float32 else "_f64"


def ray_setup(cam, W, H, vol_shape, sr=1.0, fov_deg=30.0, near=0.1, jitter_seed=0, view=0, dtype=np.float32):
    """VR.py:221-259. Returns entry (W,H), exit (W,H), rays (W,H,3), n (W,H) int32."""
    dt = np.dtype(dtype)
    cam = np.ascontiguousarray(cam, dtype=dt)
    entry = np.empty((W, H), dt); exit_ = np.empty((W, H), dt)
    rays = np.empty((W, H, 3), dt); n = np.empty((W, H), np.int32)
    fn = getattr(lib(), "dro_ray_setup" + _suffix(dt))
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                   ctypes.c_double, ctypes.c_double, _ct(dt), ctypes.c_uint32, ctypes.c_uint32,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(_p(cam), W, H, *map(int, vol_shape), float(np.radians(fov_deg)), float(near), float(sr),
            int(jitter_seed), int(view), _p(entry), _p(exit_), _p(rays), _p(n))
    assert rc == 0
    return entry, exit_, rays, n


def march_fwd(vol, tf, cam, entry, exit_, rays, n, S, sr=1.0, mode=0):
    """mode 0: VR.py:261-306 + 363-372 (differentiable path); mode 1: VR.py:308-361 (nondiff).
    Returns rgba (W,H,4), steps (W,H) int32."""
    dt = vol.dtype
    vol = np.ascontiguousarray(vol); tf = np.ascontiguousarray(tf, dtype=dt)
    cam = np.ascontiguousarray(cam, dtype=dt)
    entry = np.ascontiguousarray(entry, dtype=dt); exit_ = np.ascontiguousarray(exit_, dtype=dt)
    rays = np.ascontiguousarray(rays, dtype=dt); n = np.ascontiguousarray(n, dtype=np.int32)
    W, H = n.shape
    out = np.empty((W, H, 4), dt); steps = np.empty((W, H), np.int32)
    fn = getattr(lib(), "dro_march_fwd" + _suffix(dt))
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, _ct(dt), ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(_p(vol), *vol.shape, _p(tf), tf.shape[0], _p(cam), _p(entry), _p(exit_), _p(rays), _p(n),
            W, H, int(S), float(sr), int(mode), _p(out), _p(steps))
    assert rc == 0
    return out, steps


def march_bwd(vol, tf, cam, entry, exit_, rays, n, S, sr, grad_out, want_vol=True, want_tf=True):
    """Adjoint of march_fwd(mode=0) w.r.t. vol and tf (VR.py:460-461,470-471). Returns (d_vol, d_tf)."""
    dt = vol.dtype
    vol = np.ascontiguousarray(vol); tf = np.ascontiguousarray(tf, dtype=dt)
    cam = np.ascontiguousarray(cam, dtype=dt)
    entry = np.ascontiguousarray(entry, dtype=dt); exit_ = np.ascontiguousarray(exit_, dtype=dt)
    rays = np.ascontiguousarray(rays, dtype=dt); n = np.ascontiguousarray(n, dtype=np.int32)
    grad_out = np.ascontiguousarray(grad_out, dtype=dt)
    W, H = n.shape
    d_vol = np.zeros_like(vol) if want_vol else None
    d_tf = np.zeros_like(tf) if want_tf else None
    fn = getattr(lib(), "dro_march_bwd" + _suffix(dt))
    fn.restype = ctypes.c_int
    fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                   ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                   ctypes.c_int, ctypes.c_int, ctypes.c_int, _ct(dt), ctypes.c_void_p,
                   ctypes.c_void_p, ctypes.c_void_p]
    rc = fn(_p(vol), *vol.shape, _p(tf), tf.shape[0], _p(cam), _p(entry), _p(exit_), _p(rays), _p(n),
            W, H, int(S), float(sr), _p(grad_out), _p(d_vol), _p(d_tf))
    assert rc == 0
    return d_vol, d_tf


def mse_loss_grad(out, ref, inv_norm=None):
    """(loss, grad) of inv_norm * sum (out-ref)^2; inv_norm defaults to 1/numel (torch mse_loss, EX.py:439-443)."""
    out = np.ascontiguousarray(out, np.float32)
    ref = np.ascontiguousarray(ref, np.float32)
    inv_norm = 1.0 / out.size if inv_norm is None else inv_norm
    grad = np.empty_like(out)
    fn = lib().dro_mse_loss_grad_f32
    fn.restype = ctypes.c_double
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_float, ctypes.c_void_p]
    loss = fn(_p(out), _p(ref), out.size, np.float32(inv_norm), _p(grad))
    return loss, grad


def tf_momentum_step(tf, d_tf, momentum, lr, gamma, max_grad):
    """In-place apply_grad (EX.py:375-381) on copies; returns (tf, momentum)."""
    tf = np.array(tf, np.float32, order="C")
    mom = np.array(momentum, np.float32, order="C")
    g = np.ascontiguousarray(d_tf, np.float32)
    fn = lib().dro_tf_momentum_step_f32
    fn.restype = None
    fn.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_float,
                   ctypes.c_float]
    fn(_p(tf), _p(g), _p(mom), tf.size, lr, gamma, max_grad)
    return tf, mom


def render(vol, tf, cam, out_shape, S=1 << 30, sr=1.0, mode=0, jitter_seed=0, view=0, fov_deg=30.0, near=0.1):
    """ray_setup + march_fwd in one call (the sequence of VR.py:431-438)."""
    W, H = out_shape
    S = min(S, 1 << 30)
    e, x, r, n = ray_setup(cam, W, H, vol.shape, sr, fov_deg, near, jitter_seed, view, vol.dtype)
    rgba, steps = march_fwd(vol, tf, cam, e, x, r, n, S, sr, mode)
    return rgba, steps, (e, x, r, n)


# ---- shared synthetic inputs; numpy only so tests and bench agree ----
def synth_volume(N, seed=1234, dtype=np.float32):
    """Smooth, nowhere-flat volume in [0,1]: gaussian blobs + radial falloff + linear ramp."""
    if isinstance(N, int):
        N = (N, N, N)
    rng = np.random.RandomState(seed)
    centres = rng.uniform(-0.6, 0.6, size=(6, 3))
    sigmas = rng.uniform(0.15, 0.5, size=6)
    ax = [np.linspace(-1.0, 1.0, n, dtype=np.float64) for n in N]
    X, Y, Z = np.meshgrid(*ax, indexing="ij", sparse=True)
    acc = np.zeros(N, np.float64)
    for c, s in zip(centres, sigmas):
        acc += np.exp(-((X - c[0]) ** 2 + (Y - c[1]) ** 2 + (Z - c[2]) ** 2) / (2 * s * s))
    v = 0.2 + 0.6 * acc / acc.max() - 0.03 * (X * X + Y * Y + Z * Z) + 0.02 * (X + 2 * Y + 3 * Z) / 6.0
    return np.clip(v, 0.0, 1.0).astype(dtype)  # stays inside (0.09, 0.83): the clip never bites


def bench_tf(R, alpha, dtype=np.float32):
    """Sinusoidal RGB, constant alpha (keeps early termination from firing when alpha = 3/n_max)."""
    i = np.arange(R, dtype=np.float64) / max(R - 1, 1)
    tf = np.empty((R, 4), np.float64)
    for k, phi in enumerate((0.0, 2.1, 4.2)):
        tf[:, k] = 0.5 + 0.5 * np.sin(2 * math.pi * i + phi)
    tf[:, 3] = alpha
    return tf.astype(dtype)


def peaks_tf(R, dtype=np.float32):
    """A 'realistic' TF with two opacity peaks (tf1-like, UT.py:9-21) so early termination fires."""
    i = np.arange(R, dtype=np.float64) / max(R - 1, 1)
    tf = np.empty((R, 4), np.float64)
    tf[:, 0] = 0.9 - 0.5 * i
    tf[:, 1] = 0.3 + 0.6 * i
    tf[:, 2] = 0.5 + 0.4 * np.sin(6.0 * i)
    tf[:, 3] = 0.35 * np.exp(-((i - 0.55) / 0.05) ** 2) + 0.08 * np.exp(-((i - 0.75) / 0.03) ** 2)
    return tf.astype(dtype)


def in_circles(i, y=0.7, dist=2.5):
    """UT.py:80-83."""
    return np.array([math.cos(i) * dist, y, math.sin(i) * dist], np.float32)
