#!/usr/bin/env python3
"""Second, independently derived gradient reference: tests/golden/autograd_*.npz.

A float64 PyTorch TRANSLITERATION of the reference's differentiable march -- `raycast` + `get_final_image`
(differender/volume_raycaster.py:261-306, 363-372, with its helpers :7-21, :153-219) -- written from the reference's
text, statement by statement, vectorised over the pixels. Its backward pass is NOT written anywhere: torch.autograd
differentiates the forward program, which is how the reference itself obtains its gradients (Taichi autodiff,
VR.py:460-461,470-471). Branch predicates are frozen (`.detach()` masks), as in any reverse-mode AD of a program
with branches; max(x, 0) passes the gradient iff 0 < x and min(1, L) iff not 1 < L, Taichi's rules.

The hand-derived, tape-free adjoint in oracle/dr_oracle_impl.inc and in the HIP kernels shares no code and no
derivation with this file: tests/test_autograd_golden.py requires the f64 oracle to agree with these vectors on EVERY
element to 1e-9, and the HIP path within the parity tolerance. (Still "parity unpinned" w.r.t. a running reference --
the reference holds no fixtures and Taichi is not installed -- but the adjoint is no longer its own witness.)

Inputs (ray buffers included) come from the oracle's float64 ray setup; they are stored in the fixture, so the
vectors do not depend on it.   Run:  python tests/golden/make_autograd_golden.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

F64 = torch.float64


def mix(x, y, a):                      # taichi_glsl.mix
    return x * (1.0 - a) + y * a


def low_high_frac(x):                  # VR.py:7-21
    x = torch.where(x > 0, x, torch.zeros_like(x))      # max(x, 0): gradient iff 0 < x
    low = torch.floor(x).detach()
    return low.long(), low.long() + 1, x - low


def sample_volume_trilinear(vol, pos):  # VR.py:153-189
    shape = torch.tensor(vol.shape, dtype=F64)
    p = torch.clamp(0.5 * pos + 0.5, 0.0, 1.0) * (shape - 1.0 - 1e-4)
    xl, xh, xf = low_high_frac(p[:, 0])
    yl, yh, yf = low_high_frac(p[:, 1])
    zl, zh, zf = low_high_frac(p[:, 2])
    xh = torch.clamp(xh, max=vol.shape[0] - 1); yh = torch.clamp(yh, max=vol.shape[1] - 1); zh = torch.clamp(zh, max=vol.shape[2] - 1)
    xl = torch.clamp(xl, max=vol.shape[0] - 1); yl = torch.clamp(yl, max=vol.shape[1] - 1); zl = torch.clamp(zl, max=vol.shape[2] - 1)
    a = mix(vol[xl, yl, zl], vol[xh, yl, zl], xf)
    b = mix(vol[xl, yh, zl], vol[xh, yh, zl], xf)
    z_low = mix(a, b, yf)
    a = mix(vol[xl, yl, zh], vol[xh, yl, zh], xf)
    b = mix(vol[xl, yh, zh], vol[xh, yh, zh], xf)
    z_high = mix(a, b, yf)
    return mix(z_low, z_high, zf)


def get_volume_normal(vol, pos):        # VR.py:191-203
    delta = 1e-3
    ex = torch.tensor([delta, 0.0, 0.0], dtype=F64); ey = torch.tensor([0.0, delta, 0.0], dtype=F64)
    ez = torch.tensor([0.0, 0.0, delta], dtype=F64)
    dx = sample_volume_trilinear(vol, pos + ex) - sample_volume_trilinear(vol, pos - ex)
    dy = sample_volume_trilinear(vol, pos + ey) - sample_volume_trilinear(vol, pos - ey)
    dz = sample_volume_trilinear(vol, pos + ez) - sample_volume_trilinear(vol, pos - ez)
    g = torch.stack([dx, dy, dz], dim=1)
    norm = g.norm(dim=1, keepdim=True)
    flat = (norm == 0).detach()
    # .normalized(); where the gradient vanishes the reference produces NaN. Decision D1 of DESIGN.md: such
    # samples are lit by the ambient term only and send no gradient through the normal.
    return torch.where(flat, torch.zeros_like(g), g / torch.where(flat, torch.ones_like(norm), norm)), flat[:, 0]


def apply_transfer_function(tf, intensity):   # VR.py:205-219
    R = tf.shape[0]
    lo, hi, fr = low_high_frac(intensity * float(R - 1))
    hi = torch.clamp(hi, max=R - 1)
    lo = torch.clamp(lo, max=R - 1)           # defined-domain guard for intensity > 1 (the reference reads out of bounds)
    return mix(tf[lo], tf[hi], fr[:, None])


def raycast(vol, tf, cam, entry, exit_, rays, n, max_samples, sampling_rate):
    """VR.py:261-306 + 363-372 for all pixels at once. Returns (P,4) output_rgba and the live-sample counts."""
    P = entry.shape[0]
    tape = torch.zeros((P, 4), dtype=F64)                 # render_tape[i, j, sample_idx - 1]; H1: tape[-1] := 0
    count = torch.zeros(P, dtype=torch.long)
    ambient, diffuse_k, specular_k, shininess = 0.4, 0.8, 0.3, 32.0
    light_pos = cam + torch.tensor([0.0, 1.0, 0.0], dtype=F64)
    nf = n.to(F64)
    for s in range(int(n.max())):
        active = ((s < n) & (tape[:, 3] < 0.99) & (s < max_samples)).detach()   # VR.py:267-269
        if not bool(active.any()):
            continue
        ray_len = exit_ - entry
        tmin = entry + 0.5 * ray_len / nf
        frac = torch.where(n > 1, float(s) / torch.clamp(nf - 1.0, min=1.0), torch.zeros_like(nf))
        pos = cam[None, :] + mix(tmin, exit_, frac)[:, None] * rays
        pos = torch.where(active[:, None], pos, torch.zeros_like(pos))          # inactive pixels: any in-range position
        intensity = sample_volume_trilinear(vol, pos)
        sample_color = apply_transfer_function(tf, intensity)
        opacity = 1.0 - torch.pow(1.0 - sample_color[:, 3], 1.0 / sampling_rate)
        normal, flat = get_volume_normal(vol, pos)
        ld = pos - light_pos[None, :]
        light_dir = ld / ld.norm(dim=1, keepdim=True)
        ndl_raw = (normal * light_dir).sum(1)
        n_dot_l = torch.where(ndl_raw > 0, ndl_raw, torch.zeros_like(ndl_raw))
        r = light_dir - 2.0 * (normal * light_dir).sum(1, keepdim=True) * normal    # tl.reflect(I, N) = I - 2 dot(N, I) N
        rdv_raw = (r * (-rays)).sum(1)
        r_dot_v = torch.where(rdv_raw > 0, rdv_raw, torch.zeros_like(rdv_raw))
        r_dot_v = torch.where(flat, torch.zeros_like(r_dot_v), r_dot_v)            # NaN-suppressing max on a NaN normal
        specular = specular_k * torch.pow(r_dot_v, shininess)
        Lraw = diffuse_k * n_dot_l + specular + ambient
        L = torch.where(Lraw > 1.0, torch.ones_like(Lraw), Lraw)                   # ti.min(1.0, .)
        shaded = torch.cat([(L * opacity)[:, None] * sample_color[:, :3], opacity[:, None]], dim=1)
        new = (1.0 - tape[:, 3:4]) * shaded + tape
        tape = torch.where(active[:, None], new, tape)                            # else-branch: tape[s] = tape[s-1]
        count = count + active.long()
    return tape, count                                                             # get_final_image: output += tape[n-1]


CASES = {
    # name: (volume shape, image, R, sampling rate, max_samples, TF kind, camera angle, jitter seed)
    "a_sr1": ((16, 16, 16), (16, 16), 8, 1.0, 4096, "thin", 0.8, 0),
    "b_sr2_ert": ((20, 16, 24), (16, 12), 16, 2.0, 4096, "opaque", 2.1, 0),
    "c_clip": ((24, 24, 24), (12, 16), 12, 1.0, 23, "thin", 4.0, 0),
    "d_sr1_ert_jit": ((18, 22, 16), (16, 16), 32, 1.0, 4096, "opaque", 5.2, 4242),
}


def make_inputs(name):
    from oracle import oracle as O
    vshape, WH, R, sr, S, kind, ang, seed = CASES[name]
    rng = np.random.RandomState({"a_sr1": 1, "b_sr2_ert": 2, "c_clip": 3, "d_sr1_ert_jit": 4}[name])
    vol = O.synth_volume(vshape, dtype=np.float64)
    vol = np.clip(vol + 0.02 * rng.standard_normal(vshape), 0.0, 1.0)      # de-smooth: exercises every corner weight
    tf = rng.uniform(0.05, 0.95, size=(R, 4))
    tf[:, 3] = np.linspace(0.01, 0.06, R) if kind == "thin" else np.linspace(0.0, 0.9, R) ** 2 + 0.05
    cam = O.in_circles(ang).astype(np.float64)
    e, x, r, n = O.ray_setup(cam, *WH, vshape, sr=sr, jitter_seed=seed, view=2, dtype=np.float64)
    g = rng.standard_normal((*WH, 4))
    g[n == 1] = 0.0     # single-sample rays divide 0 by 0 in the reference (VR.py:279-280): kept out of the comparison
    return dict(vol=vol, tf=tf, cam=cam, entry=e, exit=x, rays=r, n=n, grad_out=g, sr=np.float64(sr), max_samples=np.int32(S))


def run_case(inp):
    T = lambda a: torch.from_numpy(np.ascontiguousarray(a))
    vol = T(inp["vol"]).requires_grad_(True)
    tf = T(inp["tf"]).requires_grad_(True)
    W, H = inp["n"].shape
    hit = inp["n"].reshape(-1) > 1      # n == 1: see make_inputs
    sel = torch.from_numpy(np.nonzero(hit)[0])
    out_hit, cnt_hit = raycast(vol, tf, T(inp["cam"]), T(inp["entry"]).reshape(-1)[sel], T(inp["exit"]).reshape(-1)[sel],
                               T(inp["rays"]).reshape(-1, 3)[sel], T(inp["n"]).reshape(-1)[sel].long(),
                               int(inp["max_samples"]), float(inp["sr"]))
    out = torch.zeros((W * H, 4), dtype=F64).index_copy(0, sel, out_hit)
    (out * T(inp["grad_out"]).reshape(-1, 4)).sum().backward()
    steps = np.zeros(W * H, np.int32); steps[sel.numpy()] = cnt_hit.numpy()
    return dict(rgba=out.detach().numpy().reshape(W, H, 4), steps=steps.reshape(W, H), dvol=vol.grad.numpy(), dtf=tf.grad.numpy())


def main():
    for name in CASES:
        inp = make_inputs(name)
        res = run_case(inp)
        path = os.path.join(HERE, f"autograd_{name}.npz")
        np.savez_compressed(path, **inp, **res)
        print("wrote", path, os.path.getsize(path), "bytes; live/planned samples", int(res["steps"].sum()), int(inp["n"].sum()),
              "max|dvol| %.3e max|dtf| %.3e" % (np.abs(res["dvol"]).max(), np.abs(res["dtf"]).max()))


if __name__ == "__main__":
    main()
