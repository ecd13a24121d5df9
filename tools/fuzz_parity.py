"""Randomised parity fuzzer (GPU box): HIP fast path against the CPU oracle on the same ray buffers, far wider than
tests/test_gpu_parity.py::test_random_configurations -- ragged and anisotropic volumes (2..96 per axis), images up to 96 px,
1..3 views sharing a volume, TF sizes 1..300 with empty / opaque / spiky alpha, sampling rates 0.3..16, clipped
max_samples, fp16 and strided volumes, jitter, cameras anywhere (inside the volume, on a face, far away, axis aligned).

    python tools/fuzz_parity.py [seconds=300] [first_seed=0]          (FUZZ_SCALE=3: volumes and images three times the size)

Prints one line per failing configuration (with the seed that reproduces it) and a summary; exit code 1 on any failure.
Tolerances are those of the parity tests: ray setup and steps bit-exact, RGBA 1e-5, gradients 1e-4 of the tensor's largest
magnitude -- or, where the order of the float atomics alone moves the f32 result by that much (gauged by the baseline
kernels, which share the oracle's arithmetic), 3 x the baseline's distance from the oracle + 1e-4.
"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as Fn  # noqa: E402
from oracle import oracle as O  # noqa: E402

dev = torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev)


SCALE = int(os.environ.get("FUZZ_SCALE", "1"))   # 3: volumes up to 288 per axis, images up to 288 px (a case takes seconds)
PROFILE = os.environ.get("FUZZ_PROFILE", "")     # "inside": every camera inside or next to the volume (work items of heavy
                                                 # bricks); "ert": opaque TFs at high sampling rates (termination decisions)


def make_case(seed):
    rng = np.random.default_rng(770000 + seed)
    big = rng.random() < 0.25
    hi = (97 if big else 40) * SCALE - (SCALE - 1)
    vshape = tuple(int(v) for v in rng.integers(2, hi, 3))
    if rng.random() < 0.15:   # strongly anisotropic
        vshape = tuple(int(v) for v in (rng.integers(2, 9), rng.integers(min(30, hi - 1), hi), rng.integers(2, hi)))
        vshape = tuple(np.array(vshape)[rng.permutation(3)].tolist())
    wh_hi = (97 if big else 49) * SCALE - (SCALE - 1)
    WH = (int(rng.integers(1, wh_hi)), int(rng.integers(1, wh_hi)))
    R = int(rng.choice([1, 2, 3, 7, 16, 64, 128, 256, 300]))
    sr = float(rng.choice([0.3, 0.6, 1.0, 1.0, 1.0, 1.5, 2.0, 3.0, 4.0, 8.0, 16.0]))
    mode = int(rng.random() < 0.35)
    n_views = int(rng.choice([1, 1, 2, 3]))
    S = int(rng.choice([5000, 5000, 5000, 1, 2, 7, 33, 150]))
    vol = rng.random(vshape, dtype=np.float32)
    kind = rng.integers(0, 4)
    if kind >= 1:   # smooth (normals are not pure noise)
        for ax in range(3):
            vol = (vol + np.roll(vol, 1, ax) + np.roll(vol, -1, ax)) / 3.0
    if kind == 2:   # flat regions: zero gradient -> ambient-only samples (D1)
        vol = np.round(vol * 3.0) / 3.0
    if kind == 3:   # sparse: mostly empty space
        vol = np.clip((vol - 0.5) * 8.0, 0.0, 1.0)
    vol = np.ascontiguousarray(vol.astype(np.float32))
    f16 = rng.random() < 0.25
    if f16:
        vol = vol.astype(np.float16).astype(np.float32)
    tf = rng.random((R, 4), dtype=np.float32)
    ak = rng.integers(0, 6)
    if PROFILE == "ert":
        ak = [2, 3, 4, 3][seed % 4]; sr = [2.0, 4.0, 8.0, 16.0, 3.0, 1.0][seed % 6]
    tf[:, 3] *= [0.004, 0.02, 0.2, 0.9, 1.0, 0.05][ak]
    if ak == 4:     # alpha reaches exactly 1 somewhere: opacity 1, transmittance 0
        tf[rng.integers(0, R), 3] = 1.0
    if ak == 5 and R > 4:   # empty ranges (alpha exactly 0) + a spike
        tf[: R // 2, 3] = 0.0
        tf[rng.integers(R // 2, R), 3] = 0.95
    # PROFILE "d4" (round 5): transparent ranges at alpha 1e-6 instead of 0 around an opaque structure, rates >= 2 -- sub-ulp
    # contributions behind opaque samples, which sequential float32 compositing drops and the brick kernels' partials keep (DESIGN.md
    # D4; since round 6 such rays are recomputed sample by sample). Drawn from a generator of its own and only under that profile, so
    # that the other fields of a seed stay what they were.
    d4 = PROFILE == "d4" and R > 4
    if d4:
        r4 = np.random.default_rng(770000 + seed)
        tf[:, 3] = np.where(tf[:, 3] < 0.5 * tf[:, 3].max(), np.float32(1e-6), tf[:, 3])
        tf[r4.integers(0, R), 3] = np.float32(r4.choice([0.3, 0.6, 0.95]))
        sr = float(r4.choice([2.0, 3.0, 4.0, 8.0]))
    cams = []
    for _ in range(n_views):
        d = rng.standard_normal(3); d /= np.linalg.norm(d)
        ck = rng.integers(0, 8)
        if ck == 0:
            d = np.eye(3)[rng.integers(0, 3)] * rng.choice([-1.0, 1.0])   # axis aligned
        if abs(d[1]) > 0.97:    # (anti)parallel to the up vector is degenerate in VR.py:143
            d = np.array([0.6, 0.3, 0.74]); d /= np.linalg.norm(d)
        dist = float(rng.choice([0.05, 0.5, 0.9, 1.0, 1.2, 1.75, 2.5, 6.0, 40.0]))
        if PROFILE == "inside":
            dist = [0.05, 0.3, 0.5, 0.9, 1.0, 1.1][seed % 6]
        cams.append((d * dist).astype(np.float32))
    cam = np.stack(cams)
    jitter = int(rng.integers(0, 2)) * int(rng.integers(1, 1 << 30))
    strided = (not f16) and rng.random() < 0.2
    g = rng.standard_normal((n_views, *WH, 4)).astype(np.float32)
    if rng.random() < 0.2:
        g *= np.exp(rng.uniform(-9, 9, size=(n_views, *WH, 1))).astype(np.float32)   # wide dynamic range of the upstream gradient
    want = [(True, True), (True, True), (True, False), (False, True)][int(rng.integers(0, 4))]
    variant = 1 if seed % 6 == 5 else 0     # every sixth case runs the baseline kernels (DR_VARIANT_BASELINE) instead
    # every fourth multi-view case: one volume and one TF PER VIEW (VR.py:415-427 "batched" inputs), per-item gradients.
    # (drawn from a generator of their own, so that the other fields of a seed stay what they were)
    vols, tfs = None, None
    if n_views > 1 and seed % 4 == 1:
        r2 = np.random.default_rng(990000 + seed)
        vols = np.stack([np.clip(vol + 0.2 * (r2.random(vshape, dtype=np.float32) - 0.5) * (v > 0), 0.0, 1.0) for v in range(n_views)])
        if f16:
            vols = vols.astype(np.float16).astype(np.float32)
        tfs = np.stack([tf] + [np.clip(tf * r2.uniform(0.5, 1.2, size=tf.shape).astype(np.float32), 0.0, 1.0) for _ in range(n_views - 1)])
    return dict(vshape=vshape, WH=WH, R=R, sr=sr, mode=mode, n_views=n_views, S=S, vol=vol, f16=f16, tf=tf, cam=cam,
                jitter=jitter, strided=strided, g=g, want=want, vol_kind=int(kind), alpha_kind=int(ak), variant=variant, seed=seed, vols=vols, tfs=tfs, d4=d4)


def describe(c):
    return {k: c[k] for k in ("vshape", "WH", "R", "sr", "mode", "n_views", "S", "f16", "jitter", "strided", "want", "vol_kind", "alpha_kind", "variant", "seed")} | {"batched": c["vols"] is not None} | {"cam": c["cam"].tolist()}


def run_case(c):
    """Returns a list of failure strings (empty = pass)."""
    fails = []
    vol_h, tf_h, cam_h = c["vol"], c["tf"], c["cam"]
    WH, vshape, sr, S, mode = c["WH"], c["vshape"], c["sr"], c["S"], c["mode"]
    batched = c["vols"] is not None
    vol_v = [c["vols"][v] if batched else vol_h for v in range(c["n_views"])]   # what view v renders
    tf_v = [c["tfs"][v] if batched else tf_h for v in range(c["n_views"])]
    src = c["vols"] if batched else vol_h
    if c["f16"]:
        vol = T(src.astype(np.float16))
    elif c["strided"]:
        big = torch.zeros(src.shape[:-3] + tuple(v + 3 for v in vshape), device=dev)
        vol = big[..., 1:1 + vshape[0], 2:2 + vshape[1], 0:vshape[2]]
        vol.copy_(T(src))
    else:
        vol = T(src)
    tf_h_dev = c["tfs"] if batched else tf_h
    tf, cam = T(tf_h_dev), T(cam_h)
    variant = c["variant"]
    # the image as one piece, or (every seventh case) as two bands of rows the way a strong-scaling run splits one view
    # across GPUs (dr_*_rows): the bands' ray buffers, images and step counts are concatenated, their gradients summed
    W = WH[0]
    cut = (1 + c["seed"] % (W - 1)) if (c["seed"] % 7 == 3 and W >= 2) else 0
    bands = [(0, W)] if cut == 0 else [(0, cut), (cut, W - cut)]
    c["bands"] = len(bands)
    pieces = []
    for row0, nr in bands:
        rows = None if cut == 0 else (row0, W)
        eb, xb, rb, nb = Fn.ray_setup(cam, (nr, WH[1]), vshape, sr, jitter_seed=c["jitter"], rows=rows)
        # the TF-only backward of the fast path runs over the per-sample tape (DR_TAPE_TF) on every second such case
        use_tape = variant == 0 and mode == 0 and tuple(c["want"]) == (False, True) and c["seed"] % 2 == 0
        c["tape"] = use_tape
        wsb = Fn.alloc_workspace(c["n_views"], (nr, WH[1]), vshape, c["R"], dev, tape=(S, sr) if use_tape else None) if variant == 0 else None
        use_tape = use_tape and wsb is not None
        # caller hints (include/differender_hip.h), right or WRONG, must never change a result: every third fast-path case
        # claims "no ray terminates early" (false for most of the opaque TFs: the device repairs the view), every third
        # claims "many rays terminate" (the grouped pre-pass also below sampling rate 3)
        hint = 0 if variant != 0 else (0, 0x100, 0x200)[c["seed"] % 3]
        ob, sb = Fn.march_fwd(vol, tf, cam, eb, xb, rb, nb, S, sr, mode, variant=variant, workspace=wsb, rows=rows, hints=hint, tape=use_tape)
        pieces.append((rows, row0, nr, eb, xb, rb, nb, wsb, ob, sb, use_tape))
    e, x, r, n, out, steps = (torch.cat([p[k] for p in pieces], dim=1) for k in (3, 4, 5, 6, 8, 9))
    eh, xh, rh, nh = (t.cpu().numpy() for t in (e, x, r, n))
    out_h, steps_h = out.cpu().numpy(), steps.cpu().numpy()
    if not np.isfinite(out_h).all():
        fails.append("non-finite forward output")
    g = c["g"].copy()
    for v in range(c["n_views"]):
        eo, xo, ro, no = O.ray_setup(cam_h[v], *WH, vshape, sr=sr, jitter_seed=c["jitter"], view=v)
        same = all(np.array_equal(a, b, equal_nan=True) for a, b in ((eo, eh[v]), (xo, xh[v]), (ro, rh[v]))) and np.array_equal(no, nh[v])
        if not same:
            fails.append(f"ray setup differs (view {v})")
        ref, sref = O.march_fwd(vol_v[v], tf_v[v], cam_h[v], eh[v], xh[v], rh[v], nh[v], S, sr, mode)
        diff = steps_h[v] != sref     # (1 - a)^(1/sr) is one specified function in oracle and kernels: no excuse at any rate
        if diff.any():
            fails.append(f"steps differ (view {v}): {int(diff.sum())} pixels")
            g[v][diff] = 0.0
        ok = ~diff
        err = float(np.abs(out_h[v] - ref)[ok].max()) if ok.any() else 0.0
        if not err <= 1e-5:
            # (no excuse under any profile: until round 6 the "d4" profile let the fast path sit up to 5e-5 from the float32 oracle where
            #  float64 sided with it -- the bar is the reference's sequential float32 result, and the per-ray passes now hold it:
            #  d4_risk / ray_exact_kernel)
            fails.append(f"forward error {err:.3e} (view {v})")
    if mode == 0:
        dv_ref = np.zeros(src.shape, np.float32); dt_ref = np.zeros_like(tf_h_dev)
        for v in range(c["n_views"]):
            a, b = O.march_bwd(vol_v[v], tf_v[v], cam_h[v], eh[v], xh[v], rh[v], nh[v], S, sr, g[v])
            if batched:
                dv_ref[v] = a; dt_ref[v] = b      # per-item gradients
            else:
                dv_ref += a; dt_ref += b
        wv, wt = c["want"]
        base = None
        dv = dt = None
        gt = T(g)
        for rows, row0, nr, eb, xb, rb, nb, wsb, ob, sb, tp in pieces:
            dvb, dtb = Fn.march_bwd(vol, tf, cam, eb, xb, rb, nb, S, sr, gt[:, row0:row0 + nr].contiguous(), ob, wv, wt,
                                    variant=variant, workspace=wsb, rows=rows, tape=tp)
            dv = dvb if dv is None or dvb is None else dv + dvb
            dt = dtb if dt is None or dtb is None else dt + dtb
        for name, got, ref in (("d_vol", dv, dv_ref), ("d_tf", dt, dt_ref)):
            if got is None:
                continue
            got = got.float().cpu().numpy()
            if not np.isfinite(ref).all():
                # alpha == 1 at a sampling rate != 1: d/da (1 - a)^(1/sr) is infinite there, in the reference as well
                # (its NaN / inf are what RaycastFunction.backward's nan_to_num is for): nothing to compare against.
                # The fast path still has to return finite values; the baseline kernels return what the oracle does.
                c["_undefined"] = True
                if variant == 0 and not np.isfinite(got).all():
                    fails.append(f"{name} non-finite")
                continue
            if not np.isfinite(got).all():
                fails.append(f"{name} non-finite"); continue
            # (floor: a gradient whose largest entry is a millionth of the upstream gradient is the rounding residue of terms
            # that cancel -- n_bar - n (n . n_bar) where the lighting is clamped or turned away -- not a quantity to match)
            scale = max(float(np.abs(ref).max()), 1e-6 * float(np.abs(g).max()), 1e-12)
            err = float(np.abs(got - ref).max()) / scale
            if variant == 1:
                # the baseline kernels add every contribution with a float atomic (as the reference does): thousands land on
                # a few texels of d_tf, in arbitrary order
                if not err <= 1e-3:
                    fails.append(f"{name} error {err:.3e} of max {scale:.3e} (baseline kernels)")
                continue
            if not err <= 1e-4:
                # Ill-conditioned case? Normalising a nearly vanishing gradient amplifies by 1/|grad|, and the f32 result then
                # depends on the order of the float atomics at this level. The baseline kernels -- the oracle's arithmetic,
                # float atomics in another order -- gauge that noise: the fast path may be 3 x as far from the oracle + 1e-4.
                if base is None:
                    bv, bt = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(g), out, True, True, variant=1)
                    base = {"d_vol": bv.float().cpu().numpy(), "d_tf": bt.cpu().numpy()}
                e_base = float(np.abs(base[name] - ref).max()) / scale
                ok = err <= 3.0 * e_base + 1e-4
                e64 = None
                if not ok:
                    # One run of the baseline kernels is a noisy gauge (their own distance from the oracle moves by 8 x from run
                    # to run on such cases). The conditioning itself can be measured: the same backward in float64. Where the
                    # f32 oracle is more than 1 % of the maximum away from it, two f32 evaluations in different summation orders
                    # cannot be asked to agree to 1e-4: the fast path may be 1 % of THAT distance away (still 100 x closer to
                    # the f32 oracle than the f32 oracle is to the exact result).
                    if "_ref64" not in c:
                        f8 = np.float64
                        dv64 = np.zeros(src.shape, f8); dt64 = np.zeros(tf_h_dev.shape, f8)
                        for v in range(c["n_views"]):
                            a, b = O.march_bwd(vol_v[v].astype(f8), tf_v[v].astype(f8), cam_h[v].astype(f8), eh[v].astype(f8),
                                               xh[v].astype(f8), rh[v].astype(f8), nh[v], S, sr, g[v].astype(f8))
                            if batched:
                                dv64[v] = a; dt64[v] = b
                            else:
                                dv64 += a; dt64 += b
                        c["_ref64"] = {"d_vol": dv64, "d_tf": dt64}
                    r64 = c["_ref64"][name]
                    if np.isfinite(r64).all():
                        e64 = float(np.abs(ref - r64).max()) / scale
                        ok = e64 > 1e-2 and err <= 1e-2 * e64
                if not ok:
                    fails.append(f"{name} error {err:.3e} of max {scale:.3e} (baseline kernels: {e_base:.3e}"
                                 + (f", f32 oracle vs f64: {e64:.3e})" if e64 is not None else ")"))
                else:
                    c["_illcond"] = True
    return fails


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 300.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); n_run = 0; n_bad = 0; last = t0; n_undef = 0; n_ill = 0
    while time.time() - t0 < budget:
        c = make_case(seed)
        try:
            fails = run_case(c)
        except Exception as exc:   # an error return of the library is a finding too
            fails = [f"exception {type(exc).__name__}: {exc}"]
        if fails:
            n_bad += 1
            print("FAIL seed", seed, fails, describe(c), flush=True)
        n_run += 1; seed += 1; n_undef += bool(c.get("_undefined")); n_ill += bool(c.get("_illcond"))
        if time.time() - last > 30:
            last = time.time(); print(f"... {n_run} cases, {n_bad} failing, next seed {seed}", flush=True)
    print(f"fuzz done: {n_run} cases in {time.time() - t0:.0f} s, {n_bad} failing, seeds up to {seed - 1}; "
          f"{n_undef} cases with an infinite reference gradient, {n_ill} ill-conditioned cases judged against the baseline kernels' noise")
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
