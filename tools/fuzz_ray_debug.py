"""Which ray makes the fast backward differ from the baseline kernels? One backward per pixel (upstream gradient masked to
that pixel), fast path against baseline, for one fuzz seed.   python tools/fuzz_ray_debug.py seed [view]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_argv = sys.argv; sys.argv = sys.argv[:1]
import fuzz_parity as fz
from differender_amd import functional as Fn
from oracle import oracle as O
sys.argv = _argv
T = fz.T
seed = int(sys.argv[1]); view = int(sys.argv[2]) if len(sys.argv) > 2 else 0
c = fz.make_case(seed)
print(fz.describe(c))
vol_h, tf_h, cam_h, WH, vshape, sr, S = c["vol"], c["tf"], c["cam"][view:view + 1], c["WH"], c["vshape"], c["sr"], c["S"]
vol = T(vol_h.astype(np.float16)) if c["f16"] else T(vol_h)
tf, cam = T(tf_h), T(cam_h)
e, x, r, n = Fn.ray_setup(cam, WH, vshape, sr, jitter_seed=c["jitter"], view_base=view)
ws = Fn.alloc_workspace(1, WH, vshape, c["R"], fz.dev)
out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, 0, workspace=ws)
g_all = c["g"][view:view + 1]
worst = []
for i in range(WH[0]):
    for j in range(WH[1]):
        if int(n[0, i, j]) <= 0:
            continue
        g = np.zeros_like(g_all); g[0, i, j] = g_all[0, i, j]
        gt = T(g)
        dv, _ = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, gt, out, True, False, workspace=ws)
        db, _ = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, gt, out, True, False, variant=1)
        d = (dv.float() - db.float()).abs()
        m = float(d.max()); s = float(db.float().abs().max())
        if m > 1e-5 * max(s, 1e-30):
            idx = np.unravel_index(int(d.argmax()), d.shape)
            worst.append((m / max(s, 1e-30), i, j, int(n[0, i, j]), int(steps[0, i, j]), idx, float(dv.float()[idx]), float(db.float()[idx])))
worst.sort(reverse=True)
print(len(worst), "pixels differ by more than 1e-5 of their own maximum")
for w in worst[:10]:
    print("  rel %.3e pixel (%d,%d) n %d steps %d voxel %s fast %.6e baseline %.6e" % w)
if worst:
    _, i, j, nn, st, idx, _, _ = worst[0]
    eh, xh, rh = float(e[0, i, j]), float(x[0, i, j]), r[0, i, j].cpu().numpy()
    print("  ray: entry", eh, "exit", xh, "dir", rh, "cam", cam_h[0])
    # sample positions near the offending voxel (oracle's formula, VR.py:276-281)
    t0 = np.float32(eh) + np.float32(0.5) * (np.float32(xh) - np.float32(eh)) / np.float32(nn)
    for s_ in range(min(nn, st + 1)):
        q = np.float32(s_) / np.float32(max(nn - 1, 1))
        t = t0 * (1 - q) + np.float32(xh) * q
        pos = cam_h[0] + t * rh
        vc = (np.clip(0.5 * pos + 0.5, 0, 1) * (np.array(vshape, np.float32) - 1 - 1e-4))
        if np.all(np.abs(vc - np.array(idx)) < 2.5):
            print("    sample", s_, "voxel coords", vc, "frac", vc - np.floor(vc))

if worst:
    # which sample? the differentiable march stops after max_samples samples (H2): find the smallest S at which the two differ
    _, i, j, nn, st, idx, _, _ = worst[0]
    g = np.zeros_like(g_all); g[0, i, j] = g_all[0, i, j]; gt = T(g)

    def differs(S_):
        o_, s_ = Fn.march_fwd(vol, tf, cam, e, x, r, n, S_, sr, 0, workspace=ws)
        dv, _ = Fn.march_bwd(vol, tf, cam, e, x, r, n, S_, sr, gt, o_, True, False, workspace=ws)
        db, _ = Fn.march_bwd(vol, tf, cam, e, x, r, n, S_, sr, gt, o_, True, False, variant=1)
        d = (dv.float() - db.float()).abs()
        return float(d.max()) > 1e-5 * max(float(db.float().abs().max()), 1e-30), dv, db
    lo_, hi_ = 0, min(nn, st + 1)
    assert differs(hi_)[0]
    while hi_ - lo_ > 1:
        mid = (lo_ + hi_) // 2
        if differs(mid)[0]: hi_ = mid
        else: lo_ = mid
    s_bad = hi_ - 1
    print("  first differing sample index:", s_bad, "(S =", hi_, ")")
    for s_ in range(max(0, s_bad - 1), min(nn, s_bad + 2)):
        q = np.float32(s_) / np.float32(max(nn - 1, 1))
        t = t0 * (1 - q) + np.float32(xh) * q
        pos = cam_h[0] + t * rh
        for nm, dp in (("centre", 0.0), ("+d", 1e-3), ("-d", -1e-3)):
            vc = (np.clip(0.5 * (pos + np.float32(dp)) + 0.5, 0, 1) * (np.array(vshape, np.float32) - 1 - 1e-4))
            print("    sample", s_, nm, "voxel coords", vc, "cells", np.floor(vc).astype(int), "cells mod 12", np.floor(vc).astype(int) % 12)
    _, dv, db = differs(hi_)
    d = (dv.float() - db.float()).cpu().numpy()
    order = np.argsort(np.abs(d).ravel())[::-1][:10]
    print("  at S =", hi_, "differences:", [(tuple(int(q) for q in np.unravel_index(o, d.shape)), float(d.ravel()[o])) for o in order])
