"""Dig into one seed of tools/fuzz_parity.py: fast path, baseline kernels, f32 oracle and f64 oracle side by side.
    python tools/fuzz_debug.py seed [seed...]"""
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


def rel(a, b):
    return float(np.abs(a - b).max()) / max(float(np.abs(b).max()), 1e-30)


for seed in map(int, sys.argv[1:]):
    c = fz.make_case(seed)
    print("=== seed", seed, fz.describe(c))
    vol_h, tf_h, cam_h, WH, vshape, sr, S, mode = c["vol"], c["tf"], c["cam"], c["WH"], c["vshape"], c["sr"], c["S"], c["mode"]
    vol = T(vol_h.astype(np.float16)) if c["f16"] else T(vol_h)
    tf, cam = T(tf_h), T(cam_h)
    e, x, r, n = Fn.ray_setup(cam, WH, vshape, sr, jitter_seed=c["jitter"])
    eh, xh, rh, nh = (t.cpu().numpy() for t in (e, x, r, n))
    for v in range(c["n_views"]):
        ro = O.ray_setup(cam_h[v], *WH, vshape, sr=sr, jitter_seed=c["jitter"], view=v)
        for nm, a, b in zip(("entry", "exit", "rays", "n"), ro, (eh[v], xh[v], rh[v], nh[v])):
            d = ~((a == b) | (np.isnan(a) & np.isnan(b))) if a.dtype != np.int32 else a != b
            if d.any():
                i = tuple(np.argwhere(d)[0])
                print(f"  ray setup view {v}: {nm} differs at {int(d.sum())} elements, e.g. {i}: oracle {a[i]!r} device {b[i]!r}; n there {ro[3][i[:2]]} / {nh[v][i[:2]]}")
    res = {}
    for name, variant in (("flat", 0), ("base", 1)):
        ws = Fn.alloc_workspace(c["n_views"], WH, vshape, c["R"], fz.dev) if variant == 0 else None
        out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, mode, variant=variant, workspace=ws)
        dv = dt = None
        if mode == 0:
            dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(c["g"]), out, True, True, variant=variant, workspace=ws)
            dv, dt = dv.float().cpu().numpy(), dt.cpu().numpy()
        res[name] = (out.cpu().numpy(), steps.cpu().numpy(), dv, dt)
        if ws is not None:
            print("  workspace stats", Fn.workspace_stats(ws)[:6].tolist())
    for name, dt_ in (("o32", np.float32), ("o64", np.float64)):
        outs, stps = [], []
        dv = np.zeros(vshape, dt_); dtf = np.zeros(tf_h.shape, dt_)
        for v in range(c["n_views"]):
            args = (vol_h.astype(dt_), tf_h.astype(dt_), cam_h[v].astype(dt_), eh[v].astype(dt_), xh[v].astype(dt_), rh[v].astype(dt_), nh[v])
            o, s = O.march_fwd(*args, S, sr, mode); outs.append(o); stps.append(s)
            if mode == 0:
                a, b = O.march_bwd(*args, S, sr, c["g"][v].astype(dt_)); dv += a; dtf += b
        res[name] = (np.stack(outs), np.stack(stps), dv if mode == 0 else None, dtf if mode == 0 else None)
    o64 = res["o64"]
    for name in ("flat", "base", "o32"):
        rr = res[name]
        line = f"  {name}: fwd vs o64 {np.abs(rr[0] - o64[0]).max():.3e}  vs o32 {np.abs(rr[0] - res['o32'][0]).max():.3e}  steps!=o32 {int((rr[1] != res['o32'][1]).sum())} !=o64 {int((rr[1] != o64[1]).sum())}"
        if mode == 0:
            line += f"  d_vol rel vs o64 {rel(rr[2], o64[2]):.3e} vs o32 {rel(rr[2], res['o32'][2]):.3e}  d_tf vs o64 {rel(rr[3], o64[3]):.3e} vs o32 {rel(rr[3], res['o32'][3]):.3e}"
        print(line)
    if mode == 0:
        d = np.abs(res["flat"][2] - res["o32"][2]); idx = np.unravel_index(np.argmax(d), d.shape)
        print("  worst d_vol voxel", idx, {k: float(res[k][2][idx]) for k in res}, "max|o64|", float(np.abs(o64[2]).max()),
              "nan in o32/o64:", bool(np.isnan(res['o32'][2]).any()), bool(np.isnan(o64[2]).any()))
        order = np.argsort(d.ravel())[::-1][:12]
        print("  top differing voxels (flat - o32):", [(tuple(int(q) for q in np.unravel_index(o, d.shape)), float((res['flat'][2] - res['o32'][2]).ravel()[o])) for o in order])
        print("  g range", float(np.abs(c['g']).min()), float(np.abs(c['g']).max()), "sum of differences", float((res['flat'][2] - res['o32'][2]).sum()), "sum |o32|", float(np.abs(res['o32'][2]).sum()))
    bad = np.argwhere(res["flat"][1] != res["o32"][1])
    for b in bad[:4]:
        b = tuple(b)
        print("  steps differ at", b, "n", int(nh[b]), {k: int(res[k][1][b]) for k in res}, "alpha", {k: float(res[k][0][b][3]) for k in res})
