#!/usr/bin/env python3
"""TF-only backward against the oracle on a smooth and on a binary (mask) volume: per-texel relative errors."""
import sys, os
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O
from differender_amd import functional as F
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
for name in ("smooth", "binary", "smooth_peaks"):
    N, R, WH = 24, 16, (24, 24)
    vol = O.synth_volume(N)
    if name == "binary":
        vol = (vol > 0.45).astype(np.float32)
    tf = O.peaks_tf(R) if name == "smooth_peaks" else O.bench_tf(R, 0.03)
    if name != "smooth_peaks":
        tf[:, 3] = np.linspace(0.01, 0.08, R)
    cam = O.in_circles(0.3)
    e, x, r, n = O.ray_setup(cam, *WH, vol.shape, 1.0)
    ref, _ = O.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, 0)
    g = np.ones_like(ref)
    _, dt0 = O.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, True, True)
    tv, tt, tc = T(vol), T(tf), T(cam[None])
    te, tx, tr, tn = T(e[None]), T(x[None]), T(r[None]), T(n[None])
    ws = F.alloc_workspace(1, WH, vol.shape, R, dev)
    out, _ = F.march_fwd(tv, tt, tc, te, tx, tr, tn, 4096, 1.0, workspace=ws)
    _, dt = F.march_bwd(tv, tt, tc, te, tx, tr, tn, 4096, 1.0, T(g[None]), out, want_vol=False, want_tf=True, workspace=ws)
    _, dt2 = F.march_bwd(tv, tt, tc, te, tx, tr, tn, 4096, 1.0, T(g[None]), out, want_vol=True, want_tf=True, workspace=ws)
    dt, dt2 = dt.cpu().numpy(), dt2.cpu().numpy()
    print(name, "fwd err %.2e" % np.abs(out[0].cpu().numpy() - ref).max(), "tf-only err %.3e" % (np.abs(dt - dt0).max() / np.abs(dt0).max()),
          "vol+tf err %.3e" % (np.abs(dt2 - dt0).max() / np.abs(dt0).max()))
    rel = (dt - dt0) / np.maximum(np.abs(dt0), 1e-12)
    print("   per-texel rel err (alpha channel):", np.array2string(rel[:, 3], precision=4, max_line_width=200))
