"""Elementwise d_volume error of the fast path vs the oracle on the opaque 'peaks' scene (no bad pixels)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from oracle import oracle as O
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vol_h = O.synth_volume(32); tf_h = O.peaks_tf(32); cam_h = O.in_circles(0.3)
WH = (24, 24)
vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0)
rays = tuple(t[0].cpu().numpy() for t in (e, x, r, n))
g = np.random.default_rng(4).standard_normal((1, *WH, 4)).astype(np.float32)
for (i, j) in ((3, 4), (10, 11), (17, 5)): g[0, i, j] = 0.0
dv64, _ = O.march_bwd(vol_h.astype(np.float64), tf_h.astype(np.float64), cam_h.astype(np.float64), *[a.astype(np.float64) if a.dtype != np.int32 else a for a in rays], 4096, 1.0, g[0].astype(np.float64))
dv_o, _ = O.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, g[0])
for variant in (0, 1):
    ws = F.alloc_workspace(1, WH, vol.shape, 32, dev) if variant == 0 else None
    out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, variant=variant, workspace=ws)
    dv, _ = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g), out, variant=variant, workspace=ws)
    dv = dv.cpu().numpy()
    for name, ref in (("oracle f32", dv_o), ("oracle f64", dv64)):
        err = np.abs(dv - ref); b = np.abs(ref)
        q = err / (1e-4 * b + 1e-6 * b.max())
        print("variant", variant, "vs", name, ": max|ref| %.3e  max err %.3e  violations %d / %d  max err/bound %.2f  f64bricks %s" % (
            b.max(), err.max(), int((q > 1).sum()), q.size, q.max(), int(F.workspace_stats(ws)[4]) if ws is not None else None))
err = np.abs(dv_o - dv64); b = np.abs(dv64); q = err / (1e-4 * b + 1e-6 * b.max())
print("oracle f32 vs f64: max err %.3e violations %d max err/bound %.2f" % (err.max(), int((q > 1).sum()), q.max()))
