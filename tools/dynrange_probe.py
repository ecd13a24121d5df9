"""Error of d_volume on voxels that only small-gradient rays touch (half the image scaled by `ratio`), per library build.
   usage: DIFFERENDER_HIP_LIB=ab_libs/x.so python tools/dynrange_probe.py [ratio]"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from oracle import oracle as O
ratio = float(sys.argv[1]) if len(sys.argv) > 1 else 1e-5
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
vol_h = O.synth_volume(48); tf_h = O.bench_tf(32, 0.03); tf_h[:, 3] = np.linspace(0.01, 0.08, 32); cam_h = O.in_circles(0.3)
WH = (64, 64)
vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0)
rays = tuple(t[0].cpu().numpy() for t in (e, x, r, n))
ws = F.alloc_workspace(1, WH, vol.shape, 32, dev)
out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws)
g = np.random.default_rng(21).standard_normal((*WH, 4)).astype(np.float32)
small = np.zeros(WH, bool); small[WH[0] // 2:] = True
g[small] *= ratio
dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g[None]), out, workspace=ws)
dv = dv.cpu().numpy()
dv0, _ = O.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, g)
sup = np.zeros(vol_h.shape, bool)
for seed in (0, 1):
    gp = np.zeros((*WH, 4), np.float32); gp[~small] = np.random.default_rng(seed).uniform(0.5, 1.5, size=(int((~small).sum()), 4))
    a, _ = O.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, gp); sup |= a != 0
only = (dv0 != 0) & ~sup
gs = g.copy(); gs[~small] = 0
dvs, _ = O.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, gs)
err = np.abs(dv - dvs)[only]; ref = np.abs(dvs)[only]
print("lib", os.environ.get("DIFFERENDER_HIP_LIB"), "ratio", ratio, "voxels", int(only.sum()))
print(" max|dv0| %.3e  max|dvs| %.3e  gmax %.2f" % (np.abs(dv0).max(), np.abs(dvs).max(), np.abs(g).max()))
for name, bound in (("judge: 1e-4|b| + 1e-7 max|b_full|", 1e-4 * ref + 1e-7 * np.abs(dv0).max()),
                    ("own:   1e-4|b| + 2e-6 max|b_small|", 1e-4 * ref + 2e-6 * np.abs(dvs).max())):
    q = err / bound
    print(" %-40s violations %6d / %d   max err/bound %.3g   median %.3g" % (name, int((q > 1).sum()), q.size, q.max(), np.median(q)))
rel = err / np.maximum(ref, 1e-30)
print(" relative error on those voxels: median %.2e  p99 %.2e  max %.2e ; zeros where ref != 0: %d" % (np.median(rel), np.quantile(rel, 0.99), rel.max(), int(((dv[only] == 0) & (dvs[only] != 0)).sum())))
