import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from oracle import oracle as O
from differender_amd import functional as F
dev = torch.device("cuda:0")
T = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
sr = float(sys.argv[1]) if len(sys.argv) > 1 else 2.0
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
N = int(sys.argv[3]) if len(sys.argv) > 3 else 48
vol = O.synth_volume(N); tf = O.bench_tf(64, 0.02); cam = O.in_circles(0.3)
WH = (64, 64)
e0, x0, r0, n0 = O.ray_setup(cam, *WH, vol.shape, sr=sr)
ref, sref = O.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, mode)
e, x, r, n = F.ray_setup(T(cam[None]), WH, vol.shape, sr)
for variant in (0, 2, 1):
    ws = F.alloc_workspace(1, WH, vol.shape, 64, dev) if variant != 1 else None
    out, steps = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, mode, variant=variant, workspace=ws)
    o = out[0].cpu().numpy(); d = np.abs(o - ref).max(-1)
    bad = np.argwhere(d > 1e-5)
    print("variant", variant, "max err", d.max(), "n bad", len(bad), "steps eq", np.array_equal(steps[0].cpu().numpy(), sref),
          "stats", F.workspace_stats(ws)[:2].tolist() if ws is not None else None)
    for b in bad[:6]:
        print("   pix", tuple(b), "n", n0[tuple(b)], "err", d[tuple(b)], "out", o[tuple(b)], "ref", ref[tuple(b)])

# per-layer comparison of the F1 partials for the worst pixel (nondiff mode keeps F1 output in the workspace)
def segs(variant):
    ws = F.alloc_workspace(1, WH, vol.shape, 64, dev)
    out, steps = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, 1, variant=variant, workspace=ws)
    NP = WH[0] * WH[1]
    NB = [(s - 1 + 11) // 12 for s in vol.shape]; NL = sum(NB) - 2
    rgba = ws[256:256 + NL * NP * 16].view(torch.float32).view(NL, NP, 4).cpu().numpy()
    cnt = ws[256 + NL * NP * 16:256 + NL * NP * 20].view(torch.int32).view(NL, NP).cpu().numpy()
    return rgba, cnt
ra, ca = segs(0); rb, cb = segs(2)
pix = 23 * 64 + 27
print("cnt flat ", ca[:, pix]); print("cnt brick", cb[:, pix])
for l in range(ca.shape[0]):
    if cb[l, pix] or ca[l, pix]:
        print(l, ca[l, pix], cb[l, pix], ra[l, pix], rb[l, pix], np.abs(ra[l, pix] - rb[l, pix]).max())
dall = np.abs(np.where(cb[..., None] > 0, ra - rb, 0)).max(-1)
print("segments differing > 1e-6:", (dall > 1e-6).sum(), "of", (cb > 0).sum(), "cnt mismatch:", (ca != cb).sum())
