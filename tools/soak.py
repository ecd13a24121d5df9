"""Soak: repeated full-size forward/backward must be bit-reproducible (forward) and stable (backward sums), no ray may
fall back to individual marching, across different cameras and both transfer functions."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from differender_amd.utils import get_tf
from bench import synth_volume_torch, bench_tf_torch, in_circles
dev = torch.device("cuda:0")
N, IMG, R = 512, 512, 256
vol = synth_volume_torch(N, dev)
ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, R, dev)
bad = 0
for tfname in ("bench", "tf1"):
    tf = bench_tf_torch(R, 1e-3, dev) if tfname == "bench" else get_tf("tf1", R).t().contiguous().to(dev)
    for ci in range(6):
        cam = torch.tensor([in_circles(0.37 * ci + 0.05)], dtype=torch.float32, device=dev)
        e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, 1.0)
        ref = None
        for rep in range(4):
            out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
            rep_fallback = int(F.workspace_stats(ws)[0])
            g = torch.ones_like(out)
            dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g, out, workspace=ws)
            cur = (out.clone(), steps.clone(), float(dv.double().sum()), float(dt.double().sum()), float(dv.abs().max()))
            if ref is None:
                ref = cur
            else:
                same_fwd = torch.equal(cur[0], ref[0]) and torch.equal(cur[1], ref[1])
                rel = abs(cur[2] - ref[2]) / max(abs(ref[2]), 1e-30)
                if not same_fwd or rel > 1e-6 or not torch.isfinite(dv).all():
                    bad += 1
                    print("MISMATCH", tfname, ci, rep, same_fwd, rel)
            if rep_fallback:
                bad += 1; print("FALLBACK rays", tfname, ci, rep_fallback)
        print(tfname, "cam", ci, "ok: steps %.3g  sum(dv) %.6e  sum(dt) %.6e  max|dv| %.3e" % (float(ref[1].sum()), ref[2], ref[3], ref[4]), flush=True)
print("soak done, problems:", bad)
sys.exit(1 if bad else 0)
