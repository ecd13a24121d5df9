"""How good is the D4 bound (d4_risk, csrc/dr_brick_common.h)? Per pixel: the bound F2 computes from the partials against what
sequential float32 compositing really does to the ray (|fast composite - sequential kernels|).
    python tools/d4_bound_probe.py dump OUT.npz N WH SR TF [MODE]   under DIFFERENDER_HIP_LIB = a -DDR_D4_DEBUG build (bounds),
                                                                    a -DDR_D4_BUDGET_OVERRIDE=3e38f build (pure fast composite)
                                                                    and the shipped library
    python tools/d4_bound_probe.py compare BOUNDS.npz FAST.npz SHIPPED.npz
(tools/gpu_job.sh runs the three dumps; the what-if builds need DIFFERENDER_ALLOW_EXPERIMENT=1.)"""
import sys

import numpy as np


def dump(out, N, wh, sr, tfname, mode):
    import torch
    sys.path.insert(0, ".")
    import bench
    from differender_amd import functional as Fn
    from differender_amd.utils import get_tf
    dev = torch.device("cuda:0")
    R = 128
    vol = bench.synth_volume_torch(N, dev)
    cam = torch.tensor([bench.in_circles(2.1)], dtype=torch.float32, device=dev)
    tf = get_tf("tf1", R).t().contiguous().to(dev)
    if tfname == "d4":
        tf[:, 3] = torch.where(tf[:, 3] == 0, torch.full_like(tf[:, 3], 1e-6), tf[:, 3])
    elif tfname == "bench":
        tf = bench.bench_tf_torch(R, 1e-3, dev)
    e, x, r, n = Fn.ray_setup(cam, (wh, wh), vol.shape, sr)
    ws = Fn.alloc_workspace(1, (wh, wh), vol.shape, R, dev)
    o, st = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws, hints=0)
    ref, sref = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, variant=1)
    np.savez(out, out=o[0].cpu().numpy(), steps=st[0].cpu().numpy(), ref=ref[0].cpu().numpy(), sref=sref[0].cpu().numpy(),
             exact=int(Fn.workspace_stats(ws)[15]))


def compare(fb, ff, fs):
    b, f, s = np.load(fb), np.load(ff), np.load(fs)
    bound = b["out"].max(-1)
    true = np.abs(f["out"] - f["ref"]).max(-1)
    ship = np.abs(s["out"] - s["ref"]).max(-1)
    ok = f["steps"] == f["sref"]
    print(f"steps equal on {ok.mean():.6f} of the rays; exact rays (shipped) {int(s['exact'])}")
    print(f"true |fast - sequential|: max {true[ok].max():.3e}, > 1e-5 on {(true[ok] > 1e-5).sum()} px, > 3e-6 on {(true[ok] > 3e-6).sum()}")
    print(f"shipped: max {ship[ok].max():.3e}, > 1e-5 on {(ship[ok] > 1e-5).sum()} px")
    for thr in (1e-6, 3e-6, 1e-5, 3e-5):
        fl = bound > thr
        print(f"bound > {thr:.0e}: {fl.sum()} rays; true error of the rest: max {true[ok & ~fl].max() if (ok & ~fl).any() else 0:.3e}; "
              f"flagged rays whose true error is below 1e-6: {(fl & (true < 1e-6)).sum()}")
    worst = np.argsort(-(true * ok).ravel())[:8]
    for w in worst:
        i, j = np.unravel_index(w, true.shape)
        print(f"  px ({i},{j}) true {true[i, j]:.3e} bounds {b['out'][i, j]} fast {f['out'][i, j]} seq {f['ref'][i, j]} steps {f['steps'][i, j]}")
    # under-estimates: where the true error exceeds bound + 2e-6
    under = ok & (true > bound + 2e-6)
    print(f"under-estimated (true > bound + 2e-6): {under.sum()} rays, worst excess {(true - bound)[under].max() if under.any() else 0:.3e}")
    ex = np.where(under, true - bound, 0.0)
    for w in np.argsort(-ex.ravel())[:12]:
        i, j = np.unravel_index(w, true.shape)
        print(f"  UNDER px ({i},{j}) true {true[i, j]:.3e} bounds {b['out'][i, j]} fast {f['out'][i, j]} seq {f['ref'][i, j]} steps {f['steps'][i, j]}")


if __name__ == "__main__":
    if sys.argv[1] == "dump":
        dump(sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), float(sys.argv[5]), sys.argv[6], int(sys.argv[7]) if len(sys.argv) > 7 else 0)
    else:
        compare(*sys.argv[2:5])
