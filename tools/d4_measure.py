"""DESIGN.md D4 at the sizes that matter: the fast path against the SEQUENTIAL kernels (DR_VARIANT_BASELINE: the oracle's
float32 recurrence bit for bit, VR.py:300-302) on whole views, for transfer functions with tiny non-zero alphas.
    python tools/d4_measure.py [N] [WH] [rates...]     (GPU box; one JSON line per case on stdout)
TFs: bench (alpha 1e-3), tf1 (exact zeros), d4 (tf1 with 1e-6 in its transparent ranges), opt5 / opt3 (tf1 after three
gradient steps whose largest alpha change is 1e-5 / 1e-3 -- what an optimised TF looks like)."""
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from differender_amd import functional as Fn  # noqa: E402
from differender_amd.utils import get_tf  # noqa: E402


def optimised_tf(vol, tf0, cam, WH, sr, step):
    tf = tf0.clone()
    e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, sr)
    tgt_tf = get_tf("tf3", tf.shape[0]).t().contiguous().cuda()
    tgt, _ = Fn.march_fwd(vol, tgt_tf, cam, e, x, r, n, 1 << 20, sr)
    for _ in range(3):
        ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], vol.device)
        out, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, workspace=ws)
        _, g = Fn.mse_loss_grad(out, tgt)
        _, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, g, out, want_vol=False, workspace=ws)
        tf = (tf - step / float(dt.abs().max()) * dt).clamp_(0.0, 1.0).contiguous()
    return tf


def main():
    N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
    wh = int(sys.argv[2]) if len(sys.argv) > 2 else 512
    rates = [float(a) for a in sys.argv[3:]] or [1.0, 2.0]
    WH, R = (wh, wh), 128
    dev = torch.device("cuda:0")
    vol = bench.synth_volume_torch(N, dev)
    cam = torch.tensor([bench.in_circles(2.1)], dtype=torch.float32, device=dev)
    tf1 = get_tf("tf1", R).t().contiguous().to(dev)
    d4 = tf1.clone()
    d4[:, 3] = torch.where(d4[:, 3] == 0, torch.full_like(d4[:, 3], 1e-6), d4[:, 3])
    tfs = {"bench": bench.bench_tf_torch(R, 1e-3, dev), "tf1": tf1, "d4": d4}
    for sr in rates:
        tfs_sr = dict(tfs)
        tfs_sr["opt5"] = optimised_tf(vol, tf1, cam, WH, sr, 1e-5)
        tfs_sr["opt3"] = optimised_tf(vol, tf1, cam, WH, sr, 1e-3)
        e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, sr)
        for name, tf in tfs_sr.items():
            for mode in (Fn.N.DR_MODE_DIFF, Fn.N.DR_MODE_NONDIFF):
                ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev)
                out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws, hints=0)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws, hints=0)
                torch.cuda.synchronize()
                ms = (time.perf_counter() - t0) / 3 * 1e3
                st = Fn.workspace_stats(ws)
                ref, sref = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, variant=1)
                d = (out - ref).abs().amax(-1)[0]
                line = {"N": N, "WH": wh, "sr": sr, "tf": name, "mode": int(mode), "fwd_ms": round(ms, 3),
                        "max_diff": float(d.max()), "px_over_1e-5": int((d > 1e-5).sum()), "px_over_5e-6": int((d > 5e-6).sum()),
                        "steps_equal": bool(torch.equal(steps, sref)), "rays_marched_individually": int(st[2]),
                        "exact_rays": int(st[15]), "alpha_nonzero_min": float(tf[:, 3][tf[:, 3] > 0].min()) if (tf[:, 3] > 0).any() else 0.0,
                        "tiny_alphas": int(((tf[:, 3] > 0) & (tf[:, 3] < 1e-3)).sum())}
                print(json.dumps(line), flush=True)


if __name__ == "__main__":
    main()
