#!/usr/bin/env python3
"""Counterpart of the main loop of the reference's examples/taichi_volume_raycaster.py (EX.py:583-601) on
synthetic data: fit a transfer function to a reference render with momentum gradient descent. Each iteration
is render -> loss + d(loss)/d(image) -> backward -> momentum step, all through the C ABI
(dr_march_fwd, dr_mse_loss_grad, dr_march_bwd, dr_tf_momentum_step): no host round trip in the loop."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender.utils import get_tf, in_circles  # noqa: E402
from differender_amd import functional as F  # noqa: E402
from examples.render_nondiff_synthetic import synthetic_volume  # noqa: E402

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--vol", type=int, default=128)
    ap.add_argument("--img", type=int, default=256)
    ap.add_argument("--tf-res", type=int, default=128)
    ap.add_argument("--iterations", type=int, default=100)
    ap.add_argument("--lr", type=float, default=0.5)          # EX.py's defaults are tuned to its dataset
    ap.add_argument("--lr-decay", type=float, default=0.995)
    ap.add_argument("--mom", type=float, default=0.8)
    ap.add_argument("--clip-grads", type=float, default=1.0)
    ap.add_argument("--bw-sampling-rate", type=float, default=1.0)
    args = ap.parse_args()
    dev = torch.device("cuda")
    N, R, WH, S = args.vol, args.tf_res, (args.img, args.img), 1 << 20
    vol = synthetic_volume(N, dev)[0].permute(2, 0, 1)         # (1,D,H,W) -> field order (W,D,H), a strided view
    tf_gt = get_tf("tf1", R).t().contiguous().float().to(dev)  # (R,4)
    tf = get_tf("gray", R).t().contiguous().float().to(dev)
    mom = torch.zeros_like(tf)
    cam = in_circles(1.7).float().to(dev)[None]
    sr = args.bw_sampling_rate
    # only the TF is optimised: the forward leaves a per-sample tape (DR_TAPE_TF) and the backward never touches the volume again
    ws = F.alloc_workspace(1, WH, vol.shape, R, dev, tape=(S, sr))
    e, x, r, n = F.ray_setup(cam, WH, vol.shape, sr)
    ref, _ = F.march_fwd(vol, tf_gt, cam, e, x, r, n, S, sr, workspace=ws)
    ref = ref.clone()
    lr, losses = args.lr, []
    for i in range(args.iterations):
        e, x, r, n = F.ray_setup(cam, WH, vol.shape, sr, jitter_seed=F.new_jitter_seed())
        out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, S, sr, workspace=ws, tape=True)
        loss, g = F.mse_loss_grad(out, ref)
        _, d_tf = F.march_bwd(vol, tf, cam, e, x, r, n, S, sr, g, out, want_vol=False, workspace=ws, tape=True)
        F.tf_momentum_step(tf, d_tf, mom, lr, args.mom, args.clip_grads)
        lr *= args.lr_decay
        losses.append(loss)                                      # stays on the device until the loop is done
        if i % 20 == 0 or i == args.iterations - 1:
            print(f"[{i:05d}] loss {float(loss):.6f}  lr {lr:.3f}  max|dTF| {float(d_tf.abs().max()):.2e}")
    print(f"loss {float(losses[0]):.6f} -> {float(losses[-1]):.6f}; "
          f"TF error {float((tf - tf_gt).abs().mean()):.4f}")
    assert float(losses[-1]) < float(losses[0])
