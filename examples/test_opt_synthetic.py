#!/usr/bin/env python3
"""Counterpart of the reference's examples/test_opt_tf.py (OPT.py:32-90) on synthetic data: reconstruct a
volume from batched views with the reference's DSSIM + MSE loss (OPT.py:70-72) against non-differentiable ground-truth
renders. Same call pattern (`raycast.raycast_nondiff(vol_gt, tf_gt, lf, sampling_rate=8.0)`, `raycast(vol, tf, lf)`,
AdamW + OneCycle, clamp to [0,1]); `ssim2d` is this repository's restatement of pytorch_msssim.ssim (not installed here,
semantics from memory: differender_amd/utils/losses.py); the plotting needs packages that are not installed here."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender.utils import get_tf, in_circles, get_rand_pos  # noqa: E402
from differender.volume_raycaster import Raycaster  # noqa: E402
from differender_amd.utils import ssim2d  # noqa: E402  (stands in for `from pytorch_msssim import ssim as ssim2d`, OPT.py:14)
from examples.render_nondiff_synthetic import synthetic_volume  # noqa: E402

if __name__ == "__main__":
    TF_RES, BS, ITERATIONS, N = 128, 8, int(os.environ.get("ITERS", "60")), int(os.environ.get("VOL", "128"))
    dev = torch.device("cuda")
    tf = get_tf("tf1", TF_RES)
    tf_gt = get_tf("tf1", TF_RES).to(dev).expand(BS, -1, -1).float()
    vol_gt = synthetic_volume(N, dev)
    vol = vol_gt.clone()
    mask = torch.rand_like(vol) < 0.05
    vol[mask] = torch.rand_like(vol[mask])
    raycast = Raycaster(vol.shape[-3:], (256, 256), TF_RES, jitter=True, max_samples=1024)
    vol = vol.float().requires_grad_(True)
    tf = tf.to(dev).float().requires_grad_(True)
    opt = torch.optim.AdamW([vol], weight_decay=0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=ITERATIONS)
    first = last = None
    for i in range(ITERATIONS):
        lf = torch.cat([in_circles(0.1 * i)[None], get_rand_pos(BS - 1)], dim=0).float().to(dev)
        with torch.no_grad():
            gt = raycast.raycast_nondiff(vol_gt.detach(), tf_gt.detach(), lf.detach(), sampling_rate=8.0)
        opt.zero_grad()
        res = raycast(vol, tf, lf)
        dssim_loss = 1.0 - ssim2d(res, gt, data_range=1.0, size_average=True, nonnegative_ssim=True)   # OPT.py:70
        mse_loss = F.mse_loss(res, gt)
        loss = torch.nan_to_num(dssim_loss) + mse_loss
        loss.backward()
        if i % 10 == 0 or i == ITERATIONS - 1:
            print(f"Step {i:03d}:   Loss: {loss.item():0.3f}   SSIM: {1.0 - dssim_loss.item():0.3f}   MSE: {mse_loss.item():0.5f}   "
                  f"LR: {sched.get_last_lr()[0]:.1e}   Vol Grad AbsMax: {vol.grad.abs().max():.1e}")
        first = loss.item() if first is None else first
        last = loss.item()
        opt.step(); sched.step()
        with torch.no_grad():
            tf.clamp_(0.0, 1.0); vol.clamp_(0.0, 1.0)
    print(f"loss {first:.6f} -> {last:.6f}")
