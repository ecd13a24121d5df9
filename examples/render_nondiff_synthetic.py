#!/usr/bin/env python3
"""Counterpart of the reference's examples/render_nondiff.py (ND.py:15-29) on a synthetic volume: one
non-differentiable render (BASELINE config C1: 64^3 volume, 128^2 image) written as a PPM (no torchvision here)."""
import math
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender.utils import get_tf, in_circles  # noqa: E402
from differender.volume_raycaster import Raycaster  # noqa: E402


def synthetic_volume(n, device):
    ax = torch.linspace(-1, 1, n, device=device)
    z, y, x = torch.meshgrid(ax, ax, ax, indexing="ij")
    r = (x * x + y * y + z * z).sqrt()
    shell = torch.exp(-((r - 0.55) / 0.08) ** 2) * 0.4 + torch.exp(-((r - 0.25) / 0.1) ** 2) * 0.3
    return (0.1 + shell + 0.05 * torch.sin(9 * x) * torch.sin(7 * y)).clamp(0, 1)[None]  # (1, D, H, W)


if __name__ == "__main__":
    dev = torch.device("cuda")
    vol = synthetic_volume(64, dev)
    tf = get_tf("tf1", 128).to(dev)
    raycaster = Raycaster(vol.shape[-3:], (128, 128), 128, jitter=False, max_samples=1)
    lf = in_circles(1.7 * math.pi).float().to(dev)
    im = raycaster.raycast_nondiff(vol[None], tf[None], lf[None], sampling_rate=16.0)  # (1, 4, H, W)
    rgb = (im[0, :3].clamp(0, 1) * 255).byte().permute(1, 2, 0).cpu().contiguous()
    with open("nondiff_render.ppm", "wb") as f:
        f.write(b"P6 %d %d 255\n" % (rgb.shape[1], rgb.shape[0]) + rgb.numpy().tobytes())
    print("wrote nondiff_render.ppm", tuple(im.shape), "alpha max", float(im[0, 3].max()))
