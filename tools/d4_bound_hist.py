"""Histogram of the per-ray D4 bound (a -DDR_D4_DEBUG build: the image holds the bound of the worst channel in its third component):
how many rays of a view lie between the image budget (3e-6) and candidate thresholds for the backward (DR_D4_BWD_BUDGET).
    DIFFERENDER_ALLOW_EXPERIMENT=1 DIFFERENDER_HIP_LIB=ab_libs/d4dbg.so python tools/d4_bound_hist.py"""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd())
import bench
from differender_amd import functional as F
from differender_amd.utils import get_tf
dev = torch.device("cuda:0")
N, IMG, R = 512, 512, 256
vol = bench.synth_volume_torch(N, dev)
edges = [3e-6, 4e-6, 5e-6, 7e-6, 1e-5, 2e-5, 1e-4, 1.0]
for name in ("bench", "tf1", "d4"):
    tf = bench.bench_tf_torch(R, 1e-3, dev) if name == "bench" else get_tf("tf1", R).t().contiguous().float().to(dev)
    if name == "d4":
        tf[:, 3] = torch.where(tf[:, 3] == 0, torch.full_like(tf[:, 3], 1e-6), tf[:, 3])
    for v in range(6):
        cam = torch.tensor([bench.in_circles(0.1 * v)], dtype=torch.float32, device=dev)
        e, x, r, n = F.ray_setup(cam, (IMG, IMG), vol.shape, 1.0)
        ws = F.alloc_workspace(1, (IMG, IMG), vol.shape, R, dev)
        out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
        b = out[0, :, :, 2].flatten().cpu().numpy()
        h = [int(((b > lo) & (b <= hi)).sum()) for lo, hi in zip(edges[:-1], edges[1:])]
        print(name, "camera", v, "rays with a bound in", dict(zip(["3-4e-6", "4-5e-6", "5-7e-6", "7e-6-1e-5", "1-2e-5", "2e-5-1e-4", ">1e-4"], h)), flush=True)
