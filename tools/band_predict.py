#!/usr/bin/env python3
"""One-GPU predictor for `bench.py --split rows` (ONE view split into G bands of image rows, one band per GPU).

A rank of a G-GPU run renders only its band and needs no collective for that: this tool renders every band of
G = 1, 2, 4, 8 on ONE GPU, times forward and backward of each (HIP events, same kernels and arguments a rank would use),
and prints what the slowest band implies for strong scaling:  efficiency(G) = T(1) / (G * max_band T_band(G)),
without and with the gradient all-reduce of the SURVEY section-5 model added un-overlapped (512 MiB d_volume at 512^3).
Why it is not 1: every band stages every brick its rows' rays cross -- a brick's candidate rectangle is cut by the band,
its staging and listing are not -- and the per-launch fixed costs are paid by every rank.

  python tools/band_predict.py [--vol 512 --img 512 --reps 5]        (runs ON THE GPU BOX)
"""
import argparse
import json
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench as B                                   # synthetic scene of the headline
from differender_amd import functional as F
from differender_amd.distributed import row_work_estimate, shard_rows


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--vol", type=int, default=512)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--tf-res", type=int, default=256)
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--gs", default="1,2,4,8")
    ap.add_argument("--balance", default="rows", choices=["rows", "work"],
                    help="rows: bands of equal height; work: bands of equal estimated work (distributed.row_work_estimate)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    N, IMG, R = a.vol, a.img, a.tf_res
    vol = B.synth_volume_torch(N, dev)
    n_max = 2.0 * math.sqrt(3.0) * math.sqrt(3.0) * (N - 1)
    tf = B.bench_tf_torch(R, 3.0 / n_max, dev)
    cam_host = B.in_circles(0.3)
    cam = torch.tensor([cam_host], dtype=torch.float32, device=dev)
    weights = row_work_estimate(cam_host, IMG, IMG) if a.balance == "work" else None
    gen = torch.Generator(device="cpu").manual_seed(4321)
    target = torch.rand((1, IMG, IMG, 4), generator=gen).to(dev)
    S, sr = 1 << 20, 1.0
    res = {}
    for G in [int(g) for g in a.gs.split(",")]:
        bands = []
        for rank in range(G):
            row0, rows = shard_rows(IMG, rank, G, weights=weights)
            rows_arg = (row0, IMG) if G > 1 else None
            ws = F.alloc_workspace(1, (rows, IMG), (N, N, N), R, dev)
            e, x, r, n = F.ray_setup(cam, (rows, IMG), (N, N, N), sr, rows=rows_arg)
            tgt = target[:, row0:row0 + rows].contiguous()
            tf_ms, tb_ms, steps = [], [], 0
            for k in range(2 + a.reps):
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
                ev[0].record()
                out, st = F.march_fwd(vol, tf, cam, e, x, r, n, S, sr, workspace=ws, rows=rows_arg)
                ev[1].record()
                _, g = F.mse_loss_grad(out, tgt)
                ev[2].record()
                F.march_bwd(vol, tf, cam, e, x, r, n, S, sr, g, out, workspace=ws, rows=rows_arg)
                ev[3].record()
                torch.cuda.synchronize()
                if k >= 2:
                    tf_ms.append(ev[0].elapsed_time(ev[1])); tb_ms.append(ev[2].elapsed_time(ev[3]))
                steps = int(st.sum())
            bands.append({"rank": rank, "row0": row0, "rows": rows, "fwd_ms": float(np.median(tf_ms)),
                          "bwd_ms": float(np.median(tb_ms)), "voxel_steps": steps})
            del ws
        res[G] = bands
    t1 = res[1][0]["fwd_ms"] + res[1][0]["bwd_ms"] if 1 in res else None
    grad_bytes = N ** 3 * 4 + R * 16
    print(f"# band predictor: {N}^3 f32 volume, {IMG}^2 image, one view in G row bands of equal {'height' if a.balance == 'rows' else 'estimated work'}, camera in_circles(0.3); ms per band (median of {a.reps})")
    summary = {}
    for G, bands in res.items():
        slow = max(b["fwd_ms"] + b["bwd_ms"] for b in bands)
        tot_steps = sum(b["voxel_steps"] for b in bands)
        line = {"G": G, "slowest_band_ms": round(slow, 3), "sum_of_bands_ms": round(sum(b["fwd_ms"] + b["bwd_ms"] for b in bands), 3),
                "voxel_steps": tot_steps}
        if t1:
            line["efficiency_no_allreduce"] = round(t1 / (G * slow), 3)
            m = B.allreduce_model(grad_bytes, G)
            if m:
                line["efficiency_allreduce_all_links_exposed"] = round(t1 / (G * (slow + m["all_links_ms"])), 3)
                line["efficiency_allreduce_one_link_exposed"] = round(t1 / (G * (slow + m["one_link_ring_ms"])), 3)
                line["allreduce_model_ms"] = m
        summary[G] = line
        print(json.dumps(line))
        for b in bands:
            print(f"   G={G} rank {b['rank']}: rows [{b['row0']}, {b['row0'] + b['rows']}) fwd {b['fwd_ms']:.3f} bwd {b['bwd_ms']:.3f} "
                  f"ms, {b['voxel_steps']} voxel-steps")
    print("# weak scaling (--split views, the default of bench.py --gpus N): every rank runs the G = 1 line above on its own view;"
          " the only shared cost is the all-reduce, overlapped with the next step's forward")


if __name__ == "__main__":
    main()
