#!/usr/bin/env python3
"""Headline benchmark: Mvoxel-steps/s, forward + backward, 512^3 f32 volume @ 512^2 image (BASELINE.json).

One "step" = one pass of the hot path over one view per rank: ray setup -> forward march -> loss gradient
-> backward march (d_volume and d_tf) [-> RCCL all-reduce of the shared gradients when N > 1].
A voxel-step = one marched sample (sum of the per-pixel executed-step counters, VR.py:303,381).
Inputs are synthetic and resident in HBM before the timed region. Prints ONE JSON line on rank 0.

  python bench.py [--gpus N --steps K --warmup W]            (N > 1: launched by torch.distributed.run)
"""
import argparse
import json
import math
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
B_FWD, B_BWD_VOL, B_BWD_TF = 32.0, 96.0, 32.0  # algorithmic bytes per voxel-step


def synth_volume_torch(N, device, seed=1234):
    """Same field as oracle.synth_volume (gaussian blobs + falloff + ramp), generated on the device."""
    rng = np.random.RandomState(seed)
    centres = rng.uniform(-0.6, 0.6, size=(6, 3))
    sigmas = rng.uniform(0.15, 0.5, size=6)
    ax = torch.linspace(-1.0, 1.0, N, dtype=torch.float64, device=device)
    X, Y, Z = ax[:, None, None], ax[None, :, None], ax[None, None, :]
    acc = torch.zeros((N, N, N), dtype=torch.float32, device=device)
    for c, s in zip(centres, sigmas):
        gx = torch.exp(-((X - c[0]) ** 2) / (2 * s * s)).float()
        gy = torch.exp(-((Y - c[1]) ** 2) / (2 * s * s)).float()
        gz = torch.exp(-((Z - c[2]) ** 2) / (2 * s * s)).float()
        acc += gx * gy * gz
    r2 = (X * X + Y * Y + Z * Z).float()
    ramp = ((X + 2 * Y + 3 * Z) / 6.0).float()
    v = 0.2 + 0.6 * acc / acc.max() - 0.03 * r2 + 0.02 * ramp
    return v.clamp_(0.0, 1.0)


def bench_tf_torch(R, alpha, device):
    i = torch.arange(R, dtype=torch.float64) / max(R - 1, 1)
    tf = torch.empty((R, 4), dtype=torch.float64)
    for k, phi in enumerate((0.0, 2.1, 4.2)):
        tf[:, k] = 0.5 + 0.5 * torch.sin(2 * math.pi * i + phi)
    tf[:, 3] = alpha
    return tf.float().to(device)


def in_circles(i, y=0.7, dist=2.5):
    return [math.cos(i) * dist, y, math.sin(i) * dist]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--vol", type=int, default=512)
    ap.add_argument("--img", type=int, default=512)
    ap.add_argument("--tf-res", type=int, default=256)
    ap.add_argument("--grads", default="vol+tf", choices=["vol+tf", "tf", "vol", "none"],
                    help="vol+tf = C4 (default); tf = C3; none = forward only (C2-style)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 baseline kernels")
    ap.add_argument("--tf", default="bench", choices=["bench", "tf1"],
                    help="bench: constant alpha (no early termination, the headline); tf1: the reference's preset "
                         "(UT.py:9-21) -- empty ranges and early termination, reported separately")
    ap.add_argument("--views", type=int, default=1, help="views per rank per step (one native batched launch)")
    ap.add_argument("--vol-dtype", default="f32", choices=["f32", "f16"], help="volume storage (arithmetic is f32 either way; C5 uses f16)")
    ap.add_argument("--jitter", action="store_true", help="jittered ray starts (C5)")
    ap.add_argument("--split", default="views", choices=["views", "rows"],
                    help="N > 1: 'views' = one view per rank per step (weak scaling, the default); 'rows' = ONE view per "
                         "step split into N bands of image rows (strong scaling, SURVEY 8(e))")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-img", type=int, default=224, help="image edge of the bounded CPU-baseline sample")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)  # (% ndev only matters for the 1-GPU rehearsal below)
    dev = torch.device("cuda", local_rank % ndev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        # DR_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a 1-GPU box (ranks share the card);
        # the real run uses RCCL ("nccl" on ROCm) over xGMI.
        backend = os.environ.get("DR_BENCH_BACKEND", "nccl")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from differender_amd import functional as F
    from differender_amd.distributed import all_reduce_gradients

    N, IMG, R = args.vol, args.img, args.tf_res
    want_vol = args.grads in ("vol+tf", "vol")
    want_tf = args.grads in ("vol+tf", "tf")
    want_bwd = want_vol or want_tf
    # n_max <= 2*sqrt(3)*diag ~ 3.47*N*sqrt(3): alpha = 3/n_max keeps early termination from firing
    n_max = 2.0 * math.sqrt(3.0) * math.sqrt(3.0) * (N - 1)
    alpha = 3.0 / n_max
    vol = synth_volume_torch(N, dev)
    if args.vol_dtype == "f16":
        vol = vol.half()
    tf = bench_tf_torch(R, alpha, dev)
    if args.tf == "tf1":
        from differender_amd.utils import get_tf
        tf = get_tf("tf1", R).t().contiguous().to(dev)
    gen = torch.Generator(device="cpu").manual_seed(4321)
    V = args.views
    from differender_amd.distributed import shard_rows
    bands = args.split == "rows" and world > 1
    row0, ROWS = shard_rows(IMG, rank, world) if bands else (0, IMG)
    rows_arg = (row0, IMG) if bands else None
    target = torch.rand((1, IMG, IMG, 4), generator=gen).to(dev)[:, row0:row0 + ROWS].expand(V, ROWS, IMG, 4).contiguous()
    loss_acc = torch.zeros((), dtype=torch.float64, device=dev)
    S = 1 << 20  # tape-free: no depth limit needed
    sr = 1.0
    total_steps = torch.zeros((), dtype=torch.int64, device=dev)
    planned_steps = torch.zeros((), dtype=torch.int64, device=dev)  # sum of sample_step_nums (VR.py:259)
    ev = {"fwd": [], "bwd": []}
    # scratch of the brick-centric kernels (coarse tape); allocated once, reused by every step
    ws = F.alloc_workspace(V, (ROWS, IMG), (N, N, N), R, dev) if args.variant == 0 else None

    # all camera positions are uploaded before the timed region (a host->device copy inside the loop would
    # synchronise the stream every step)
    nstep_total = args.warmup + args.steps
    cams_all = torch.tensor([[in_circles(0.1 * ((k if bands else k * world + rank) * V + i)) for i in range(V)]
                             for k in range(nstep_total)], dtype=torch.float32, device=dev)

    def step(k, timed):
        cam = cams_all[k]
        e, x, r, n = F.ray_setup(cam, (ROWS, IMG), (N, N, N), sr, rows=rows_arg, jitter_seed=(42 if args.jitter else 0),
                                 view_base=k * V)
        # (warm-up steps run the very same host code, events included: their first use has a one-time host cost)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, S, sr, variant=args.variant, workspace=ws, rows=rows_arg)
        a1.record()
        if timed:
            ev["fwd"].append((a0, a1))
        if want_bwd:
            _, grad_out = F.mse_loss_grad(out, target, loss=loss_acc)  # loss + d(loss)/d(out) in one pass
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, S, sr, grad_out, out, want_vol=want_vol, want_tf=want_tf,
                                 variant=args.variant, workspace=ws, rows=rows_arg)
            b1.record()
            if timed:
                ev["bwd"].append((b0, b1))
            if world > 1:
                # RCCL sum of the shared gradients on its own stream: it overlaps the next step's forward; the
                # previous step's reduction is awaited first so at most one is in flight (and all before timing ends)
                for h in pending:
                    h.wait()
                pending[:] = all_reduce_gradients([g for g in (dv, dt) if g is not None], async_op=True)
                keep_alive[:] = [dv, dt]
        if timed:
            total_steps.add_(steps.sum()); planned_steps.add_(n.sum())
        else:
            steps.sum(); n.sum()  # same launches as a timed step

    pending, keep_alive = [], []

    def barrier():
        for h in pending:
            h.wait()
        pending[:] = []
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    barrier()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(args.warmup + k, True)
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_steps, op=dist.ReduceOp.SUM)
        dist.all_reduce(planned_steps, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    vsteps = int(total_steps.item())
    passes = 2 if want_bwd else 1  # a voxel-step counted once per marched sample of the fwd(+bwd) pass

    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"]]))
    bwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"]])) if ev["bwd"] else 0.0
    steps_per_launch = vsteps / world / max(args.steps, 1)

    def roof(name, ms, bytes_per_step):
        ach = steps_per_launch * bytes_per_step / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
        return {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(ach / HBM_PEAK_GBS, 5), "traffic": None, "avg_launch_ms": round(ms, 4),
                "bytes_per_voxel_step": bytes_per_step, "voxel_steps_per_launch": int(steps_per_launch)}

    # HBM bytes per launch from the latest committed PMC profile (collected with tools/profile_round.sh in
    # separate --pmc passes: FETCH_SIZE, WRITE_SIZE). FETCH_SIZE counts 128-B fabric requests at 64 B on gfx950: it is
    # doubled, as the MI355X guide prescribes. Calibrated on this access pattern (tools/microbench/fetch_calib.hip,
    # profiles/r01_fetch_calibration.txt): a 512 MiB stream of 16-B loads reads exactly 1/2, and the brick staging
    # pattern (15-float rows, 4 B per lane) tallies its 128-B line requests the same way. Memory-side requests
    # include infinity-cache hits, so this is an upper bound on HBM bytes.
    traffic = {}
    try:
        pm = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_per_launch.json")))
        for k, v in pm.items():
            if "brick_flat_kernel" in k and "FETCH_SIZE" in v:
                targs = [t.strip(" >") for t in k.split("<")[1].split(",")]  # <VT, MODE, BWD, VOL, TF, ALPHA>
                if len(targs) >= 6 and targs[5] == "true":
                    continue  # the (gated) alpha pre-pass
                traffic["bwd" if targs[2] == "true" else "fwd"] = int((2.0 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0)) * 1024)
    except Exception:
        pass
    roof_fwd = roof("march_fwd", fwd_ms, B_FWD)
    roof_bwd = roof("march_bwd", bwd_ms, B_BWD_VOL if want_vol else B_BWD_TF) if want_bwd else None
    if N == 512 and IMG == 512 and args.variant == 0:
        roof_fwd["traffic"] = traffic.get("fwd")
        if roof_bwd and want_vol and want_tf:
            roof_bwd["traffic"] = traffic.get("bwd")
    dominant = roof_bwd if (roof_bwd and bwd_ms >= fwd_ms) else roof_fwd

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(vol, tf, args, sr, want_vol, want_tf)

    if dist is not None:
        dist.destroy_process_group()
    if rank == 0:
        workload = {"vol+tf": "C4: fwd+bwd w.r.t. volume and TF", "tf": "C3: fwd+bwd w.r.t. TF",
                    "vol": "fwd+bwd w.r.t. volume", "none": "forward only"}[args.grads]
        line = {
            "metric": "Mvoxel-steps/s fwd+bwd, 512^3 vol @ 512^2 img",
            "value": round(vsteps / elapsed / 1e6, 3),
            "unit": "Mvoxel-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
            "higher_is_better": True, "scaling": "strong" if bands else "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{workload}; {N}^3 {args.vol_dtype} volume, {IMG}^2 image, {R}-entry TF, sr=1.0, "
                                   f"{V} view(s) per rank per step, orbit cameras in_circles(0.1*v), jitter {'on' if args.jitter else 'off'}",
                       "volume": N, "image": IMG, "tf_res": R, "views_per_rank_per_step": V,
                       "parallelism": (f"one view in {world} row bands" if bands else f"view-sharded x{world}") +
                                      (" + RCCL all-reduce(d_vol,d_tf)" if world > 1 else ""),
                       "passes_per_voxel_step": passes, "kernel_variant": args.variant, "tf": args.tf},
            "voxel_steps_per_step": int(vsteps / max(args.steps, 1)),
            "planned_steps_per_step": int(int(planned_steps.item()) / max(args.steps, 1)),  # executed/planned < 1 = early termination
            "roofline": dominant, "roofline_fwd": roof_fwd, "roofline_bwd": roof_bwd,
            "rays_marched_individually": (int(F.workspace_stats(ws)[0]) if ws is not None else None),
            "cpu_baseline": cpu_baseline,
        }
        print(json.dumps(line))


def run_cpu_baseline(vol, tf, args, sr, want_vol, want_tf):
    """Oracle (kind 'port': this repo's C restatement, not Taichi) timed on the host cores on a bounded
    sample of the same workload: same volume/TF/camera, smaller image."""
    from oracle import oracle as O
    O.build()
    cores = os.cpu_count() or 1
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        pass
    cores = min(cores, 16)  # a 1-GPU box's CPU share (the host exposes all its cores to every box)
    os.environ["OMP_NUM_THREADS"] = str(cores)
    vol_h = vol.cpu().numpy()
    tf_h = tf.cpu().numpy()
    cam = np.array(in_circles(0.0), np.float32)
    W = args.cpu_img
    N = args.vol
    # untimed warm-up on a 16x16 image: the first parallel region pays for the OpenMP thread pool and first touches
    ew, xw, rw, nw = O.ray_setup(cam, 16, 16, (N, N, N), sr)
    ow, _ = O.march_fwd(vol_h, tf_h, cam, ew, xw, rw, nw, 1 << 20, sr, 0)
    if want_vol or want_tf:
        O.march_bwd(vol_h, tf_h, cam, ew, xw, rw, nw, 1 << 20, sr, np.ones_like(ow), want_vol, want_tf)
    e, x, r, n = O.ray_setup(cam, W, W, (N, N, N), sr)
    t0 = time.perf_counter()
    out, steps = O.march_fwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, 0)
    if want_vol or want_tf:
        g = (2.0 / out.size) * (out - 0.5)
        O.march_bwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, g.astype(np.float32), want_vol, want_tf)
    dt = time.perf_counter() - t0
    nst = int(steps.sum())
    res = {"value": round(nst / dt / 1e6, 4), "unit": "Mvoxel-steps/s", "cores": cores, "kind": "port",
           "sample": f"same {N}^3 volume/TF, camera in_circles(0), {W}x{W} image "
                     f"({nst} voxel-steps, fwd{'+bwd' if (want_tf or want_vol) else ''}), C oracle with OpenMP, {dt:.1f} s"}
    # single-thread figure on a 16x smaller sample (SURVEY 8(d) asks for both)
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        W1 = max(W // 4, 8)
        e, x, r, n = O.ray_setup(cam, W1, W1, (N, N, N), sr)
        t0 = time.perf_counter()
        out, steps = O.march_fwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, 0)
        if want_vol or want_tf:
            g = (2.0 / out.size) * (out - 0.5)
            O.march_bwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, g.astype(np.float32), want_vol, want_tf)
        dt1 = time.perf_counter() - t0
        gomp.omp_set_num_threads(cores)
        res["value_1thread"] = round(int(steps.sum()) / dt1 / 1e6, 4)
        res["sample_1thread"] = f"{W1}x{W1} image, {int(steps.sum())} voxel-steps, {dt1:.1f} s"
    except OSError:
        pass
    return res


if __name__ == "__main__":
    main()
