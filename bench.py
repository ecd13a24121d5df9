#!/usr/bin/env python3
"""Headline benchmark: Mvoxel-steps/s, forward + backward, 512^3 f32 volume @ 512^2 image (BASELINE.json).

One "step" = one pass of the hot path over one view per rank: ray setup -> forward march -> loss gradient
-> backward march (d_volume and d_tf) [-> RCCL all-reduce of the shared gradients when N > 1].
A voxel-step = one marched sample (sum of the per-pixel executed-step counters, VR.py:303,381).
Inputs are synthetic and resident in HBM before the timed region. Prints ONE JSON line on rank 0.

  python bench.py [--gpus N --steps K --warmup W]

N > 1: either launched by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment, what the
driver does), or started plainly -- then this process spawns the N ranks itself, before touching the GPU, and
relays rank 0's line. On a box with fewer than N GPUs the spawned ranks share the card(s) and talk over gloo (a
rehearsal of the control flow, said so in the line); on a real node they use RCCL ("nccl") over xGMI.

  python bench.py --workload opt     the reference's demo loop (examples/test_opt_tf.py:33-88) through Raycaster
"""
import argparse
import csv
import glob
import json
import math
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec


def algorithmic_bytes(vol_dtype):
    """Algorithmic bytes per voxel-step, from the volume's STORAGE type: the forward
    gathers the 8 corners of the centre cell (8 x sizeof(voxel)); the backward w.r.t. the TF re-marches (the same);
    the backward w.r.t. the volume adds the read-modify-write of the centre cell's 8 f32 gradients (8 x 4 B x 2).
    f32: 32 / 32 / 96; f16: 16 / 16 / 80. With the per-sample tape (DR_TAPE_TF) the forward also writes TAPE_BYTES per sample and the
    TF-only backward reads those and no voxel: forward + 8, backward 8."""
    b_fwd = 8.0 * {"f32": 4, "f16": 2}[vol_dtype]
    return b_fwd, b_fwd + 64.0, b_fwd   # forward, backward w.r.t. volume (+TF), backward w.r.t. TF only


TAPE_BYTES = 8.0   # (intensity, lighting term) of a marched sample: two floats


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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default="march", choices=["march", "opt"],
                    help="march: the headline (C4 and its variants below); opt: the reference's demo loop "
                         "(examples/test_opt_tf.py:33-88: 256^3, 256^2, 8 views, tf1, jittered fwd+bwd + sr-8 ground-truth "
                         "render + AdamW step per iteration) through the drop-in Raycaster module")
    ap.add_argument("--vol", type=int, default=None, help="volume edge (default 512; opt: 256)")
    ap.add_argument("--img", type=int, default=None, help="image edge (default 512; opt: 256)")
    ap.add_argument("--tf-res", type=int, default=None, help="TF entries (default 256; opt: 128)")
    ap.add_argument("--grads", default="vol+tf", choices=["vol+tf", "tf", "vol", "none"],
                    help="vol+tf = C4 (default); tf = C3; none = forward only (C2-style)")
    ap.add_argument("--variant", type=int, default=0, help="0 auto, 1 baseline kernels")
    ap.add_argument("--tf", default="bench", choices=["bench", "tf1"],
                    help="bench: constant alpha (no early termination, the headline); tf1: the reference's preset "
                         "(UT.py:9-21) -- empty ranges and early termination, reported separately")
    ap.add_argument("--views", type=int, default=None, help="views per rank per step (one native batched launch; opt: 8)")
    ap.add_argument("--vol-dtype", default="f32", choices=["f32", "f16"], help="volume storage (arithmetic is f32 either way; C5 uses f16)")
    ap.add_argument("--jitter", action="store_true", help="jittered ray starts (C5)")
    ap.add_argument("--scene", default="blobs", choices=["blobs", "ct"],
                    help="blobs: the headline's field (no voxel is 0); ct: the same field inside a ball of radius 0.6, exactly "
                         "0 (air) outside it -- with --tf tf1, whose alpha is 0 below intensity 0.084, three bricks in four hold "
                         "nothing but transparent samples (a CT-like scene; reported separately)")
    ap.add_argument("--cam", default="orbit", choices=["orbit", "inside"], help="inside: camera inside the volume (every ray starts behind the eye)")
    ap.add_argument("--split", default="views", choices=["views", "rows"],
                    help="N > 1: 'views' = one view per rank per step (weak scaling, the default); 'rows' = ONE view per "
                         "step split into N bands of image rows (strong scaling, SURVEY 8(e))")
    ap.add_argument("--rccl-channels", type=int, default=None,
                    help="N > 1: cap RCCL at this many channels (NCCL_MAX_NCHANNELS / NCCL_MIN_NCHANNELS, set before the "
                         "communicator is created). Each channel is a workgroup on a CU: fewer channels leave more CUs to "
                         "the forward march the gradient all-reduce overlaps with, at a lower all-reduce bandwidth")
    ap.add_argument("--hints", default="auto", choices=["auto", "off"],
                    help="caller hints of dr_march_fwd (DR_HINT_*): 'auto' = derived from the TF's largest alpha once it is "
                         "known (functional._TerminationHints, no host sync); 'off' = never")
    ap.add_argument("--dssim", action="store_true", help="(default since round 6; kept so that older command lines still parse)")
    ap.add_argument("--no-dssim", action="store_true",
                    help="opt workload: MSE only instead of the reference's DSSIM + MSE (examples/test_opt_tf.py:70-72). The DSSIM "
                         "term runs through the torch restatement of pytorch_msssim.ssim (differender_amd/utils/losses.py): +2.7 ms "
                         "per iteration of torch convolutions outside the raycasting path -- rounds 1-5 quoted the loop without it "
                         "(tools/abn_opt.sh still does: same-device A/B of the raycasting path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-tape", action="store_true", help="--grads tf: the brick-centric TF-only backward instead of the per-sample tape (DR_TAPE_TF)")
    ap.add_argument("--cpu-img", type=int, default=None,
                    help="image edge of the CPU-baseline sample (default: the whole view, capped at 512: ~20 s on 16 cores)")
    ap.add_argument("--pmc", default="auto", choices=["auto", "live", "off"],
                    help="roofline.traffic: 'live' collects FETCH_SIZE / WRITE_SIZE with rocprofv3 around short child runs of "
                         "this very command; 'auto' = live for the default single-GPU headline, else off")
    ap.add_argument("--pmc-json", default=None, help="use this per-launch counter file (tools/profile_round.sh) instead")
    return ap.parse_args(argv)


# ------------------------------------------------------------------------------------------------ N > 1 self-launch
def self_launch(args):
    """--gpus N without a launcher: start the N ranks as fresh child processes (this process has not touched the GPU;
    torch.cuda.device_count() does not initialise it) and relay rank 0's JSON line."""
    import socket
    n = args.gpus
    ndev = torch.cuda.device_count()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if ndev < n and "DR_BENCH_BACKEND" not in env:
        env["DR_BENCH_BACKEND"] = "gloo"  # rehearsal: ranks share the card(s)
        print(f"[bench] {ndev} GPU(s) for {n} ranks: rehearsing over gloo on shared devices", file=sys.stderr)
    procs, logs = [], []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r), OMP_NUM_THREADS=env.get("OMP_NUM_THREADS", "4"))
        # rank 0's stdout carries the JSON line (a pipe would have to be drained while polling: a temporary file instead)
        logs.append(tempfile.TemporaryFile() if r == 0 else None)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=e,
                                      stdout=logs[r] if r == 0 else subprocess.DEVNULL))
    # Poll ALL ranks: when one dies (RCCL init failure, a fault) the others sit in a collective for ever -- end them at
    # once instead of waiting for each in turn.
    rcs = [None] * n
    failed = None
    while any(rc is None for rc in rcs):
        for r, p in enumerate(procs):
            if rcs[r] is None:
                rcs[r] = p.poll()
                if rcs[r] not in (None, 0) and failed is None:
                    failed = r
        if failed is not None:
            print(f"[bench] rank {failed} exited with code {rcs[failed]}: stopping the other ranks", file=sys.stderr)
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    p.terminate()
            deadline = time.time() + 10.0
            for r, p in enumerate(procs):
                if rcs[r] is None:
                    try:
                        rcs[r] = p.wait(timeout=max(deadline - time.time(), 0.1))
                    except subprocess.TimeoutExpired:
                        p.kill()
                        rcs[r] = p.wait()
            break
        time.sleep(0.05)
    logs[0].seek(0)
    out = logs[0].read()
    if failed is None:
        for ln in out.decode().splitlines():      # rank 0's JSON line (its stdout carries nothing else: claim_stdout)
            if ln.startswith("{"):
                sys.stdout.write(ln + "\n")
        sys.stdout.flush()
    return max(abs(rc) for rc in rcs)


# ------------------------------------------------------------------------------------------------ live HBM counters
PROFILER_ENV_PREFIXES = ("ROCP_", "ROCPROF", "ROCPROFILER_", "HSA_TOOLS_", "ROCTRACER_")


def under_profiler(env=None):
    """True if this process was started under rocprofv3 / rocprof (their preload or tool variables are set)."""
    env = os.environ if env is None else env
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(PROFILER_ENV_PREFIXES) for k in env)


# One rocprofv3 pass per tuple (MI355X guide, "rocprofv3 PMC slots": TCC has 4 slots -- FETCH_SIZE costs 3, WRITE_SIZE 2, so
# they cannot share a pass; SQ has 8 slots, GRBM 2 of its own). Counter passes run with --kernel-trace only.
PMC_PASSES = (("FETCH_SIZE",), ("WRITE_SIZE",),
              ("SQ_INSTS_VALU", "SQ_INSTS_LDS", "SQ_WAVE_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_LDS_BANK_CONFLICT",
               "SQ_LDS_IDX_ACTIVE", "SQ_WAVES", "GRBM_GUI_ACTIVE"))


def measure_traffic_live(argv_workload):
    """roofline.traffic and roofline_valu, measured in THIS job: rocprofv3 --pmc around short child runs of the same
    command, one pass per entry of PMC_PASSES, --kernel-trace only.
    Returns ({kernel: {counter: average per launch, "launches": n}}, note)."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    # Never nest profilers: under `rocprofv3 -- python3 bench.py` this process carries the profiler's preload, a child
    # rocprofv3 (a python script that exec's its target) would inherit it, and the preloaded tool initialises the GPU
    # before that exec -- the exec of a GPU-initialised process that takes a box down.
    if under_profiler():
        return None, "already running under a profiler: live counters skipped"
    res = {}
    work = tempfile.mkdtemp(prefix="dr_pmc_", dir="/tmp")
    env = dict(os.environ, TMPDIR="/tmp")
    for k in list(env):
        if k in ("RANK", "LOCAL_RANK", "WORLD_SIZE") or k.startswith(PROFILER_ENV_PREFIXES):
            env.pop(k, None)
    if "rocprof" in env.get("LD_PRELOAD", "").lower():
        env.pop("LD_PRELOAD", None)
    failed = []
    try:
        for counters in PMC_PASSES:
            out = os.path.join(work, counters[0])
            cmd = [exe, "--pmc", *counters, "--kernel-trace", "--output-format", "csv", "-d", out, "--",
                   sys.executable, os.path.abspath(__file__), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                   "--pmc", "off"] + argv_workload
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=420)
            files = glob.glob(os.path.join(out, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                if counters[0] in ("FETCH_SIZE", "WRITE_SIZE"):
                    return None, f"rocprofv3 --pmc {counters[0]} failed (rc {p.returncode})"
                failed.append(counters[0])   # the instruction counters are an extra: the traffic figures stand without them
                continue
            agg, cnt = {}, {}
            for row in csv.DictReader(open(files[0])):
                if "dr::" not in row["Kernel_Name"] or row["Counter_Name"] not in counters:
                    continue
                k = (row["Kernel_Name"].split("(")[0].replace("void ", ""), row["Counter_Name"])
                agg[k] = agg.get(k, 0.0) + float(row["Counter_Value"])
                cnt[k] = cnt.get(k, 0) + 1
            for (k, c) in agg:
                res.setdefault(k, {})[c] = agg[(k, c)] / cnt[(k, c)]
                res[k]["launches"] = cnt[(k, c)]
    except (subprocess.TimeoutExpired, OSError) as ex:
        return None, f"live PMC pass failed: {ex}"
    finally:
        shutil.rmtree(work, ignore_errors=True)
    note = "rocprofv3 --pmc, one pass per counter group (" + " | ".join(" ".join(c) for c in PMC_PASSES) + \
           ") around child runs of this command, same job"
    if failed:
        note += f"; pass(es) {failed} failed"
    return res, note


def _dominant_kernel(pm, want_bwd_kernel):
    for k, v in pm.items():
        if "brick_flat_kernel<" not in k:
            continue
        targs = [t.strip(" >") for t in k.split("<")[1].split(",")]  # <VT, MODE, BWD, VOL, TF, ALPHA, K>
        if len(targs) >= 6 and targs[5] not in ("false", "0"):
            continue  # the (gated) alpha pre-pass (a bool until round 4, the pre-pass's configuration 1 / 2 since)
        if (targs[2] == "true") == want_bwd_kernel:
            return k, v
    return None, None


# Issue cost of a VALU wave-instruction on gfx950, in SHADER CLOCKS (what GRBM_GUI_ACTIVE counts). tools/microbench/oprate_bench
# (profiles/r02_microbench_oprate.txt, four waves per SIMD) reports 2.6-3.0 "cycles" for f32 add / mul / fma, v_mov, v_add_u32,
# v_ashrrev and 4.3-4.7 for every other class (min / max, compares, conversions, v_fract, v_cndmask, DPP, shifts, 3-operand integer
# ops) -- in units of the NOMINAL 2.4 GHz clock. A 64-lane instruction of the second class occupies a 16-lane SIMD for exactly
# 4 clocks, so the chip ran that benchmark at 2.4 x 4 / 4.5 = 2.13 GHz (the counters put these kernels at 2.1-2.2 GHz too) and the
# costs in real clocks are 4.0 and 2.3-2.7 (2.4 for the kernels' mix of mov / add / mul / fma). Share of the first class in the
# hot loops, from the listings (tools/isa_hist.py, blocks of the per-sample loop + the per-pass scan): forward 0.70, backward 0.43.
VALU_COST_FAST, VALU_COST_SLOW = 2.4, 4.0
VALU_FAST_SHARE = {False: 0.70, True: 0.43}
N_SIMD, N_CU, N_XCD = 1024, 256, 8


def valu_roofline(pm, want_bwd_kernel, steps_per_launch):
    """What actually bounds the brick kernels: VALU issue slots and the LDS, from the SQ counters of
    the dominant kernel, per launch. SQ cycle counters tick once per 4 clocks per wave; SQ_LDS_* are LDS-array cycles summed
    over the CUs; GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
    k, v = _dominant_kernel(pm, want_bwd_kernel)
    if not v or "SQ_INSTS_VALU" not in v or not v.get("GRBM_GUI_ACTIVE"):
        return None
    clocks = v["GRBM_GUI_ACTIVE"] / N_XCD                       # kernel duration in shader clocks
    per_simd = v["SQ_INSTS_VALU"] / N_SIMD
    cyc_per_instr = clocks / per_simd
    share = VALU_FAST_SHARE[want_bwd_kernel]
    model = share * VALU_COST_FAST + (1.0 - share) * VALU_COST_SLOW
    wc = max(v.get("SQ_WAVE_CYCLES", 0.0), 1.0)
    return {"kernel": k, "bound": "valu-issue",
            "valu_lane_instr_per_voxel_step": round(v["SQ_INSTS_VALU"] * 64.0 / max(steps_per_launch, 1), 1),
            "valu_wave_instr_per_simd": int(per_simd), "kernel_clocks": int(clocks),
            "clocks_per_valu_instr": round(cyc_per_instr, 3),
            "issue_cost_model_clocks_per_instr": round(model, 3),
            "issue_slot_utilisation": round(min(model / cyc_per_instr, 1.0), 3),
            "issue_slot_utilisation_if_all_full_rate": round(VALU_COST_FAST / cyc_per_instr, 3),
            "lds_conflict_ratio": round(v.get("SQ_LDS_BANK_CONFLICT", 0.0) / max(v.get("SQ_LDS_IDX_ACTIVE", 0.0), 1.0), 3),
            "lds_array_busy": round(v.get("SQ_LDS_IDX_ACTIVE", 0.0) / N_CU / clocks, 3),
            "lds_instr_per_voxel_step": round(v.get("SQ_INSTS_LDS", 0.0) * 64.0 / max(steps_per_launch, 1), 1),
            "waves_per_simd": round(4.0 * wc / N_SIMD / clocks, 2),
            "wave_time_split": {"parked_waitcnt_or_barrier": round(v.get("SQ_WAIT_ANY", 0.0) / wc, 3),
                                "issue_stalled": round(v.get("SQ_WAIT_INST_ANY", 0.0) / wc, 3)},
            "note": "issue_slot_utilisation = (share of full-rate VALU x 2.4 + rest x 4.0 shader clocks: measured issue costs and "
                    "the static mix of the hot loop) / observed clocks per VALU wave-instruction per SIMD, capped at 1; "
                    "issue_slot_utilisation_if_all_full_rate is the floor no mix can go below; the HBM fractions in `roofline` "
                    "price a kernel that is bound HERE (and, in the backward, by its LDS atomics)"}


def observed_bound(vr, hbm_frac_measured):
    """Name of what limits a brick kernel according to ITS counters (roofline_valu block vr): VALU issue slots (>= 80 % full
    against the two-class issue-cost model), the LDS (array >= 45 % busy: in the backward that is its atomics), HBM only if the
    memory side is the busiest of the three. The HBM *model* fraction stays in `frac`."""
    parts = []
    if vr["issue_slot_utilisation"] >= 0.8:
        parts.append("valu-issue")
    if vr["lds_array_busy"] >= 0.45:
        targs = [t.strip(" >") for t in vr["kernel"].split("<")[1].split(",")]  # <VT, MODE, BWD, VOL, TF, ALPHA, K, NARROW>
        parts.append("lds-atomics" if len(targs) > 2 and targs[2] == "true" else "lds")
    if hbm_frac_measured is not None and hbm_frac_measured >= max(0.6, vr["issue_slot_utilisation"]):
        parts = ["hbm"]
    return "+".join(parts) if parts else "latency (no unit above 80 % / 45 % busy)"


def traffic_bytes(pm, want_bwd_kernel):
    """HBM-side bytes per launch of the forward / backward brick kernel from a per-launch counter table.
    FETCH_SIZE counts 128-B fabric requests at 64 B on gfx950: doubled, as the MI355X guide prescribes (calibrated on
    this access pattern: tools/microbench/fetch_calib.hip, profiles/r01_fetch_calibration.txt). WRITE_SIZE is exact for
    16-B stores and float atomics. Both are in KB. Memory-side requests include Infinity-Cache hits: an upper bound."""
    k, v = _dominant_kernel(pm, want_bwd_kernel)
    if v and "FETCH_SIZE" in v:
        return int((2.0 * v["FETCH_SIZE"] + v.get("WRITE_SIZE", 0.0)) * 1024)
    return None


def roofline_entry(name, ms, bytes_per_step, marched_per_launch, evaluated, key):
    """One `roofline*` object of the bench line. `achieved` / `frac` price the ALGORITHMIC bytes of the samples the
    kernel really EVALUATED: evaluated[key] / evaluated['marched'] of the marched count when the counting step ran (brick
    kernels), the marched count otherwise. A fraction above 1 would mean the numerator prices work the kernel did not do: it is
    refused (null + the reason), never printed. `bound` is None until counters of THIS run name it (observed_bound)."""
    ratio, kind = 1.0, "model: algorithmic bytes of the marched samples / time / HBM peak"
    if evaluated is not None and evaluated.get(key) is not None and evaluated.get("marched"):
        ratio = evaluated[key] / evaluated["marched"]
        kind = "model: algorithmic bytes of the EVALUATED samples / time / HBM peak"
        if ratio < 0.98:
            kind += f" (work-skipping active: {ratio:.3f} of the marched samples were evaluated)"
    per_launch = marched_per_launch * ratio
    ach = per_launch * bytes_per_step / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
    frac = ach / HBM_PEAK_GBS
    entry = {"kernel": name, "bound": None, "model_bound": "hbm",
             "achieved": round(ach, 2), "peak": HBM_PEAK_GBS, "unit": "GB/s",
             "frac": round(frac, 5), "frac_kind": kind,
             "traffic": None, "hbm_gbs_measured": None, "hbm_frac_measured": None,
             "avg_launch_ms": round(ms, 4), "bytes_per_voxel_step": bytes_per_step,
             "voxel_steps_per_launch": int(marched_per_launch),
             "evaluated_voxel_steps_per_launch": int(per_launch),
             "note": "achieved/frac price the ALGORITHMIC bytes (model); hbm_gbs_measured = traffic / avg_launch is what "
                     "the memory-side counters saw; bound: from the counters of this run, else null (unmeasured)"}
    if frac > 1.0:
        entry["achieved"] = entry["frac"] = None
        entry["frac_refused"] = (f"{frac:.2f} > 1: the timed launches did not do the work the numerator prices "
                                 "(samples skipped that the count still holds)")
    return entry


def allreduce_model(nbytes, world):
    """SURVEY section 5 cost model of the gradient all-reduce over xGMI (7 links x ~153 GB/s per GPU, point to point):
    a single ring is bound by one link, 2 (G-1)/G S / 153 GB/s; with all links in use (direct reduce-scatter + all-gather,
    S/G per peer and phase) 2 (G-1)/G S / (min(G-1, 7) x 153 GB/s). Reported beside the measured allreduce_ms."""
    if not nbytes or world < 2:
        return None
    link = 153e9
    vol = 2.0 * (world - 1) / world * nbytes
    return {"one_link_ring_ms": round(vol / link * 1e3, 3),
            "all_links_ms": round(vol / (min(world - 1, 7) * link) * 1e3, 3)}


# ------------------------------------------------------------------------------------------------ main
def init_dist(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)
    ndev = max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank % ndev)  # (% ndev only matters for the shared-card rehearsal)
    dev = torch.device("cuda", local_rank % ndev)
    dist, backend = None, None
    # DR_BENCH_FORCE_DIST=1 (with DR_ALLREDUCE_SINGLE_RANK=1): the whole N > 1 control flow -- RCCL communicator, overlapped
    # gradient all-reduce, barriers, max over ranks -- with the ONE rank a one-GPU box allows (a rehearsal, not a scaling number)
    if world > 1 or os.environ.get("DR_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        backend = os.environ.get("DR_BENCH_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm
        if getattr(args, "rccl_channels", None):
            os.environ["NCCL_MAX_NCHANNELS"] = str(args.rccl_channels)
            os.environ["NCCL_MIN_NCHANNELS"] = str(min(args.rccl_channels, int(os.environ.get("NCCL_MIN_NCHANNELS", "1"))))
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return world, rank, dev, dist, backend


_JSON_OUT = None


def claim_stdout():
    """The contract is ONE JSON line on stdout. RCCL prints a banner there when a communicator is created (C-level stdio),
    and any other library might: from here on file descriptor 1 is stderr, and the line goes to the descriptor kept here."""
    global _JSON_OUT
    if _JSON_OUT is None:
        sys.stdout.flush()
        _JSON_OUT = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
    return _JSON_OUT


def emit(line):
    out = claim_stdout()
    out.write(json.dumps(line) + "\n")
    out.flush()


def main():
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args))
    claim_stdout()
    if args.workload == "opt":
        return main_opt(args)
    args.vol = args.vol or 512
    args.img = args.img or 512
    args.tf_res = args.tf_res or 256
    args.views = args.views or 1
    world, rank, dev, dist, backend = init_dist(args)

    from differender_amd import functional as F
    from differender_amd.distributed import GradientReducer, all_reduce_gradients, shard_rows

    N, IMG, R = args.vol, args.img, args.tf_res
    want_vol = args.grads in ("vol+tf", "vol")
    want_tf = args.grads in ("vol+tf", "tf")
    want_bwd = want_vol or want_tf
    # n_max <= 2*sqrt(3)*diag ~ 3.47*N*sqrt(3): alpha = 3/n_max keeps early termination from firing
    n_max = 2.0 * math.sqrt(3.0) * math.sqrt(3.0) * (N - 1)
    alpha = 3.0 / n_max
    vol = synth_volume_torch(N, dev)
    if args.scene == "ct":
        ax = torch.linspace(-1.0, 1.0, N, device=dev)
        r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
        vol = torch.where(r2 < 0.36, vol, torch.zeros_like(vol))
        del r2
    if args.vol_dtype == "f16":
        vol = vol.half()
    tf = bench_tf_torch(R, alpha, dev)
    if args.tf == "tf1":
        from differender_amd.utils import get_tf
        tf = get_tf("tf1", R).t().contiguous().to(dev)
    gen = torch.Generator(device="cpu").manual_seed(4321)
    V = args.views
    bands = args.split == "rows" and world > 1
    # (one view for all ranks: the cameras are host numbers, so the per-row work can be estimated without touching the device;
    #  the bands of a step are cut for the FIRST timed camera and kept -- buffers and workspace keep their shapes)
    from differender_amd.distributed import row_work_estimate
    band_weights = row_work_estimate(in_circles(0.1 * args.warmup * args.views), IMG, IMG) if bands and args.cam == "orbit" else None
    row0, ROWS = shard_rows(IMG, rank, world, weights=band_weights) if bands else (0, IMG)
    rows_arg = (row0, IMG) if bands else None
    target = torch.rand((1, IMG, IMG, 4), generator=gen).to(dev)[:, row0:row0 + ROWS].expand(V, ROWS, IMG, 4).contiguous()
    loss_acc = torch.zeros((), dtype=torch.float64, device=dev)
    S = 1 << 20  # tape-free: no depth limit needed
    sr = 1.0
    total_steps = torch.zeros((), dtype=torch.int64, device=dev)
    planned_steps = torch.zeros((), dtype=torch.int64, device=dev)  # sum of sample_step_nums (VR.py:259)
    ev = {"fwd": [], "bwd": []}
    # scratch of the brick-centric kernels (coarse tape); allocated once, reused by every step
    # C3 (--grads tf): the forward leaves a per-sample tape of (intensity, lighting), the TF-only backward is a per-ray pass over it
    # (DR_TAPE_TF, csrc/tf_tape.hip) -- what RaycastFunction does when only the TF requires grad; --no-tape: the brick-centric backward
    use_tape = want_tf and not want_vol and args.variant == 0 and not args.no_tape
    ws = F.alloc_workspace(V, (ROWS, IMG), (N, N, N), R, dev, tape=(S, sr) if use_tape else None) if args.variant == 0 else None
    use_tape = use_tape and ws is not None

    # all camera positions are uploaded before the timed region (a host->device copy inside the loop would
    # synchronise the stream every step)
    nstep_total = args.warmup + args.steps

    def cam_of(v):
        if args.cam == "inside":  # a slow orbit INSIDE the box, looking at the origin
            return [0.45 * math.cos(0.1 * v), 0.2, 0.45 * math.sin(0.1 * v)]
        return in_circles(0.1 * v)

    cams_all = torch.tensor([[cam_of((k if bands else k * world + rank) * V + i) for i in range(V)]
                             for k in range(nstep_total)], dtype=torch.float32, device=dev)
    reducer = GradientReducer()

    def step(k, timed, sync_grads=False, count=False):
        """One pass of the hot path. sync_grads: the gradient all-reduce is AWAITED before the step ends (what an optimisation
        loop needs: backward -> optimiser step -> next forward, examples/test_opt_tf.py:73-86) instead of overlapping the next
        step's forward. count: measurement only -- the brick kernels count the samples they evaluate (DR_COUNT_EVALUATED)."""
        cam = cams_all[k]
        e, x, r, n = F.ray_setup(cam, (ROWS, IMG), (N, N, N), sr, rows=rows_arg, jitter_seed=(42 if args.jitter else 0),
                                 view_base=k * V)
        # (warm-up steps run the very same host code, events included: their first use has a one-time host cost)
        a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a0.record()
        out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, S, sr, variant=args.variant, workspace=ws, rows=rows_arg,
                                 hints=(F.N.DR_COUNT_EVALUATED if count else ("auto" if args.hints == "auto" else 0)), tape=use_tape)
        a1.record()
        if timed:
            ev["fwd"].append((a0, a1))
        dv = dt = None
        if want_bwd:
            _, grad_out = F.mse_loss_grad(out, target, loss=loss_acc)  # loss + d(loss)/d(out) in one pass
            b0, b1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            b0.record()
            dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, S, sr, grad_out, out, want_vol=want_vol, want_tf=want_tf,
                                 variant=args.variant, workspace=ws, rows=rows_arg, count_evaluated=count, tape=use_tape)
            b1.record()
            if timed:
                ev["bwd"].append((b0, b1))
            if dist is not None:
                # RCCL sum of the shared gradients on its own stream: it overlaps the next step's forward; the
                # previous step's reduction is awaited first so at most one is in flight (and all before timing ends)
                reducer.submit([dv, dt])
                if sync_grads:
                    reducer.wait()   # the reduced gradient is in hand before the next step starts
        if timed:
            total_steps.add_(steps.sum()); planned_steps.add_(n.sum())
        elif count:
            return int(steps.sum().item())
        else:
            steps.sum(); n.sum()  # same launches as a timed step
        return dv, dt

    def barrier():
        reducer.wait()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(args.warmup):
        step(k, False)
    barrier()
    t0 = time.perf_counter()
    last = (None, None)
    for k in range(args.steps):
        last = step(args.warmup + k, True)
    barrier()
    elapsed_local = time.perf_counter() - t0

    # N > 1: the number an OPTIMISATION loop would see, next to the overlapped one -- the same steps once more with every
    # step's gradient all-reduce awaited before the next forward starts (a throughput benchmark may hide the
    # exchange behind the next forward, backward -> optimiser step -> next forward cannot)
    ms_per_step_sync, sync_steps = None, 0
    if dist is not None and want_bwd:
        sync_steps = min(max(args.steps, 1), 10)
        barrier()
        ts = time.perf_counter()
        for k in range(sync_steps):
            step(args.warmup + k, False, sync_grads=True)
        barrier()
        el_sync = torch.tensor([time.perf_counter() - ts], dtype=torch.float64, device=dev)
        dist.all_reduce(el_sync, op=dist.ReduceOp.MAX)
        ms_per_step_sync = float(el_sync.item()) / sync_steps * 1e3

    # Samples the brick kernels EVALUATED, against the samples marched (the reference's unit of work, VR.py:303): one untimed
    # step under DR_COUNT_EVALUATED. The work-skipping paths (empty bricks, unlit segments) evaluate nothing for samples that
    # still count as marched, and a roofline priced with the marched count would credit a kernel with bytes it never touched
    # (profiles/r05_bench_lines.jsonl carried frac = 8.93 for a 0.34 ms backward).
    evaluated = None
    if ws is not None:
        barrier()
        try:
            marched_one = step(args.warmup + max(args.steps, 1) - 1, False, count=True)
            reducer.wait()
            torch.cuda.synchronize()
            ev_pre, ev_fwd, ev_bwd = F.evaluated_samples(ws)
            if use_tape:
                ev_bwd = marched_one   # (the pass over the tape evaluates every live sample, by construction; it keeps no counter)
            evaluated = {"marched": marched_one, "alpha_prepass": ev_pre, "march_fwd": ev_fwd,
                         "march_bwd": (ev_bwd if want_bwd else None)}
        except RuntimeError as exc:   # (a library of an earlier round named by DIFFERENDER_HIP_LIB for an A/B run: no such flag)
            print(f"[bench] no evaluated-sample count: {exc}", file=sys.stderr)

    # the exchange step on its own (not overlapped): K synchronous all-reduces of the gradient buffers
    allreduce_ms = None
    allreduce_bytes = None
    if dist is not None and want_bwd:
        grads = [g for g in last if g is not None]
        allreduce_bytes = int(sum(g.numel() * g.element_size() for g in grads))
        barrier()
        t1 = time.perf_counter()
        for _ in range(max(args.steps, 1)):
            all_reduce_gradients(grads)
        barrier()
        allreduce_ms = (time.perf_counter() - t1) / max(args.steps, 1) * 1e3

    # The overlap question of the first real multi-GPU run: in the timed loop every forward runs with
    # the previous step's gradient all-reduce in flight on RCCL's stream. The same forwards once more with nothing in
    # flight, per rank, so that ONE run shows what the collective's channels take from the march.
    fwd_alone_ms = None
    if dist is not None and want_bwd:
        barrier()
        pairs = []
        for k in range(min(max(args.steps, 1), 5)):
            cam = cams_all[args.warmup + k]
            e, x, r, n = F.ray_setup(cam, (ROWS, IMG), (N, N, N), sr, rows=rows_arg, jitter_seed=(42 if args.jitter else 0),
                                     view_base=(args.warmup + k) * V)
            a0, a1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a0.record()
            F.march_fwd(vol, tf, cam, e, x, r, n, S, sr, variant=args.variant, workspace=ws, rows=rows_arg,
                        hints=("auto" if args.hints == "auto" else 0), tape=use_tape)
            a1.record()
            pairs.append((a0, a1))
        barrier()
        fwd_alone_ms = float(np.mean([a.elapsed_time(b) for a, b in pairs]))

    el = torch.tensor([elapsed_local], dtype=torch.float64, device=dev)
    rank_ms = [elapsed_local / max(args.steps, 1) * 1e3]
    rank_phase_ms = None
    rank_counters = None
    if dist is not None:
        gathered = [torch.zeros_like(el) for _ in range(world)]
        dist.all_gather(gathered, el)
        rank_ms = [float(g.item()) / max(args.steps, 1) * 1e3 for g in gathered]
        mine = torch.tensor([float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"]])) if ev["fwd"] else 0.0,
                             fwd_alone_ms or 0.0,
                             float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"]])) if ev["bwd"] else 0.0],
                            dtype=torch.float64, device=dev)
        ph = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(ph, mine)
        rank_phase_ms = {"fwd_with_allreduce_in_flight": [round(float(t[0]), 4) for t in ph],
                         "fwd_alone": [round(float(t[1]), 4) for t in ph] if fwd_alone_ms is not None else None,
                         "bwd": [round(float(t[2]), 4) for t in ph]}
        # every rank's workspace counters after its last step (functional.workspace_stats): rays repaired by the count check, rays
        # marched one by one, wrong "no early termination" hints, backward calls that did not find their forward's tape -- a
        # mis-sharded run (bands that do not match their buffers, a stale workspace) shows up here, in the one line the driver keeps
        st_mine = (F.workspace_stats(ws)[:16].to(torch.int64).to(dev) if ws is not None else torch.zeros(16, dtype=torch.int64, device=dev))
        st_all = [torch.zeros_like(st_mine) for _ in range(world)]
        dist.all_gather(st_all, st_mine)
        rank_counters = {"rays_repaired": [int(t[0]) for t in st_all], "rays_marched_individually": [int(t[2]) for t in st_all],
                         "rays_recomputed_sequentially": [int(t[15]) for t in st_all],
                         "wrong_hint_views": [int(t[8]) for t in st_all], "stale_workspace_backwards": [int(t[9]) for t in st_all]}
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(total_steps, op=dist.ReduceOp.SUM)
        dist.all_reduce(planned_steps, op=dist.ReduceOp.SUM)
    elapsed = float(el.item())
    vsteps = int(total_steps.item())
    passes = 2 if want_bwd else 1  # a voxel-step counted once per marched sample of the fwd(+bwd) pass

    fwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["fwd"]]))
    bwd_ms = float(np.mean([a.elapsed_time(b) for a, b in ev["bwd"]])) if ev["bwd"] else 0.0
    steps_per_launch = vsteps / world / max(args.steps, 1)
    stats = F.workspace_stats(ws) if ws is not None else None

    if dist is not None:
        dist.destroy_process_group()
    if rank != 0:
        return

    def roof(name, ms, bytes_per_step, evaluated_key):
        return roofline_entry(name, ms, bytes_per_step, steps_per_launch, evaluated, evaluated_key)

    b_fwd, b_bwd_vol, b_bwd_tf = algorithmic_bytes(args.vol_dtype)
    if use_tape:   # the tape's bytes are algorithmic bytes of these two kernels, the voxels are not the backward's
        b_fwd, b_bwd_tf = b_fwd + TAPE_BYTES, TAPE_BYTES
    roof_fwd = roof("march_fwd", fwd_ms, b_fwd, "march_fwd")
    if evaluated is not None and evaluated.get("alpha_prepass") and fwd_ms > 0:
        # the timed forward also ran the alpha pre-pass (a centre tap per sample, live or behind a termination point): the same byte
        # model over everything the forward's kernels evaluated, beside the colour march's own figure
        both = (evaluated["alpha_prepass"] + evaluated["march_fwd"]) / max(evaluated["marched"], 1) * steps_per_launch
        roof_fwd["frac_incl_alpha_prepass"] = round(min(both * b_fwd / (fwd_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 1.0), 5)
        roof_fwd["evaluated_incl_alpha_prepass_per_launch"] = int(both)
    roof_bwd = roof("march_bwd", bwd_ms, b_bwd_vol if want_vol else b_bwd_tf, "march_bwd") if want_bwd else None

    # roofline.traffic: HBM-side bytes per launch of the dominant kernels
    pm, traffic_source = None, None
    default_cmd = world == 1 and args.variant == 0
    if args.pmc_json:
        pm = json.load(open(args.pmc_json)); traffic_source = f"counter file {args.pmc_json}"
    elif args.pmc == "live" or (args.pmc == "auto" and default_cmd and N == 512 and IMG == 512 and args.tf == "bench"
                                and args.grads == "vol+tf" and V == 1):
        wl = ["--vol", str(N), "--img", str(IMG), "--tf-res", str(R), "--grads", args.grads, "--tf", args.tf,
              "--views", str(V), "--vol-dtype", args.vol_dtype, "--cam", args.cam] + (["--jitter"] if args.jitter else [])
        pm, traffic_source = measure_traffic_live(wl)
    if pm:
        for rf, is_bwd in ((roof_fwd, False), (roof_bwd, True)):
            if rf is None:
                continue
            tb = traffic_bytes(pm, is_bwd)
            if tb:
                rf["traffic"] = tb
                rf["hbm_gbs_measured"] = round(tb / (rf["avg_launch_ms"] * 1e-3) / 1e9, 1)
                rf["hbm_frac_measured"] = round(rf["hbm_gbs_measured"] / HBM_PEAK_GBS, 4)
    dominant = roof_bwd if (roof_bwd and bwd_ms >= fwd_ms) else roof_fwd
    valu_fwd = valu_roofline(pm, False, steps_per_launch) if pm else None
    valu_bwd = valu_roofline(pm, True, steps_per_launch) if (pm and want_bwd) else None
    for rf, vr in ((roof_fwd, valu_fwd), (roof_bwd, valu_bwd)):
        if rf is not None and vr is not None:
            rf["bound"] = observed_bound(vr, rf.get("hbm_frac_measured"))

    cpu_baseline = None
    if world == 1 and not args.no_cpu_baseline:
        cpu_baseline = run_cpu_baseline(vol, tf, args, sr, want_vol, want_tf, gpu=(F, dev, ws))

    workload = {"vol+tf": "C4: fwd+bwd w.r.t. volume and TF", "tf": "C3: fwd+bwd w.r.t. TF",
                "vol": "fwd+bwd w.r.t. volume", "none": "forward only"}[args.grads]
    cams = "orbit cameras in_circles(0.1*v)" if args.cam == "orbit" else "cameras INSIDE the volume (radius 0.45)"
    line = {
        "metric": f"Mvoxel-steps/s {'fwd+bwd' if want_bwd else 'fwd'}, {N}^3 vol @ {IMG}^2 img",
        "value": round(vsteps / elapsed / 1e6, 3),
        "unit": "Mvoxel-steps/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed / max(args.steps, 1) * 1e3, 4),
        "higher_is_better": True, "scaling": "strong" if bands else "weak", "vs_baseline": None,
        "dtype": "f32" if args.vol_dtype == "f32" else "f32 arithmetic, f16 volume storage", "data": "synthetic",
        "config": {"workload": f"{workload}; {N}^3 {args.vol_dtype} volume, {IMG}^2 image, {R}-entry TF, sr=1.0, "
                               f"{V} view(s) per rank per step, {cams}, jitter {'on' if args.jitter else 'off'}",
                   "volume": N, "image": IMG, "tf_res": R, "views_per_rank_per_step": V,
                   "parallelism": (f"one view in {world} row bands" if bands else f"view-sharded x{world}") +
                                  (" + all-reduce(d_vol,d_tf)" if world > 1 else ""),
                   "backend": ({"nccl": "RCCL (nccl) over xGMI", "gloo": "gloo REHEARSAL: ranks share the card(s)"}.get(backend, backend)
                               if world > 1 else None),
                   "passes_per_voxel_step": passes, "kernel_variant": args.variant, "tf": args.tf, "scene": args.scene,
                   "tf_only_backward": (("per-sample tape (DR_TAPE_TF)" if use_tape else "brick-centric") if (want_tf and not want_vol) else None)},
        "voxel_steps_per_step": int(vsteps / max(args.steps, 1)),
        "planned_steps_per_step": int(int(planned_steps.item()) / max(args.steps, 1)),  # executed/planned < 1 = early termination
        "evaluated_voxel_steps": evaluated,   # one untimed step under DR_COUNT_EVALUATED: marched samples vs samples whose taps were evaluated
        "ms_per_step_sync": None if ms_per_step_sync is None else round(ms_per_step_sync, 4),   # N > 1: every step's all-reduce awaited before the next forward
        "sync_steps": sync_steps or None,
        "build_flags": int(F.N.lib().dr_build_flags()),   # 0 = the shipped kernels; 4 = built without a tuned compiler flag (csrc/Makefile)
        "ms_per_step_ranks": [round(v, 4) for v in rank_ms],
        "phase_ms_ranks": rank_phase_ms,   # N > 1: per-rank forward time with / without the gradient all-reduce in flight, backward
        "workspace_counters_ranks": rank_counters,   # N > 1: per-rank fallback counters (all zero / a few single-sample rays when healthy)
        "allreduce_ms": None if allreduce_ms is None else round(allreduce_ms, 4),
        "allreduce_bytes": allreduce_bytes,
        "allreduce_model_ms": allreduce_model(allreduce_bytes, world),
        "rccl_channels": getattr(args, "rccl_channels", None),
        "roofline": dominant, "roofline_fwd": roof_fwd, "roofline_bwd": roof_bwd,
        "roofline_valu": (valu_bwd if (valu_bwd and dominant is roof_bwd) else valu_fwd),
        "roofline_valu_fwd": valu_fwd, "roofline_valu_bwd": valu_bwd,
        "traffic_source": traffic_source,
        "rays_marched_individually": (int(stats[2]) if stats is not None else None),
        "rays_repaired": (int(stats[0]) if stats is not None else None),
        "rays_recomputed_sequentially": (int(stats[15]) if stats is not None else None),   # ray_exact_kernel
        "cpu_baseline": cpu_baseline,
    }
    emit(line)


# ------------------------------------------------------------------------------------------------ the demo loop
def main_opt(args):
    """examples/test_opt_tf.py:33-88 on synthetic data, through the drop-in module: per iteration 8 camera positions, a
    non-differentiable ground-truth render at sampling rate 8, the jittered differentiable render, MSE loss (--dssim: the
    reference's DSSIM + MSE, OPT.py:70-72, through this repository's torch restatement of pytorch_msssim.ssim, which is
    not installed here: +2.7 ms of torch convolutions per iteration), backward to volume and TF, AdamW + OneCycle step, clamp. This is what dropping the library into the reference's script costs end to end, host glue included."""
    N = args.vol or 256
    IMG = args.img or 256
    R = args.tf_res or 128
    BS = args.views or 8
    args.dssim = not args.no_dssim   # the reference's loss unless asked otherwise
    world, rank, dev, dist, backend = init_dist(args)
    from differender.utils import get_tf, in_circles as ic, get_rand_pos
    from differender.volume_raycaster import Raycaster
    from differender_amd.utils import dssim_mse_loss

    torch.manual_seed(1234 + rank)
    field = synth_volume_torch(N, dev)
    if args.scene == "ct":   # CT-like: the field inside a ball of radius 0.6, air (exactly 0) outside -- mostly empty under tf1
        ax = torch.linspace(-1.0, 1.0, N, device=dev)
        r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
        field = torch.where(r2 < 0.36, field, torch.zeros_like(field))
        del r2
    vol_gt = field.permute(1, 2, 0).contiguous()[None]   # (1, D, H, W)
    del field
    vol = vol_gt.clone()
    mask = torch.rand_like(vol) < 0.05                                         # OPT.py:44-45
    vol[mask] = torch.rand_like(vol[mask])
    tf = get_tf("tf1", R)
    tf_gt = get_tf("tf1", R).to(dev).expand(BS, -1, -1).float()
    raycast = Raycaster(vol.shape[-3:], (IMG, IMG), R, jitter=True, max_samples=1024)   # OPT.py:49
    vol = vol.float().requires_grad_(True)
    tf = tf.to(dev).float().requires_grad_(True)
    total = args.warmup + args.steps
    opt = torch.optim.AdamW([vol], weight_decay=0)
    sched = torch.optim.lr_scheduler.OneCycleLR(opt, max_lr=1e-3, total_steps=max(total, 2))
    nsteps = torch.zeros((), dtype=torch.int64, device=dev)
    ev = {k: [] for k in ("gt", "fwd", "bwd", "opt")}
    losses = []

    def mark():
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def iteration(i, timed):
        lf = torch.cat([ic(0.1 * (i * world + rank))[None], get_rand_pos(BS - 1)], dim=0).float().to(dev)  # OPT.py:65
        t0 = mark()
        with torch.no_grad():
            gt = raycast.raycast_nondiff(vol_gt.detach(), tf_gt.detach(), lf.detach(), sampling_rate=8.0)
        gt_steps = raycast.vr._steps.sum()
        t1 = mark()
        opt.zero_grad()
        res = raycast(vol, tf, lf)
        fw_steps = raycast.vr._steps.sum()
        t2 = mark()
        if not args.dssim:
            loss = torch.nn.functional.mse_loss(res, gt)
        else:   # OPT.py:70-72 (ssim2d: this repository's restatement of pytorch_msssim.ssim, differender_amd/utils/losses.py)
            loss = dssim_mse_loss(res, gt)[0]
        loss.backward()
        if dist is not None:
            from differender_amd.distributed import all_reduce_gradients
            all_reduce_gradients([vol.grad, tf.grad])
        t3 = mark()
        opt.step()
        sched.step()
        with torch.no_grad():
            tf.clamp_(0.0, 1.0)
            vol.clamp_(0.0, 1.0)
        t4 = mark()
        if timed:
            nsteps.add_(gt_steps + 2 * fw_steps)
            for k, a, b in (("gt", t0, t1), ("fwd", t1, t2), ("bwd", t2, t3), ("opt", t3, t4)):
                ev[k].append((a, b))
            losses.append(loss.detach())

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        iteration(i, False)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        iteration(args.warmup + i, True)
    barrier()
    elapsed = time.perf_counter() - t0
    el = torch.tensor([elapsed], dtype=torch.float64, device=dev)
    if dist is not None:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        dist.all_reduce(nsteps, op=dist.ReduceOp.SUM)
        dist.destroy_process_group()
    if rank != 0:
        return
    elapsed = float(el.item())
    ms = {k: float(np.mean([a.elapsed_time(b) for a, b in v])) for k, v in ev.items()}
    gpu_ms = sum(ms.values())
    it_ms = elapsed / max(args.steps, 1) * 1e3
    line = {
        "metric": "iterations/s of the reference's optimisation demo (examples/test_opt_tf.py:63-88) through Raycaster",
        "value": round(args.steps * world / elapsed, 4), "unit": "iterations/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(it_ms, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": {"scene": args.scene, "loss": "DSSIM + MSE" if args.dssim else "MSE",
                   "workload": f"OPT demo: {N}^3 f32 volume ({'CT-like: air outside a ball; ' if args.scene == 'ct' else ''}5 % voxels randomised), {IMG}^2 image, {BS} views per iteration "
                               f"(1 orbit + {BS - 1} random, r=2.7), tf1 with {R} entries, max_samples=1024, jitter on; per "
                               "iteration: nondiff GT render at sr 8, differentiable render at sr 1, " + ("DSSIM + MSE" if args.dssim else "MSE") + ", backward to volume "
                               "and TF, AdamW + OneCycleLR step, clamp",
                   "volume": N, "image": IMG, "tf_res": R, "views": BS},
        "mvoxel_steps_per_s": round(int(nsteps.item()) / elapsed / 1e6, 1),
        "voxel_steps_per_iteration": int(int(nsteps.item()) / max(args.steps, 1) / world),
        "ms_gt_render": round(ms["gt"], 4), "ms_forward": round(ms["fwd"], 4), "ms_loss_backward": round(ms["bwd"], 4),
        "ms_optimiser": round(ms["opt"], 4),
        "ms_host_overhead": round(it_ms - gpu_ms, 4),   # wall time per iteration not covered by the GPU phases above
        "loss_first_last": [round(float(losses[0]), 6), round(float(losses[-1]), 6)] if losses else None,
    }
    emit(line)


def host_cores():
    """Threads the CPU baseline uses: the cores this process may run on (scheduler affinity), cut to the cgroup's CPU
    quota when there is one (a container's share of a bigger host) -- what the box really gives, no cap by fiat."""
    try:
        n_aff = len(os.sched_getaffinity(0))
    except AttributeError:
        n_aff = os.cpu_count() or 1
    quota = None
    try:   # cgroup v2: "max 100000" or "<quota> <period>"
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max":
            quota = float(q) / float(per)
    except (OSError, ValueError):
        try:   # cgroup v1
            q = float(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            per = float(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                quota = q / per
        except (OSError, ValueError):
            pass
    n = n_aff if quota is None else max(1, min(n_aff, int(math.ceil(quota))))
    note = f"{n_aff} cores in the affinity mask" + ("" if quota is None else f", cgroup CPU quota {quota:.1f}")
    return n, note


def run_cpu_baseline(vol, tf, args, sr, want_vol, want_tf, gpu=None):
    """Oracle (kind 'port': this repo's C restatement, not Taichi) timed on the host cores on a bounded
    sample of the same workload: same volume/TF/camera, smaller image. With `gpu` = (functional, device, workspace) the
    same view is rendered and differentiated through the C ABI as well and compared with what the oracle just computed
    (`parity_vs_gpu`): the checker at work, outside every timed region."""
    from oracle import oracle as O
    O.build()
    cores, cores_note = host_cores()
    os.environ["OMP_NUM_THREADS"] = str(cores)
    vol_h = vol.float().cpu().numpy()
    tf_h = tf.cpu().numpy()
    cam = np.array(in_circles(0.0), np.float32)
    W = args.cpu_img or min(args.img, 512)
    N = args.vol
    # untimed warm-up on a 16x16 image: the first parallel region pays for the OpenMP thread pool and first touches
    ew, xw, rw, nw = O.ray_setup(cam, 16, 16, (N, N, N), sr)
    ow, _ = O.march_fwd(vol_h, tf_h, cam, ew, xw, rw, nw, 1 << 20, sr, 0)
    if want_vol or want_tf:
        O.march_bwd(vol_h, tf_h, cam, ew, xw, rw, nw, 1 << 20, sr, np.ones_like(ow), want_vol, want_tf)
    e, x, r, n = O.ray_setup(cam, W, W, (N, N, N), sr)
    t0 = time.perf_counter()
    out, steps = O.march_fwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, 0)
    dv_ref = dt_ref = g = None
    if want_vol or want_tf:
        g = ((2.0 / out.size) * (out - 0.5)).astype(np.float32)
        dv_ref, dt_ref = O.march_bwd(vol_h, tf_h, cam, e, x, r, n, 1 << 20, sr, g, want_vol, want_tf)
    dt = time.perf_counter() - t0
    nst = int(steps.sum())
    parity = None
    if gpu is not None and W == args.img:
        # the same view through the C ABI (the bench's own volume, TF and kernel variant), against the oracle's results above
        Fm, dev, ws = gpu
        cam_t = torch.tensor(cam[None], dtype=torch.float32, device=dev)
        eg, xg, rg, ng = Fm.ray_setup(cam_t, (W, W), (N, N, N), sr)
        og, sg = Fm.march_fwd(vol, tf, cam_t, eg, xg, rg, ng, 1 << 20, sr, variant=args.variant, workspace=ws)
        parity = {"view": "camera in_circles(0), whole image",
                  "ray_buffers_bit_exact": bool(np.array_equal(ng[0].cpu().numpy(), n) and np.array_equal(rg[0].cpu().numpy(), r)),
                  "sample_counts_equal": bool(np.array_equal(sg[0].cpu().numpy(), steps)),
                  "rgba_max_abs_err": float(np.abs(og[0].cpu().numpy() - out).max())}
        if g is not None:
            dvg, dtg = Fm.march_bwd(vol, tf, cam_t, eg, xg, rg, ng, 1 << 20, sr, torch.from_numpy(g[None]).to(dev), og,
                                    want_vol=want_vol, want_tf=want_tf, variant=args.variant, workspace=ws)
            if dvg is not None:
                parity["d_volume_max_err_over_max"] = float(np.abs(dvg.cpu().numpy() - dv_ref).max() / max(np.abs(dv_ref).max(), 1e-30))
                parity["d_volume_voxels_compared"] = int(dv_ref.size)
            if dtg is not None:
                parity["d_tf_max_err_over_max"] = float(np.abs(dtg.cpu().numpy() - dt_ref).max() / max(np.abs(dt_ref).max(), 1e-30))
            del dvg, dtg
        parity["within_north_star_bars"] = bool(parity["sample_counts_equal"] and parity["rgba_max_abs_err"] <= 1e-5 and
                                                parity.get("d_volume_max_err_over_max", 0.0) <= 1e-4 and
                                                parity.get("d_tf_max_err_over_max", 0.0) <= 1e-4)
    del dv_ref
    whole = "the WHOLE view" if W == args.img else f"a {W}x{W} rendering of the {args.img}x{args.img} view"
    res = {"value": round(nst / dt / 1e6, 4), "unit": "Mvoxel-steps/s", "cores": cores, "kind": "port",
           "cores_note": cores_note,
           "sample": f"same {N}^3 volume/TF, camera in_circles(0), {W}x{W} image = {whole} "
                     f"({nst} voxel-steps, fwd{'+bwd' if (want_tf or want_vol) else ''}), C oracle with OpenMP, {dt:.1f} s",
           "parity_vs_gpu": parity}
    # single-thread figure on a 16x smaller sample (SURVEY 8(d) asks for both)
    try:
        import ctypes
        gomp = ctypes.CDLL("libgomp.so.1")
        gomp.omp_set_num_threads(1)
        W1 = max(W // 8, 8)
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
