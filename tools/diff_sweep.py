"""Differential sweep at real sizes, GPU only: the fast path against the sequential kernels (DR_VARIANT_BASELINE -- the oracle's float32
recurrence bit for bit, which tests/test_gpu_parity.py checks) over random cameras (outside, near and inside the volume), every shipped
TF preset + the bench TF + tf1 with 1e-6 in its transparent ranges, sampling rates 0.7-4, both march modes, f32 / f16 storage; forward
(sample counts equal, RGBA within 1e-5) and, for the differentiable march, backward (d_volume and d_tf within 1e-4 of their maxima;
also over the per-sample tape for the TF alone).
    python tools/diff_sweep.py [seconds=240] [N=256] [IMG=256] [seed=0]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from differender_amd import functional as F  # noqa: E402
from differender_amd.utils import get_tf  # noqa: E402

dev = torch.device("cuda:0")
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 240.0
N = int(sys.argv[2]) if len(sys.argv) > 2 else 256
IMG = int(sys.argv[3]) if len(sys.argv) > 3 else 256
rng = np.random.default_rng(int(sys.argv[4]) if len(sys.argv) > 4 else 0)
R = 128
vols = {"blobs": bench.synth_volume_torch(N, dev)}
_ax = torch.linspace(-1.0, 1.0, N, device=dev)   # CT-like (bench.py --scene ct): the field inside a ball, air (exactly 0) outside
_r2 = _ax[:, None, None] ** 2 + _ax[None, :, None] ** 2 + _ax[None, None, :] ** 2
vols["ct"] = torch.where(_r2 < 0.36, vols["blobs"], torch.zeros_like(vols["blobs"]))
del _r2


def tf_of(name):
    if name == "bench":
        return bench.bench_tf_torch(R, 1e-3, dev)
    t = get_tf("tf1" if name == "d4" else name, R).t().contiguous().float().to(dev)
    if name == "d4":
        t[:, 3] = torch.where(t[:, 3] == 0, torch.full_like(t[:, 3], 1e-6), t[:, 3])
    return t


names = ["tf1", "tf2", "tf3", "tf4", "tf5", "black", "gray", "bench", "d4"]
if os.environ.get("SWEEP_TFS"):   # e.g. SWEEP_TFS=d4,tf2
    names = os.environ["SWEEP_TFS"].split(",")
t0, cases, bad, n_adj = time.time(), 0, [], 0
while time.time() - t0 < budget:
    vname = str(rng.choice(list(vols)))
    vol = vols[vname]
    f16 = rng.random() < 0.2
    v = vol.half() if f16 else vol
    name = str(rng.choice(names))
    tf = tf_of(name)
    mode = int(rng.random() < 0.35)
    sr = float(rng.choice([0.7, 1.0, 1.0, 2.0, 4.0] if mode == 0 else [1.0, 4.0, 8.0]))
    kind = rng.random()
    d = rng.normal(size=3); d /= np.linalg.norm(d)
    radius = 2.7 if kind < 0.5 else (1.2 + rng.random() if kind < 0.8 else 0.6 * rng.random())
    cam = torch.tensor([d * radius], dtype=torch.float32, device=dev)
    seed = int(rng.integers(1, 1 << 30)) if rng.random() < 0.5 else 0
    S = 1 << 20 if rng.random() < 0.7 else 1024
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), v.shape, sr, jitter_seed=seed)
    want_tape = mode == 0 and rng.random() < 0.3
    ws = F.alloc_workspace(1, (IMG, IMG), v.shape, R, dev, tape=(S, sr) if want_tape else None)
    out, steps = F.march_fwd(v, tf, cam, e, x, r, n, S, sr, mode=mode, workspace=ws, tape=want_tape)
    ref, sref = F.march_fwd(v, tf, cam, e, x, r, n, S, sr, mode=mode, variant=F.N.DR_VARIANT_BASELINE)
    desc = dict(vol=vname, f16=f16, tf=name, mode=mode, sr=sr, radius=round(float(radius), 3), jitter=seed, S=S, tape=want_tape)
    fails = []
    if not torch.equal(steps, sref):
        fails.append("steps differ on %d pixels" % int((steps != sref).sum()))
    dmax = float((out - ref).abs().max())
    if not dmax <= 1e-5:
        fails.append("rgba %.3e" % dmax)
    if mode == 0:
        g = torch.randn(1, IMG, IMG, 4, device=dev)
        if want_tape:
            _, dt = F.march_bwd(v, tf, cam, e, x, r, n, S, sr, g, out, want_vol=False, workspace=ws, tape=True)
            _, dt0 = F.march_bwd(v, tf, cam, e, x, r, n, S, sr, g, ref, want_vol=False, variant=F.N.DR_VARIANT_BASELINE)
            dv = dv0 = None
        else:
            dv, dt = F.march_bwd(v, tf, cam, e, x, r, n, S, sr, g, out, workspace=ws)
            dv0, dt0 = F.march_bwd(v, tf, cam, e, x, r, n, S, sr, g, ref, variant=F.N.DR_VARIANT_BASELINE)
        et = float((dt - dt0).abs().max() / dt0.abs().max().clamp_min(1e-30))
        if not et <= 1e-4:
            fails.append("d_tf %.3e" % et)
        if dv is not None:
            ev = float((dv.float() - dv0.float()).abs().max() / dv0.float().abs().max().clamp_min(1e-30))
            if not ev <= 1e-4:
                fails.append("d_vol %.3e" % ev)
    cases += 1
    if fails and os.environ.get("SWEEP_ADJUDICATE") and name != "d4" and n_adj < int(os.environ["SWEEP_ADJUDICATE"]):
        n_adj += 1
        # who is right? the float32 CPU oracle on the same ray buffers (slow: seconds per case at 256^3)
        from oracle import oracle as O
        vh = v.float().cpu().numpy(); th = tf.cpu().numpy(); ch = cam[0].cpu().numpy()
        eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
        oo, so = O.march_fwd(vh, th, ch, eh, xh, rh, nh, S, sr, mode)
        print("  oracle: rgba fast %.3e base %.3e; steps fast!=o %d base!=o %d" % (
            float(np.abs(out[0].cpu().numpy() - oo).max()), float(np.abs(ref[0].cpu().numpy() - oo).max()),
            int((steps[0].cpu().numpy() != so).sum()), int((sref[0].cpu().numpy() != so).sum())), flush=True)
        if mode == 0:
            gv, gt = O.march_bwd(vh, th, ch, eh, xh, rh, nh, S, sr, g[0].cpu().numpy(), want_vol=dv is not None)
            if dv is not None:
                a, b = dv.float().cpu().numpy(), dv0.float().cpu().numpy()
                i = np.unravel_index(np.argmax(np.abs(a - b)), a.shape)
                print("  d_vol vs oracle: fast %.3e base %.3e (of max %.3e); worst fast-base voxel %s fast %.6e base %.6e oracle %.6e; nan fast/base/oracle %d %d %d; inf %d %d %d; max|fast| %.3e max|base| %.3e" % (
                    float(np.nanmax(np.abs(a - gv)) / np.nanmax(np.abs(gv))), float(np.nanmax(np.abs(b - gv)) / np.nanmax(np.abs(gv))), float(np.nanmax(np.abs(gv))), i,
                    a[i], b[i], gv[i], int(np.isnan(a).sum()), int(np.isnan(b).sum()), int(np.isnan(gv).sum()),
                    int(np.isinf(a).sum()), int(np.isinf(b).sum()), int(np.isinf(gv).sum()), float(np.nanmax(np.abs(a))), float(np.nanmax(np.abs(b)))), flush=True)
            print("  d_tf vs oracle: fast %.3e base %.3e" % (float(np.abs(dt.cpu().numpy() - gt).max() / np.abs(gt).max()),
                                                            float(np.abs(dt0.cpu().numpy() - gt).max() / np.abs(gt).max())), flush=True)
    if fails:
        desc["cam"] = [round(float(c), 5) for c in cam[0].cpu().numpy()]
        bad.append((desc, fails))
        print("FAIL", desc, fails, flush=True)
    if cases % 20 == 0:
        print(f"... {cases} cases, {len(bad)} failing", flush=True)
print(f"diff sweep done: {cases} cases at {N}^3 / {IMG}^2 in {time.time() - t0:.0f} s, {len(bad)} failing")
sys.exit(1 if bad else 0)
