"""One seed of the fuzzer's d4 profile: fast path against the sequential kernels, where they differ most, how many rays the exact pass
took (and, under a -DDR_D4_DEBUG build, the bound of the worst ray).   FUZZ_PROFILE=d4 python tools/d4_fuzz_seed.py seed..."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_argv = sys.argv; sys.argv = sys.argv[:1]
import fuzz_parity as fz
from differender_amd import functional as Fn
sys.argv = _argv
T = fz.T
dbg = bool(Fn.N.lib().dr_build_flags() & 1)
trace = None
args = sys.argv[1:]
if len(args) >= 3 and args[1] == "trace":   # d4_fuzz_seed.py SEED trace PIXEL_INDEX  (debug build: per-segment trace of that ray)
    trace = int(args[2]); args = args[:1]
for seed in map(int, args):
    c = fz.make_case(seed)
    vol = T(c["vol"].astype(np.float16)) if c["f16"] else T(c["vol"])
    tf, cam = T(c["tf"]), T(c["cam"])
    WH, vshape, sr, S, mode = c["WH"], c["vshape"], c["sr"], c["S"], c["mode"]
    e, x, r, n = Fn.ray_setup(cam, WH, vshape, sr, jitter_seed=c["jitter"])
    ws = Fn.alloc_workspace(c["n_views"], WH, vshape, c["R"], fz.dev)
    ws[64:68].view(torch.int32)[0] = -1 if trace is None else trace
    out, st = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, mode, workspace=ws, hints=0)
    ref, sref = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, mode, variant=1)
    stats = Fn.workspace_stats(ws)
    if dbg:
        o = out[0].cpu().numpy()
        print(f"seed {seed}: DEBUG build: max bound {o[..., 2].max():.3e}; rays with bound > 3e-6: {(o[..., 2] > 3e-6).sum()}")
        w = int(np.argmax(o[..., 2])); print("  worst-bound pixel", np.unravel_index(w, o.shape[:2]), o.reshape(-1, 4)[w])
        continue
    d = (out - ref).abs().amax(-1)[0]
    w = int(torch.argmax(d)); i, j = w // WH[1], w % WH[1]
    print(f"seed {seed} {fz.describe(c)}")
    print(f"  max |fast - seq| {float(d.max()):.3e} at pixel ({i},{j}); steps equal {bool(torch.equal(st, sref))}; exact rays {int(stats[15])}, per-ray fallback {int(stats[2])}, steps there {int(st[0, i, j])} of n {int(n[0, i, j])}")
    print("  fast", out[0, i, j].tolist(), "seq", ref[0, i, j].tolist())
    ws[64:68].view(torch.int32)[0] = i * WH[1] + j
