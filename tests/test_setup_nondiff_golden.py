"""Ray setup and the non-differentiable march against an independent witness: tests/golden/setup_nondiff_*.npz hold
entry / exit / rays / sample counts and RGBA computed by a float64 numpy transliteration of compute_entry_exit
(VR.py:221-259, :127-151, :28-53) and of raycast_nondiff + get_final_image_nondiff (VR.py:308-361)
(tests/golden/make_setup_nondiff_golden.py) -- code that shares nothing with oracle/ or the kernels, and whose march runs
on its OWN ray buffers. The f64 oracle must agree to rounding, the f32 oracle and the HIP path within the parity
tolerances (forward 1e-5; sample counts exact wherever the floor() argument is not within rounding of an integer)."""
import glob
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLDEN = sorted(glob.glob(os.path.join(HERE, "golden", "setup_nondiff_*.npz")))
IDS = [os.path.basename(p)[len("setup_nondiff_"):-4] for p in GOLDEN]
FWD_TOL = 1e-5
RAY_TOL, T_TOL = 3e-6, 1e-5     # f32 ray directions / entry and exit distances against the float64 vectors
# f32 RGBA against the float64 vectors: the normal is a difference of trilinear taps 2e-3 world units (0.015 voxels of
# these 16^3-24^3 volumes) apart -- a few 1e-5 relative in f32 -- and the nondiff lighting is not clamped
F32_VS_F64_TOL = 5e-5


def test_fixtures_are_there():
    assert len(GOLDEN) == 5


def _hit(d):
    return d["n"] > 0


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_oracle_ray_setup_matches_the_transliteration(oracle, path):
    d = np.load(path)
    W, H = d["n"].shape
    ok, hit = d["ok_setup"], _hit(d)
    assert ok.mean() > 0.99
    # float64 oracle: same arithmetic up to the order of a few operations
    e, x, r, n = oracle.ray_setup(d["cam"], W, H, d["vol"].shape, sr=float(d["sr"]), dtype=np.float64)
    assert np.array_equal(n[ok], d["n"][ok])
    assert np.abs(r - d["rays"]).max() <= 1e-12
    assert np.abs(e - d["entry"])[hit].max() <= 1e-11 and np.abs(x - d["exit"])[hit].max() <= 1e-11
    # float32 oracle (the parity oracle): counts exact where f32 rounding cannot move the floor, buffers to f32 rounding
    e4, x4, r4, n4 = oracle.ray_setup(d["cam"].astype(np.float32), W, H, d["vol"].shape, sr=float(d["sr"]))
    same = n4 == d["n"]
    assert same[~hit].all()                            # a miss is a miss
    assert same.mean() >= 0.97, float(same.mean())     # a count may differ by one where the floor argument is within f32 rounding
    assert np.abs(n4.astype(np.int64) - d["n"]).max() <= 1
    # (f32 evaluation of the pinhole geometry -- a 0.1-unit near plane, three normalisations -- is ~1e-6 off in direction)
    assert np.abs(r4 - d["rays"]).max() <= RAY_TOL
    assert (np.abs(e4 - d["entry"])[hit] <= T_TOL * np.maximum(np.abs(d["entry"][hit]), 1.0)).all()
    assert (np.abs(x4 - d["exit"])[hit] <= T_TOL * np.maximum(np.abs(d["exit"][hit]), 1.0)).all()
    # the cases cover what they claim
    if "inside" in path:
        assert hit.all() and (d["entry"] < 0).all()    # a camera inside the volume: every ray starts behind the eye
    else:
        assert (~hit).sum() > 10


@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_oracle_nondiff_march_matches_the_transliteration(oracle, path):
    d = np.load(path)
    ok = d["ok_march"]
    sr = float(d["sr"])
    rgba, steps = oracle.march_fwd(d["vol"], d["tf"], d["cam"], d["entry"], d["exit"], d["rays"], d["n"], 1 << 30, sr, 1)
    assert np.array_equal(steps[ok], d["live"][ok])
    assert np.abs(rgba - d["rgba"])[ok].max() <= 1e-9       # (libm pow against numpy's power on (1 - a)^(1/sr))
    f4 = np.float32
    rgba4, steps4 = oracle.march_fwd(d["vol"].astype(f4), d["tf"].astype(f4), d["cam"].astype(f4), d["entry"].astype(f4),
                                     d["exit"].astype(f4), d["rays"].astype(f4), d["n"], 1 << 30, sr, 1)
    same = (steps4 == d["live"]) & ok
    assert same[ok].mean() >= 0.98                          # f32 inputs may move a termination decision by one sample
    assert np.abs(rgba4 - d["rgba"]).max(-1)[same].max() <= F32_VS_F64_TOL
    if "ert" in path or "aniso" in path:
        assert (d["live"] < d["n"]).sum() > 20              # early termination is exercised
    if "orbit" in path or "far" in path:
        assert (d["rgba"][..., 3] > 0).any() and (d["tf"][:, 3] <= 1e-3).any()   # ... and so is the alpha <= 1e-3 skip


def test_transliteration_reproduces_its_fixture():
    """Guards the generating script against rot: re-running it for one case gives the stored vectors."""
    sys.path.insert(0, os.path.join(HERE, "golden"))
    import make_setup_nondiff_golden as G
    d = np.load(os.path.join(HERE, "golden", "setup_nondiff_a_orbit_sr1.npz"))
    c = G.make_case("a_orbit_sr1")
    for k in ("vol", "tf", "entry", "exit", "rays", "n", "rgba", "live"):
        assert np.array_equal(c[k], d[k]), k


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_ray_setup_matches_the_transliteration(hiplib, oracle, path):
    import torch
    from differender_amd import functional as F
    d = np.load(path)
    W, H = d["n"].shape
    hit = _hit(d)
    cam = torch.from_numpy(d["cam"].astype(np.float32)[None]).to("cuda:0")
    e, x, r, n = (t[0].cpu().numpy() for t in F.ray_setup(cam, (W, H), d["vol"].shape, float(d["sr"])))
    # bit-exact against the f32 oracle (the parity bar for this stage) ...
    e4, x4, r4, n4 = oracle.ray_setup(d["cam"].astype(np.float32), W, H, d["vol"].shape, sr=float(d["sr"]))
    assert np.array_equal(n, n4) and np.array_equal(r, r4)
    assert np.array_equal(e[hit], e4[hit]) and np.array_equal(x[hit], x4[hit])
    # ... and against the independent float64 transliteration to f32 rounding
    assert np.abs(n.astype(np.int64) - d["n"]).max() <= 1 and (n == d["n"]).mean() >= 0.97 and (n == d["n"])[~hit].all()
    assert np.abs(r - d["rays"]).max() <= RAY_TOL
    assert (np.abs(e - d["entry"])[hit] <= T_TOL * np.maximum(np.abs(d["entry"][hit]), 1.0)).all()
    assert (np.abs(x - d["exit"])[hit] <= T_TOL * np.maximum(np.abs(d["exit"][hit]), 1.0)).all()


@pytest.mark.gpu
@pytest.mark.parametrize("variant", [0, 1], ids=["flat", "baseline"])
@pytest.mark.parametrize("path", GOLDEN, ids=IDS)
def test_hip_nondiff_march_matches_the_transliteration(hiplib, path, variant):
    import torch
    from differender_amd import functional as F
    from differender_amd import _native as N
    d = np.load(path)
    dev = torch.device("cuda:0")
    T = lambda a, dt=np.float32: torch.from_numpy(np.ascontiguousarray(a.astype(dt))).to(dev)
    vol, tf, cam = T(d["vol"]), T(d["tf"]), T(d["cam"][None])
    e, x, r, n = T(d["entry"][None]), T(d["exit"][None]), T(d["rays"][None]), T(d["n"][None], np.int32)
    ws = F.alloc_workspace(1, d["n"].shape, d["vol"].shape, d["tf"].shape[0], dev) if variant == 0 else None
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 30, float(d["sr"]), N.DR_MODE_NONDIFF, variant=variant,
                             workspace=ws)
    ok = d["ok_march"]
    got, got_steps = out[0].cpu().numpy(), steps[0].cpu().numpy()
    same = (got_steps == d["live"]) & ok
    assert same[ok].mean() >= 0.98
    assert np.abs(got - d["rgba"]).max(-1)[same].max() <= F32_VS_F64_TOL      # the independent float64 vectors
    # parity proper (1e-5, step counts exact): the f32 oracle on the fixture's own inputs
    from oracle import oracle as O
    f4 = np.float32
    ref, ref_steps = O.march_fwd(d["vol"].astype(f4), d["tf"].astype(f4), d["cam"].astype(f4), d["entry"].astype(f4),
                                 d["exit"].astype(f4), d["rays"].astype(f4), d["n"], 1 << 30, float(d["sr"]), 1)
    reg = d["n"] != 1
    assert np.array_equal(got_steps[reg], ref_steps[reg])
    assert np.abs(got - ref).max(-1)[reg].max() <= FWD_TOL
