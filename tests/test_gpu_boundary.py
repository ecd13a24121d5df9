"""The drop-in boundary exercised the way the reference's own code drives it (all on the GPU, against the oracle):
the step-by-step VolumeRaycaster API of VR.py:431-438 / 468-475, autocast (custom_fwd/custom_bwd, VR.py:394,441),
the `.float()` of the setters (VR.py:118-125) and BASELINE config C1 exactly as examples/render_nondiff.py:19-27."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def _scene(O, N=32, R=32):
    vol = O.synth_volume(N)
    tf = O.bench_tf(R, 0.03)
    tf[:, 3] = np.linspace(0.01, 0.08, R)
    return vol, tf, O.in_circles(0.7)


def _user_layout(vol_f, tf_f):
    """field (W,D,H) / (R,4) arrays -> the user's (1,D,H,W) / (4,R) tensors (VR.py:566,571)."""
    return T(vol_f).permute(1, 2, 0).contiguous()[None], T(tf_f).t().contiguous()


def test_step_by_step_api_like_the_reference(oracle, hiplib):
    """RaycastFunction.forward's body (VR.py:431-438) and backward's (VR.py:468-475), call by call, on a VolumeRaycaster."""
    from differender_amd.volume_raycaster import VolumeRaycaster
    vol_h, tf_h, cam_h = _scene(oracle)
    WH, S, sr = (40, 32), 4096, 1.0
    vr = VolumeRaycaster(vol_h.shape, WH, max_samples=S, tf_resolution=tf_h.shape[0])
    assert vr.fov_rad == pytest.approx(math.radians(30.0)) and vr.resolution == WH and vr.max_samples == S
    volume, tf, look_from = T(vol_h), T(tf_h), T(cam_h)
    # ---- forward, VR.py:431-438
    vr.set_cam_pos(look_from)
    vr.set_volume(volume)
    vr.set_tf_tex(tf)
    vr.clear_framebuffer()
    assert int(vr.valid_sample_step_count.to_torch().min()) == 1  # VR.py:381
    vr.compute_entry_exit(sr, False)
    vr.raycast(sr)
    vr.get_final_image()
    out = vr.output_rgba.to_torch(device=volume.device)
    ref, steps_ref, (e, x, r, n) = oracle.render(vol_h, tf_h, cam_h, WH, S=S)
    assert out.shape == (*WH, 4) and np.abs(out.cpu().numpy() - ref).max() <= 1e-5
    assert np.array_equal(vr.sample_step_nums.to_torch().cpu().numpy(), n)
    # the reference's counter starts at 1 and gains one per live sample (VR.py:303,381)
    assert np.array_equal(vr.valid_sample_step_count.to_torch().cpu().numpy(), steps_ref + 1)
    assert vr.max_valid_sample_step_count == int(steps_ref.max())  # VR.py:370-372
    # ---- backward, VR.py:468-475
    g = np.random.default_rng(1).standard_normal((*WH, 4)).astype(np.float32)
    vr.clear_grad()
    vr.output_rgba.grad.from_torch(T(g))
    vr.get_final_image.grad()
    vr.raycast.grad(sr)
    dv = torch.nan_to_num(vr.volume.grad.to_torch(device=volume.device))
    dt = torch.nan_to_num(vr.tf_tex.grad.to_torch(device=volume.device))
    dv0, dt0 = oracle.march_bwd(vol_h, tf_h, cam_h, e, x, r, n, S, sr, g)
    assert dv.shape == vol_h.shape and dt.shape == tf_h.shape
    assert np.abs(dv.cpu().numpy() - dv0).max() <= 1e-4 * np.abs(dv0).max()
    assert np.abs(dt.cpu().numpy() - dt0).max() <= 1e-4 * np.abs(dt0).max()
    # ---- the nondiff pair, VR.py:504-511
    vr.clear_framebuffer()
    vr.compute_entry_exit(4.0, False)
    vr.raycast_nondiff(4.0)
    vr.get_final_image_nondiff()
    refn, _, _ = oracle.render(vol_h, tf_h, cam_h, WH, sr=4.0, mode=1)
    assert np.abs(vr.output_rgba.to_torch().cpu().numpy() - refn).max() <= 1e-5
    # get_final_image.grad() without an upstream gradient is an error, not a silent zero
    vr.clear_grad()
    with pytest.raises(RuntimeError):
        vr.get_final_image.grad()


def test_step_api_jitter_is_replayed_by_backward(oracle, hiplib):
    """compute_entry_exit(jitter=True) draws a seed; raycast.grad differentiates the image that was rendered with it
    (the reference's non-batched path has the same property because the fields persist, VR.py:468-471)."""
    from differender_amd.volume_raycaster import VolumeRaycaster
    vol_h, tf_h, cam_h = _scene(oracle)
    WH = (24, 24)
    vr = VolumeRaycaster(vol_h.shape, WH, max_samples=4096, tf_resolution=tf_h.shape[0])
    vr.set_cam_pos(T(cam_h)); vr.set_volume(T(vol_h)); vr.set_tf_tex(T(tf_h))
    vr.clear_framebuffer()
    torch.manual_seed(7)
    vr.compute_entry_exit(1.0, True)
    vr.raycast(1.0); vr.get_final_image()
    e, x, r, n = oracle.ray_setup(cam_h, *WH, vol_h.shape, jitter_seed=vr._jitter_seed)
    assert np.array_equal(vr.entry.to_torch().cpu().numpy()[n > 0], e[n > 0])
    ref, _ = oracle.march_fwd(vol_h, tf_h, cam_h, e, x, r, n, 4096, 1.0, 0)
    assert np.abs(vr.output_rgba.to_torch().cpu().numpy() - ref).max() <= 1e-5
    g = np.ones((*WH, 4), np.float32)
    vr.clear_grad(); vr.output_rgba.grad.from_torch(T(g)); vr.get_final_image.grad(); vr.raycast.grad(1.0)
    dv0, _ = oracle.march_bwd(vol_h, tf_h, cam_h, e, x, r, n, 4096, 1.0, g)
    assert np.abs(vr.volume.grad.to_torch().cpu().numpy() - dv0).max() <= 1e-4 * np.abs(dv0).max()


def test_config_c1_render_nondiff_script(oracle, hiplib):
    """BASELINE config C1 exactly as examples/render_nondiff.py:19-27: 64^3 volume, get_tf('tf1', 128), jitter=False,
    max_samples=1, in_circles(1.7*pi), raycast_nondiff(vol[None], tf[None], lf[None], sampling_rate=16.0), 128^2."""
    from differender.utils import get_tf, in_circles
    from differender.volume_raycaster import Raycaster
    n = 64
    vol_f = oracle.synth_volume(n)                      # field layout (W,D,H)
    vol, _ = _user_layout(vol_f, np.zeros((2, 4), np.float32))
    tf = get_tf("tf1", 128)
    raycaster = Raycaster(vol.shape[-3:], (128, 128), 128, jitter=False, max_samples=1)
    vol = vol.requires_grad_(True)
    tf = tf.to(dev()).requires_grad_(True)
    lf = in_circles(1.7 * math.pi).float().to(dev())
    im = raycaster.raycast_nondiff(vol[None], tf[None], lf[None], sampling_rate=16.0)
    assert im.shape == (1, 4, 128, 128) and not im.requires_grad
    tf_f = tf.detach().t().contiguous().cpu().numpy()
    ref, steps, (e, x, r, nn) = oracle.render(vol_f, tf_f, lf.cpu().numpy(), (128, 128), sr=16.0, mode=1)
    assert int(nn.max()) > 1500 and float((steps < nn)[nn > 0].mean()) > 0.3   # sr = 16: long rays, early termination
    got = np.flip(im[0].cpu().numpy().transpose(2, 1, 0), 1)                   # (4,H,W) -> (W,H,4), undo the flip of VR.py:513
    got_steps = raycaster.vr.valid_sample_step_count.to_torch()[0].cpu().numpy() - 1
    assert np.array_equal(got_steps, steps), int((got_steps != steps).sum())   # termination decisions are the oracle's (D3, D6)
    assert np.abs(got - ref).max() <= 1e-5
    assert float(im.max()) <= 1.0                                              # VR.py:358


def test_autocast_casts_inputs_to_float32(oracle, hiplib):
    """custom_fwd(cast_inputs=torch.float32) (VR.py:394): under autocast a half-precision volume / TF is rendered in
    float32 and the gradients come back in the leaves' dtype; custom_bwd (VR.py:441) runs backward with autocast off."""
    from differender_amd.volume_raycaster import Raycaster
    vol_h, tf_h, cam_h = _scene(oracle, N=24, R=16)
    WH = (32, 24)
    vol_u, tf_u = _user_layout(vol_h, tf_h)
    rc = Raycaster(vol_u.shape[-3:], WH, tf_h.shape[0], jitter=False, max_samples=4096)
    vol16 = vol_u.half().requires_grad_(True)
    tf16 = tf_u.half().requires_grad_(True)
    with torch.autocast("cuda", dtype=torch.float16):
        img = rc(vol16, tf16, T(cam_h))
        assert img.dtype == torch.float32
        loss = (img * img).sum()
    loss.backward()
    assert vol16.grad.dtype == torch.float16 and tf16.grad.dtype == torch.float16
    # oracle on the rounded inputs (what float32-casting the half tensors gives)
    vol_r = vol16.detach().float()[0].permute(2, 0, 1).contiguous().cpu().numpy()
    tf_r = tf16.detach().float().t().contiguous().cpu().numpy()
    ref, _, (e, x, r, n) = oracle.render(vol_r, tf_r, cam_h, WH, S=4096)
    ref_img = np.ascontiguousarray(np.flip(ref, 1).transpose(2, 1, 0))
    assert np.abs(img.detach().cpu().numpy() - ref_img).max() <= 1e-5
    dv0, dt0 = oracle.march_bwd(vol_r, tf_r, cam_h, e, x, r, n, 4096, 1.0, 2.0 * ref)
    got = vol16.grad[0].float().permute(2, 0, 1).cpu().numpy()
    assert np.abs(got - dv0).max() <= 2e-3 * np.abs(dv0).max()   # the gradient itself is rounded to half on the way out
    assert np.abs(tf16.grad.float().t().cpu().numpy() - dt0).max() <= 2e-3 * np.abs(dt0).max()


@pytest.mark.parametrize("vdt,tdt", [(torch.float64, torch.float64), (torch.bfloat16, torch.float32),
                                     (torch.uint8, torch.float64)], ids=["f64", "bf16-vol", "u8-vol"])
def test_setters_convert_like_dot_float(oracle, hiplib, vdt, tdt):
    """set_volume / set_tf_tex call .float() on their argument (VR.py:118-125): double, bfloat16 and integer volumes and
    double TFs render (and differentiate) instead of raising."""
    from differender_amd.volume_raycaster import Raycaster
    vol_h, tf_h, cam_h = _scene(oracle, N=24, R=16)
    WH = (24, 24)
    vol_u, tf_u = _user_layout(vol_h, tf_h)
    if vdt == torch.uint8:
        vol_in = (vol_u > 0.45).to(torch.uint8)            # a binary mask volume
    else:
        vol_in = vol_u.to(vdt)
    tf_in = tf_u.to(tdt)
    leaf = vdt.is_floating_point
    if leaf:
        vol_in = vol_in.requires_grad_(True)
    tf_in = tf_in.requires_grad_(True)
    rc = Raycaster(vol_u.shape[-3:], WH, tf_h.shape[0], jitter=False, max_samples=4096)
    img = rc(vol_in, tf_in, T(cam_h))
    vol_r = vol_in.detach().float()[0].permute(2, 0, 1).contiguous().cpu().numpy()
    tf_r = tf_in.detach().float().t().contiguous().cpu().numpy()
    ref, _, (e, x, r, n) = oracle.render(vol_r, tf_r, cam_h, WH, S=4096)
    assert np.abs(img.detach().cpu().numpy() - np.flip(ref, 1).transpose(2, 1, 0)).max() <= 1e-5
    img.sum().backward()
    dv0, dt0 = oracle.march_bwd(vol_r, tf_r, cam_h, e, x, r, n, 4096, 1.0, np.ones_like(ref))
    assert tf_in.grad.dtype == tdt
    assert np.abs(tf_in.grad.float().t().cpu().numpy() - dt0).max() <= 1e-4 * np.abs(dt0).max()
    if leaf:
        assert vol_in.grad.dtype == vdt
        tol = 1e-4 if vdt == torch.float64 else 1e-2
        assert np.abs(vol_in.grad[0].float().permute(2, 0, 1).cpu().numpy() - dv0).max() <= tol * np.abs(dv0).max()
    nd = rc.raycast_nondiff(vol_in.detach(), tf_in.detach(), T(cam_h), sampling_rate=2.0)
    refn, _, _ = oracle.render(vol_r, tf_r, cam_h, WH, sr=2.0, mode=1)
    assert np.abs(nd.cpu().numpy() - np.flip(refn, 1).transpose(2, 1, 0)).max() <= 1e-5


def test_backward_skips_nan_to_num_only_on_the_fast_path(hiplib):
    """dr_march_bwd_variant: the brick-centric backward sanitises its gradients itself; anything that falls back to the
    plain kernels still gets the reference's nan_to_num."""
    from differender_amd import functional as F
    vol = torch.zeros((16, 16, 16), device=dev()); tf = torch.zeros((8, 4), device=dev())
    ws = torch.empty(16, dtype=torch.uint8, device=dev())
    n = torch.zeros((1, 8, 8), dtype=torch.int32, device=dev())
    assert F.bwd_is_sanitised(vol, tf, torch.zeros_like(vol), ws, n)
    assert not F.bwd_is_sanitised(vol, tf, torch.zeros_like(vol), None, n)
    big = torch.zeros((20000, 4), device=dev())                       # TF too large for LDS -> plain kernels
    assert not F.bwd_is_sanitised(vol, big, torch.zeros_like(vol), ws, n)
    # more [layer][pixel] slots than 32-bit indices hold -> plain kernels (the same function dr_march_bwd_rows asks)
    from differender_amd import _native as N
    assert N.lib().dr_march_bwd_variant(1, 8, 8, 16, 16, 16, 8, 256, 16, 1, 256, 16, 1, 1, 0, 1) == N.DR_VARIANT_AUTO
    assert N.lib().dr_march_bwd_variant(1, 40000, 40000, 16, 16, 16, 8, 256, 16, 1, 256, 16, 1, 1, 0, 1) == N.DR_VARIANT_BASELINE


def test_module_backward_recognises_its_forward_whatever_the_argument_layout(oracle, hiplib):
    """The fast backward only runs under its own forward's fingerprint (sizes, strides, addresses of volume and ray
    buffers). Through the module every combination of batched / un-batched, converted and expanded arguments must still
    match -- a mismatch would be correct but 10-40 x slower, so it is counted (workspace header word 9) and warned about."""
    import warnings
    from differender_amd.volume_raycaster import Raycaster
    vol_h, tf_h, cam_h = _scene(oracle, N=24, R=16)
    WH = (24, 24)
    vol_u, tf_u = _user_layout(vol_h, tf_h)
    rc = Raycaster(vol_u.shape[-3:], WH, tf_h.shape[0], jitter=True, max_samples=4096)
    cam = T(cam_h)
    cases = [
        (vol_u, tf_u, cam),                                               # nothing batched
        (vol_u, tf_u[None].expand(3, -1, -1).contiguous(), cam),          # batched TF, ONE look_from (expanded inside)
        (vol_u, tf_u, torch.stack([cam, cam * 0.9, cam * 1.1])),          # batched cameras, shared volume and TF
        (vol_u.double(), tf_u.double(), cam.double()),                    # everything converted on the way in
        (vol_u[None].expand(2, -1, -1, -1, -1), tf_u, cam),               # expanded (stride-0) batch of volumes
    ]
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                    # the stale-workspace warning would raise
        for vol_in, tf_in, lf in cases:
            vol_in = vol_in.clone().requires_grad_(True)
            tf_in = tf_in.clone().requires_grad_(True)
            rc(vol_in, tf_in, lf).square().sum().backward()
            torch.cuda.synchronize()
            rc(vol_in.detach(), tf_in.detach(), lf)                       # a later call looks at the backward's snapshot
            torch.cuda.synchronize()
            st = rc.vr.last_stats
            assert st is not None and int(st[9]) == 0 and int(st[3]) != 0, st[:12]


def test_hidden_tf_writes_heal_themselves(oracle, hiplib):
    """a hand-written update loop that writes the TF through `.data` (torch's version counter does not move) while
    its alphas grow. The cached "no early termination" hint goes stale; the device repairs every such render (correct images,
    header word 8), the module hears of it from its asynchronous header snapshot, warns once and withholds the hint from then
    on -- the slow repair does not persist."""
    import warnings
    from differender_amd import functional as Fn
    from differender_amd.volume_raycaster import Raycaster
    vol_f = oracle.synth_volume(32)
    tf_f = oracle.bench_tf(32, 0.004)
    cam = oracle.in_circles(0.6)
    vol_u, tf_u = _user_layout(vol_f, tf_f)
    Fn._hints.__init__()                                    # a clean cache for this test
    rc = Raycaster((32, 32, 32), (40, 40), 32, jitter=False, max_samples=4096)
    for k in range(4):                                      # the hint is learnt: no pre-pass launches from now on
        rc(vol_u, tf_u, T(cam)); torch.cuda.synchronize()
    assert rc._hints(tf_u, vol_u.squeeze(0).permute(2, 0, 1), 1.0, 0) != 0
    v0 = tf_u._version
    tf_u.data[3, :] = 0.6                                   # now most rays terminate early -- behind the version counter's back
    assert tf_u._version == v0
    tf_now = tf_u.t().contiguous().cpu().numpy()
    ref, steps_ref, _ = oracle.render(vol_f, tf_now, cam, (40, 40), S=4096)
    assert (steps_ref < oracle.ray_setup(cam, 40, 40, vol_f.shape)[3]).mean() > 0.3
    ref_img = np.ascontiguousarray(np.flip(ref, 1).transpose(2, 1, 0))
    repaired = []
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        for k in range(12):
            img = rc(vol_u, tf_u, T(cam)); torch.cuda.synchronize()
            assert np.abs(img.cpu().numpy() - ref_img).max() <= 1e-5          # right whatever the hint said
            repaired.append(int(rc.vr.last_stats[8]) if rc.vr.last_stats is not None else -1)
    assert any("DR_HINT_NO_EARLY_TERMINATION" in str(w.message) for w in caught), [str(w.message) for w in caught]
    assert max(repaired) >= 1 and repaired[-1] == 0, repaired                  # repaired at first, healed at the end
    Fn._hints.__init__()


def test_wrong_hint_reports_are_scoped_and_expire(oracle, hiplib):
    """a wrong-hint report blames the TFs that were recently GIVEN the hint -- not every cached TF, and nobody when the
    hint was the caller's own -- and the blame expires after a few clean re-readings of the TF's largest alpha."""
    import warnings
    from differender_amd import functional as Fn
    H = Fn._TerminationHints()
    shape = (32, 32, 32)
    tf_a = T(oracle.bench_tf(32, 0.004)); tf_b = T(oracle.bench_tf(32, 0.003)); tf_c = T(oracle.bench_tf(32, 0.002))

    def learn(tf):
        for _ in range(3):
            h = H.hints(tf, shape, 1.0, 4096, 0); torch.cuda.synchronize()
        return h
    assert learn(tf_a) == learn(tf_b) == learn(tf_c) == Fn.N.DR_HINT_NO_EARLY_TERMINATION
    # a report with nothing handed out recently (an explicit caller hint went wrong): nobody is blamed
    H._handed = []
    assert H.report_wrong_hint() == 0
    assert H.hints(tf_a, shape, 1.0, 4096, 0) == Fn.N.DR_HINT_NO_EARLY_TERMINATION
    # only the TF that was just given the hint is blamed
    H._handed = []
    assert H.hints(tf_b, shape, 1.0, 4096, 0) == Fn.N.DR_HINT_NO_EARLY_TERMINATION
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        assert H.report_wrong_hint() == 1
    assert any("DR_HINT_NO_EARLY_TERMINATION" in str(w.message) for w in caught)
    assert H.hints(tf_b, shape, 1.0, 4096, 0) == 0
    assert H.hints(tf_a, shape, 1.0, 4096, 0) == Fn.N.DR_HINT_NO_EARLY_TERMINATION
    assert H.hints(tf_c, shape, 1.0, 4096, 0) == Fn.N.DR_HINT_NO_EARLY_TERMINATION
    # ... and regains it after DISTRUST_REFRESHES clean re-readings (one every REFRESH_EVERY calls), reusing ONE pinned scalar
    host0 = H._seen[H._key(tf_b)]["host"]
    calls = 0
    while H.hints(tf_b, shape, 1.0, 4096, 0) == 0:
        torch.cuda.synchronize(); calls += 1
        assert calls <= (H.DISTRUST_REFRESHES + 2) * H.REFRESH_EVERY
    assert calls >= (H.DISTRUST_REFRESHES - 1) * H.REFRESH_EVERY
    assert H._seen[H._key(tf_b)]["host"] is host0 and host0 is not None


def test_every_wrong_hint_forward_is_reported(oracle, hiplib, monkeypatch):
    """the wrong-hint report was de-duplicated on the workspace fingerprint, which a training loop repeats on every
    iteration (same volume, recycled ray buffers) -- every report after the first was dropped and the distrust never escalated.
    Keyed on the forwards issued by the raycaster now: N forwards under a wrong explicit hint make N reports (one each, although
    the forward's and the backward's snapshots both show the flag), a clean forward in between makes none."""
    from differender_amd import functional as Fn
    from differender_amd.volume_raycaster import Raycaster
    vol_h = oracle.synth_volume(32)
    rc = Raycaster(vol_h.shape, (16, 16), 32, jitter=False, max_samples=4096)
    vol = T(vol_h)[None].requires_grad_(True)
    tf_opaque = T(oracle.bench_tf(32, 0.9))           # (R, 4): RaycastFunction's layout
    lf = T(oracle.in_circles(0.3))
    reports = []
    monkeypatch.setattr(Fn._hints, "report_wrong_hint", lambda: reports.append(1) or 0)
    from differender_amd.volume_raycaster import RaycastFunction
    for it in range(4):
        hint = Fn.N.DR_HINT_NO_EARLY_TERMINATION if it != 2 else 0     # iteration 2 renders without the (wrong) hint
        out = RaycastFunction.apply(rc.vr, vol[0].permute(2, 0, 1), tf_opaque, lf, 1.0, (False, 0), False, hint)
        out.sum().backward()
        torch.cuda.synchronize()
    rc.vr._watch_workspace(torch.empty(256, dtype=torch.uint8, device=vol.device), 1)   # collect the last snapshot
    assert len(reports) == 3, reports


def test_trainable_tf_is_read_on_first_sight(oracle, hiplib):
    """A TF that requires grad is a parameter somebody optimises: its largest alpha is read the first time it is seen (a plain
    tensor: the second time), so a training loop that rewrites it every iteration gets the harmless "many rays terminate" hint
    from its second or third iteration on instead of its ninth."""
    from differender_amd import functional as Fn
    H = Fn._TerminationHints()
    tf_h = oracle.bench_tf(32, 0.5)                      # opaque: rays terminate
    shape = (32, 32, 32)
    tf_p = T(tf_h).requires_grad_(True)
    tf_c = T(tf_h * 1.0)
    got_p, got_c = [], []
    for k in range(3):
        got_p.append(H.hints(tf_p, shape, 1.0, 4096, 0)); got_c.append(H.hints(tf_c, shape, 1.0, 4096, 0))
        torch.cuda.synchronize()
        with torch.no_grad():
            tf_p.mul_(0.999); tf_c.mul_(0.999)          # an optimiser step: the version moves every iteration
    assert got_p[0] == 0 and Fn.N.DR_HINT_EARLY_TERMINATION in got_p[1:], got_p
    assert got_c == [0, 0, 0], got_c                     # a plain tensor that changes every call is not read before its eighth call
    # the RESULT of a differentiable op also "requires grad" but is a new tensor every iteration: no read, no cache entry kept hot
    logits = T(tf_h).requires_grad_(True)
    for k in range(3):
        t = torch.sigmoid(logits)
        assert t.requires_grad and not t.is_leaf
        assert H.hints(t, shape, 1.0, 4096, 0) == 0
        assert H._seen[H._key(t)]["pending"] is None
        torch.cuda.synchronize(); del t
