"""Parity of the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Tolerances (BASELINE.json north_star): forward RGBA within 1e-5, gradients within 1e-4 -- the latter
relative to the largest gradient magnitude of the tensor (f32 scatter-add of ~1e2..1e3 contributions per
element). Ray-setup buffers are compared bit-for-bit and then shared between oracle and device march
so that a flipped floor() cannot mask a march bug."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

FWD_TOL = 1e-5
GRAD_TOL = 1e-4


def dev():
    return torch.device("cuda:0")


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a)).to(dev())


def scene(O, N=48, R=64, tf="bench", alpha=0.01, cam_i=0.3):
    vol = O.synth_volume(N)
    tfa = O.bench_tf(R, alpha) if tf == "bench" else O.peaks_tf(R)
    return vol, tfa, O.in_circles(cam_i)


def gpu_setup(F, cam, WH, vshape, sr=1.0, seed=0, view_base=0):
    cam_t = T(np.atleast_2d(cam).astype(np.float32))
    return F.ray_setup(cam_t, WH, vshape, sr, 30.0, 0.1, seed, view_base)


def grad_close(a, b, tol=GRAD_TOL):
    scale = max(float(np.abs(b).max()), 1e-12)
    err = float(np.abs(a - b).max()) / scale
    return err <= tol, err


class _Variant:
    """differender_amd.functional with the kernel variant pinned (AUTO = brick-centric fast path with a
    workspace shared between forward and backward; BASELINE = plain kernels, no workspace)."""

    def __init__(self, functional, variant):
        self._f, self.variant = functional, variant
        self.ray_setup = functional.ray_setup
        self._ws = None

    def march_fwd(self, vol, tf, cam, e, x, r, n, S, sr, mode=0, **kw):
        if self.variant != 1:
            self._ws = self._f.alloc_workspace(n.shape[0], n.shape[1:], vol.shape[-3:], tf.shape[-2], vol.device)
            assert self._ws is not None
        return self._f.march_fwd(vol, tf, cam, e, x, r, n, S, sr, mode, variant=self.variant, workspace=self._ws, **kw)

    def march_bwd(self, vol, tf, cam, e, x, r, n, S, sr, g, out, want_vol=True, want_tf=True):
        return self._f.march_bwd(vol, tf, cam, e, x, r, n, S, sr, g, out, want_vol, want_tf, variant=self.variant,
                                 workspace=self._ws)


@pytest.fixture(scope="module", params=[0, 1], ids=["flat", "baseline"])
def F(hiplib, request):
    assert torch.cuda.is_available(), "gpu tests need a ROCm device"
    from differender_amd import functional
    return _Variant(functional, request.param)


@pytest.mark.parametrize("WH,vshape,sr,cam_i", [((64, 64), (48, 48, 48), 1.0, 0.3), ((40, 72), (32, 48, 40), 2.0, 1.9),
                                               ((33, 21), (16, 16, 16), 0.7, 4.0)])
def test_ray_setup_bit_exact(oracle, F, WH, vshape, sr, cam_i):
    cam = oracle.in_circles(cam_i)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vshape, sr=sr)
    e, x, r, n = gpu_setup(F, cam, WH, vshape, sr)
    assert np.array_equal(n[0].cpu().numpy(), n0)
    hit = n0 > 0
    assert np.array_equal(e[0].cpu().numpy()[hit], e0[hit])
    assert np.array_equal(x[0].cpu().numpy()[hit], x0[hit])
    assert np.array_equal(r[0].cpu().numpy(), r0)


def test_ray_setup_jitter_bit_exact(oracle, F):
    cam = oracle.in_circles(2.2)
    e0, x0, r0, n0 = oracle.ray_setup(cam, 48, 48, (40, 40, 40), jitter_seed=12345, view=3)
    e, x, r, n = gpu_setup(F, cam, (48, 48), (40, 40, 40), seed=12345, view_base=3)
    hit = n0 > 0
    assert np.array_equal(n[0].cpu().numpy(), n0)
    assert np.array_equal(e[0].cpu().numpy()[hit], e0[hit])


def _fwd_both(O, F, vol, tf, cam, WH, S=1 << 20, sr=1.0, mode=0, seed=0):
    e0, x0, r0, n0 = O.ray_setup(cam, *WH, vol.shape, sr=sr, jitter_seed=seed)
    ref, steps_ref = O.march_fwd(vol, tf, cam, e0, x0, r0, n0, S, sr, mode)
    e, x, r, n = gpu_setup(F, cam, WH, vol.shape, sr, seed)
    out, steps = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, S, sr, mode)
    return ref, steps_ref, out[0].cpu().numpy(), steps[0].cpu().numpy(), (e0, x0, r0, n0), (e, x, r, n)


@pytest.mark.parametrize("sr", [1.0, 2.0])
@pytest.mark.parametrize("mode", [0, 1])
def test_forward_parity(oracle, F, sr, mode):
    vol, tf, cam = scene(oracle, alpha=0.02)
    ref, sref, out, steps, _, _ = _fwd_both(oracle, F, vol, tf, cam, (64, 64), sr=sr, mode=mode)
    assert np.array_equal(steps, sref)
    assert np.abs(out - ref).max() <= FWD_TOL


def test_forward_parity_early_termination(oracle, F):
    vol, tf, cam = scene(oracle, tf="peaks", R=128, cam_i=1.1)
    tf[:, 3] = np.linspace(0.0, 0.4, 128)
    ref, sref, out, steps, (e0, x0, r0, n0), _ = _fwd_both(oracle, F, vol, tf, cam, (64, 64))
    assert (sref < n0)[n0 > 40].mean() > 0.3, "scene must exercise early termination"
    # termination decisions are the oracle's (a decision within rounding of 0.99 is re-taken sequentially)
    assert np.array_equal(steps, sref), int((steps != sref).sum())
    assert np.abs(out - ref).max() <= FWD_TOL


def test_forward_max_samples_and_rect(oracle, F):
    vol = oracle.synth_volume((24, 40, 32))
    tf = oracle.bench_tf(32, 0.01)
    cam = oracle.in_circles(5.0)
    ref, sref, out, steps, _, _ = _fwd_both(oracle, F, vol, tf, cam, (40, 24), S=17)
    assert steps.max() == 17 and np.array_equal(steps, sref)
    assert np.abs(out - ref).max() <= FWD_TOL


def test_forward_jittered(oracle, F):
    vol, tf, cam = scene(oracle)
    ref, sref, out, steps, _, _ = _fwd_both(oracle, F, vol, tf, cam, (48, 48), seed=99)
    assert np.abs(out - ref).max() <= FWD_TOL
    ref0, _, _, _, _, _ = _fwd_both(oracle, F, vol, tf, cam, (48, 48), seed=0)
    assert np.abs(ref - ref0).max() > 1e-6  # jitter changes the image


def test_forward_flat_volume_ambient_only(oracle, F):
    R, k = 11, 4
    vol = np.full((8, 8, 8), k / (R - 1), np.float32)
    tf = np.zeros((R, 4), np.float32); tf[k] = [0.9, 0.5, 0.25, 0.3]
    cam = np.array([0.0, 0.3, 3.0], np.float32)
    ref, sref, out, steps, _, _ = _fwd_both(oracle, F, vol, tf, cam, (16, 16))
    assert np.isfinite(out).all() and np.abs(out - ref).max() <= FWD_TOL


def test_forward_strided_and_f16_volume(oracle, F):
    vol, tf, cam = scene(oracle, N=32)
    e, x, r, n = gpu_setup(F, cam, (32, 32), vol.shape)
    base, _ = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 4096, 1.0)
    # user layout (1,D,H,W): field index (i,j,k) = (W,D,H) -> memory order (D,H,W) (VR.py:566,571)
    user = T(vol).permute(1, 2, 0).contiguous()  # (D,H,W)
    view = user.permute(2, 0, 1)  # (W,D,H) view, x fastest in memory
    assert not view.is_contiguous()
    out, _ = F.march_fwd(view, T(tf), T(cam[None]), e, x, r, n, 4096, 1.0)
    assert torch.equal(out, base)
    # fp16 storage: compare with the oracle run on the rounded volume
    vol16 = vol.astype(np.float16)
    e0, x0, r0, n0 = oracle.ray_setup(cam, 32, 32, vol.shape)
    ref, _ = oracle.march_fwd(vol16.astype(np.float32), tf, cam, e0, x0, r0, n0, 4096, 1.0, 0)
    out16, _ = F.march_fwd(T(vol16), T(tf), T(cam[None]), e, x, r, n, 4096, 1.0)
    assert np.abs(out16[0].cpu().numpy() - ref).max() <= FWD_TOL


def _bwd_both(O, F, vol, tf, cam, WH, sr=1.0, S=1 << 20, seed=5, want=(True, True)):
    e0, x0, r0, n0 = O.ray_setup(cam, *WH, vol.shape, sr=sr)
    rng = np.random.RandomState(seed)
    g = rng.randn(*WH, 4).astype(np.float32)
    dv0, dt0 = O.march_bwd(vol, tf, cam, e0, x0, r0, n0, S, sr, g, *want)
    e, x, r, n = gpu_setup(F, cam, WH, vol.shape, sr)
    vt, tt, ct = T(vol), T(tf), T(cam[None])
    out, _ = F.march_fwd(vt, tt, ct, e, x, r, n, S, sr)
    dv, dt = F.march_bwd(vt, tt, ct, e, x, r, n, S, sr, T(g[None]), out, *want)
    return dv0, dt0, (None if dv is None else dv.cpu().numpy()), (None if dt is None else dt.cpu().numpy())


@pytest.mark.parametrize("sr", [1.0, 2.0])
def test_backward_parity(oracle, F, sr):
    vol, tf, cam = scene(oracle, N=40, R=32, alpha=0.03)
    tf[:, 3] = np.linspace(0.01, 0.08, 32)  # non-constant alpha so dV sees the TF slope
    dv0, dt0, dv, dt = _bwd_both(oracle, F, vol, tf, cam, (48, 48), sr=sr)
    ok, err = grad_close(dt, dt0); assert ok, f"d_tf rel err {err}"
    ok, err = grad_close(dv, dv0); assert ok, f"d_vol rel err {err}"


def test_backward_parity_early_termination(oracle, F):
    vol, tf, cam = scene(oracle, N=40, tf="peaks", R=64, cam_i=1.1)
    tf[:, 3] = np.linspace(0.0, 0.4, 64)
    dv0, dt0, dv, dt = _bwd_both(oracle, F, vol, tf, cam, (48, 48))
    ok, err = grad_close(dt, dt0); assert ok, f"d_tf rel err {err}"
    ok, err = grad_close(dv, dv0); assert ok, f"d_vol rel err {err}"


def test_backward_selective_outputs(oracle, F):
    vol, tf, cam = scene(oracle, N=24, R=16, alpha=0.03)
    dv0, dt0, dv, dt = _bwd_both(oracle, F, vol, tf, cam, (24, 24), want=(False, True))
    assert dv is None and grad_close(dt, dt0)[0]
    dv0, dt0, dv, dt = _bwd_both(oracle, F, vol, tf, cam, (24, 24), want=(True, False))
    assert dt is None and grad_close(dv, dv0)[0]


def test_backward_flat_volume_has_no_nan(oracle, F):
    vol = np.full((12, 12, 12), 0.4, np.float32)
    tf = oracle.bench_tf(16, 0.05)
    cam = oracle.in_circles(0.2)
    dv0, dt0, dv, dt = _bwd_both(oracle, F, vol, tf, cam, (16, 16))
    assert np.isfinite(dv).all() and np.isfinite(dt).all()
    assert grad_close(dv, dv0)[0] and grad_close(dt, dt0)[0]


def test_batched_views_shared_volume(oracle, F):
    """n_views > 1 in one launch; shared volume/tf accumulate one gradient (replaces VR.py:418-426,450-464)."""
    vol, tf, _ = scene(oracle, N=32, R=32, alpha=0.03)
    cams = np.stack([oracle.in_circles(0.5 * v) for v in range(3)])
    WH = (32, 32)
    rng = np.random.RandomState(1)
    g = rng.randn(3, *WH, 4).astype(np.float32)
    dv_ref = np.zeros_like(vol); dt_ref = np.zeros_like(tf); refs = []
    for v in range(3):
        e0, x0, r0, n0 = oracle.ray_setup(cams[v], *WH, vol.shape)
        ref, _ = oracle.march_fwd(vol, tf, cams[v], e0, x0, r0, n0, 4096, 1.0, 0)
        a, b = oracle.march_bwd(vol, tf, cams[v], e0, x0, r0, n0, 4096, 1.0, g[v])
        dv_ref += a; dt_ref += b; refs.append(ref)
    ct = T(cams)
    e, x, r, n = F.ray_setup(ct, WH, vol.shape, 1.0)
    out, _ = F.march_fwd(T(vol), T(tf), ct, e, x, r, n, 4096, 1.0)
    assert np.abs(out.cpu().numpy() - np.stack(refs)).max() <= FWD_TOL
    dv, dt = F.march_bwd(T(vol), T(tf), ct, e, x, r, n, 4096, 1.0, T(g), out)
    assert dv.shape == vol.shape and dt.shape == tf.shape
    assert grad_close(dv.cpu().numpy(), dv_ref)[0] and grad_close(dt.cpu().numpy(), dt_ref)[0]
    # per-view volumes/tfs: per-view gradients
    volb = T(np.stack([vol, vol * 0.9, vol * 0.8])); tfb = T(np.stack([tf, tf, tf]))
    outb, _ = F.march_fwd(volb, tfb, ct, e, x, r, n, 4096, 1.0)
    dvb, dtb = F.march_bwd(volb, tfb, ct, e, x, r, n, 4096, 1.0, T(g), outb)
    assert dvb.shape == (3, *vol.shape) and dtb.shape == (3, *tf.shape)
    e0, x0, r0, n0 = oracle.ray_setup(cams[1], *WH, vol.shape)
    a, b = oracle.march_bwd(vol * np.float32(0.9), tf, cams[1], e0, x0, r0, n0, 4096, 1.0, g[1])
    assert grad_close(dvb[1].cpu().numpy(), a)[0] and grad_close(dtb[1].cpu().numpy(), b)[0]


def test_raycaster_module_matches_oracle_layout(oracle, F):
    """End-to-end through Raycaster/RaycastFunction: (1,D,H,W) in, (4,H,W) out with the H flip (VR.py:543-548),
    autograd gradients in the user's layout."""
    from differender_amd.volume_raycaster import Raycaster
    D, H, Wv = 20, 24, 28
    vol_f = oracle.synth_volume((Wv, D, H))  # field layout (W,D,H)
    tf_f = oracle.bench_tf(16, 0.03); tf_f[:, 3] = np.linspace(0.01, 0.06, 16)
    cam = oracle.in_circles(0.9)
    out_shape = (32, 40)  # (w, h)
    rc = Raycaster((D, H, Wv), out_shape, 16, jitter=False, max_samples=4096)
    vol_u = T(vol_f).permute(1, 2, 0).contiguous()[None].requires_grad_(True)  # (1,D,H,W)
    tf_u = T(tf_f).t().contiguous().requires_grad_(True)  # (4,R)
    img = rc(vol_u, tf_u, T(cam))
    assert img.shape == (4, out_shape[1], out_shape[0]) and img.is_contiguous()
    ref, _, (e0, x0, r0, n0) = oracle.render(vol_f, tf_f, cam, out_shape, S=4096)
    ref_img = np.ascontiguousarray(np.flip(ref, 1).transpose(2, 1, 0))
    assert np.abs(img.detach().cpu().numpy() - ref_img).max() <= FWD_TOL
    rng = np.random.RandomState(2)
    g_img = rng.randn(*img.shape).astype(np.float32)
    (img * T(g_img)).sum().backward()
    g_field = np.ascontiguousarray(np.flip(g_img.transpose(2, 1, 0), 1))  # back to (W,H,4)
    dv0, dt0 = oracle.march_bwd(vol_f, tf_f, cam, e0, x0, r0, n0, 4096, 1.0, g_field)
    assert vol_u.grad.shape == vol_u.shape and tf_u.grad.shape == tf_u.shape
    assert grad_close(vol_u.grad[0].permute(2, 0, 1).cpu().numpy(), dv0)[0]
    assert grad_close(tf_u.grad.t().cpu().numpy(), dt0)[0]
    # nondiff render: default rate 4x, never jittered, clamped
    nd = rc.raycast_nondiff(vol_u.detach(), tf_u.detach(), T(cam))
    refn, _, _ = oracle.render(vol_f, tf_f, cam, out_shape, sr=4.0, mode=1)
    assert np.abs(nd.cpu().numpy() - np.flip(refn, 1).transpose(2, 1, 0)).max() <= FWD_TOL
    # batched look_from with shared volume/tf: (BS,4,H,W), one accumulated gradient
    vol_u.grad = None; tf_u.grad = None
    cams = np.stack([oracle.in_circles(0.9), oracle.in_circles(2.0)])
    imgb = rc(vol_u, tf_u, T(cams))
    assert imgb.shape == (2, 4, out_shape[1], out_shape[0])
    assert np.abs(imgb[0].detach().cpu().numpy() - ref_img).max() <= FWD_TOL
    imgb.sum().backward()
    assert vol_u.grad.shape == vol_u.shape and torch.isfinite(vol_u.grad).all()


def test_optimisation_loop_reduces_loss(oracle, F):
    """Counterpart of examples/test_opt_tf.py:63-88 on synthetic data: the loss must go down."""
    from differender_amd.volume_raycaster import Raycaster
    N = 32
    vol_gt = T(oracle.synth_volume(N)).permute(1, 2, 0).contiguous()[None]
    tf = T(oracle.peaks_tf(32)).t().contiguous()
    rc = Raycaster((N, N, N), (48, 48), 32, jitter=True, max_samples=1024)
    torch.manual_seed(0)
    vol = (vol_gt + 0.15 * torch.randn_like(vol_gt)).clamp(0, 1).requires_grad_(True)
    opt = torch.optim.Adam([vol], lr=2e-2)
    cams = T(np.stack([oracle.in_circles(0.7 * v) for v in range(4)]))
    with torch.no_grad():
        gt = rc.raycast_nondiff(vol_gt, tf, cams, sampling_rate=2.0)
    losses = []
    for it in range(12):
        opt.zero_grad()
        res = rc(vol, tf, cams)
        loss = torch.nn.functional.mse_loss(res, gt)
        loss.backward()
        opt.step()
        with torch.no_grad():
            vol.clamp_(0.0, 1.0)
        losses.append(float(loss.detach()))
    assert losses[-1] < 0.7 * losses[0], losses


def test_optimisation_loop_with_the_reference_loss(oracle, F):
    """The same loop with the loss of examples/test_opt_tf.py:70-72 -- nan_to_num(1 - ssim(res, gt)) + mse -- through the torch
    restatement of pytorch_msssim.ssim (differender_amd/utils/losses.py): both terms must go down."""
    from differender_amd.volume_raycaster import Raycaster
    from differender_amd.utils import dssim_mse_loss
    N = 32
    vol_gt = T(oracle.synth_volume(N)).permute(1, 2, 0).contiguous()[None]
    tf = T(oracle.peaks_tf(32)).t().contiguous()
    rc = Raycaster((N, N, N), (48, 48), 32, jitter=True, max_samples=1024)
    torch.manual_seed(1)
    vol = (vol_gt + 0.15 * torch.randn_like(vol_gt)).clamp(0, 1).requires_grad_(True)
    opt = torch.optim.Adam([vol], lr=2e-2)
    cams = T(np.stack([oracle.in_circles(0.7 * v) for v in range(4)]))
    with torch.no_grad():
        gt = rc.raycast_nondiff(vol_gt, tf, cams, sampling_rate=2.0)
    hist = []
    for it in range(12):
        opt.zero_grad()
        loss, dssim, mse = dssim_mse_loss(rc(vol, tf, cams), gt)
        loss.backward()
        opt.step()
        with torch.no_grad():
            vol.clamp_(0.0, 1.0)
        hist.append((float(loss.detach()), float(dssim.detach()), float(mse.detach())))
    assert hist[-1][0] < 0.8 * hist[0][0] and hist[-1][1] < hist[0][1] and hist[-1][2] < hist[0][2], hist


def test_loss_epilogue_and_momentum_step(oracle, hiplib):
    """dr_mse_loss_grad / dr_tf_momentum_step vs the oracle's statement of EX.py:368-381 (bit-exact
    elementwise results; the loss sum is carried in double on both sides)."""
    from differender_amd import functional as Fn
    rng = np.random.default_rng(11)
    out = rng.random((2, 40, 24, 4), dtype=np.float32)
    ref = rng.random((2, 40, 24, 4), dtype=np.float32)
    for inv in (None, 1.0 / (3 * 40 * 24)):
        loss_o, grad_o = oracle.mse_loss_grad(out, ref, inv)
        loss_g, grad_g = Fn.mse_loss_grad(T(out), T(ref), inv)
        assert np.array_equal(grad_g.cpu().numpy(), grad_o)
        assert abs(float(loss_g) - loss_o) <= 1e-12 * abs(loss_o) + 1e-15
    acc = torch.zeros((), dtype=torch.float64, device=dev())
    Fn.mse_loss_grad(T(out[0]), T(ref[0]), 1.0 / out.size, want_grad=False, loss=acc)
    Fn.mse_loss_grad(T(out[1]), T(ref[1]), 1.0 / out.size, want_grad=False, loss=acc)   # accumulates over views
    assert abs(float(acc) - oracle.mse_loss_grad(out, ref)[0]) < 1e-10

    tf = (rng.random((64, 4), dtype=np.float32) * 0.1)
    g = (rng.standard_normal((64, 4)) * 3).astype(np.float32)
    mom = (rng.standard_normal((64, 4)) * 0.01).astype(np.float32)
    tf_o, mom_o = oracle.tf_momentum_step(tf, g, mom, 0.05, 0.9, 1.0)
    tf_g, mom_g = Fn.tf_momentum_step(T(tf), T(g), T(mom), 0.05, 0.9, 1.0)
    assert np.array_equal(tf_g.cpu().numpy(), tf_o) and np.array_equal(mom_g.cpu().numpy(), mom_o)
    with pytest.raises(ValueError):
        Fn.tf_momentum_step(T(tf), T(g[:32]), T(mom), 0.05, 0.9, 1.0)


def test_tf_optimisation_with_fused_epilogue(oracle, F):
    """The loop of examples/taichi_volume_raycaster.py:583-601 (render, loss, backward, momentum step on the TF)
    through the C ABI only: no framework elementwise ops between the kernels. The loss must go down."""
    from differender_amd import functional as Fn
    N, R, WH = 32, 32, (48, 48)
    vol = T(oracle.synth_volume(N))
    tf_gt = T(oracle.peaks_tf(R))
    tf = T(oracle.bench_tf(R, 0.05)).clone()
    mom = torch.zeros_like(tf)
    cam = T(np.atleast_2d(oracle.in_circles(0.4)))
    e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0, 30.0, 0.1, 0, 0)
    ref, _ = F.march_fwd(vol, tf_gt, cam, e, x, r, n, 4096, 1.0)
    ref = ref.clone()
    losses = []
    for it in range(25):
        out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0)
        loss, g = Fn.mse_loss_grad(out, ref)
        _, d_tf = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, want_vol=False)
        Fn.tf_momentum_step(tf, d_tf, mom, 0.5, 0.8, 1.0)
        losses.append(float(loss))
    assert (tf >= 0).all()
    assert losses[-1] < 0.5 * losses[0], losses


@pytest.mark.parametrize("cam", [(0.2, 0.1, 0.3), (0.9, 0.3, -1.15), (0.0, 1.6, 0.05), (-0.95, 0.9, 0.97), (1.0, 0.0, 0.0)],
                         ids=["inside", "near-corner", "above-nearly-along-y", "inside-corner", "on-a-face"])
def test_unusual_cameras(oracle, F, cam):
    """Camera inside the box (every ray starts BEHIND the eye: the reference does not clip tmin at 0, VR.py:28-53),
    very close to it (bricks project to large pixel rectangles: several listing rounds per brick), nearly along the up
    vector. All of them run through the brick pipeline: a ray's layers count from the brick of its own first sample,
    and only single-sample rays are left to the per-ray fallback."""
    vol, tf, _ = scene(oracle, N=40, R=32, alpha=0.04)
    tf[:, 3] = np.linspace(0.01, 0.1, 32)
    cam = np.array(cam, np.float32)
    WH = (48, 40)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, 0)
    e, x, r, n = gpu_setup(F, cam, WH, vol.shape)
    assert np.array_equal(n[0].cpu().numpy(), n0)
    out, steps = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 1.0)
    if F.variant == 0:
        from differender_amd.functional import workspace_stats
        st = workspace_stats(F._ws)
        assert int(st[0]) == 0, "rays failed the sample-count check"
        assert int(st[2]) == int((n0 == 1).sum()), "only single-sample rays may take the per-ray fallback"
    o = out[0].cpu().numpy()
    assert np.array_equal(steps[0].cpu().numpy(), sref), int((steps[0].cpu().numpy() != sref).sum())
    assert np.abs(o - ref).max() <= FWD_TOL
    g = np.random.RandomState(4).randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, g)
    dv, dt = F.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 1.0, T(g[None]), out)
    ok, err = grad_close(dv.cpu().numpy(), dv0); assert ok, err
    ok, err = grad_close(dt.cpu().numpy(), dt0); assert ok, err


@pytest.mark.parametrize("mode,sr", [(0, 1.0), (1, 4.0)], ids=["diff", "nondiff_sr4"])
def test_camera_inside_with_early_termination(oracle, F, mode, sr):
    """Inside the volume with an opaque TF: rays start behind the eye, the alpha pre-pass runs as one phase, and the
    termination decisions are still the oracle's."""
    vol, tf, _ = scene(oracle, N=48, tf="peaks", R=64)
    tf[:, 3] = np.linspace(0.0, 0.5, 64)
    cam = np.array([0.3, -0.2, 0.25], np.float32)
    WH = (56, 48)
    ref, sref, out, steps, (e0, x0, r0, n0), _ = _fwd_both(oracle, F, vol, tf, cam, WH, sr=sr, mode=mode)
    assert (e0[n0 > 0] < 0).all(), "every ray of a camera inside the box starts at negative t"
    assert (sref < n0)[n0 > 40].mean() > 0.3
    assert np.array_equal(steps, sref), int((steps != sref).sum())
    assert np.abs(out - ref).max() <= FWD_TOL


@pytest.mark.parametrize("vshape,WH,R", [((2, 2, 2), (8, 8), 2), ((5, 7, 3), (3, 5), 1), ((13, 12, 14), (1, 1), 4),
                                         ((12, 13, 25), (17, 9), 7)])
def test_tiny_and_ragged_shapes(oracle, F, vshape, WH, R):
    rng = np.random.RandomState(1)
    vol = rng.uniform(0.1, 0.9, size=vshape).astype(np.float32)
    tf = rng.uniform(0.05, 0.5, size=(R, 4)).astype(np.float32)
    cam = oracle.in_circles(2.5)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vshape)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, 0)
    e, x, r, n = gpu_setup(F, cam, WH, vshape)
    out, steps = F.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 1.0)
    assert np.array_equal(steps[0].cpu().numpy(), sref)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    g = rng.randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, g)
    dv, dt = F.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 1.0, T(g[None]), out)
    if np.abs(dv0).max() > 0:
        assert grad_close(dv.cpu().numpy(), dv0)[0]
    if np.abs(dt0).max() > 0:
        assert grad_close(dt.cpu().numpy(), dt0)[0]


@pytest.mark.parametrize("R", [2030, 2048, 3392, 3393, 3413, 3414, 4096, 10176, 10177, 10240, 10241, 16384])
def test_large_tf_falls_back_to_supported_kernels(oracle, hiplib, R):
    """The reference has no limit on the TF resolution. A TF too large for the LDS budget of the fast kernels makes
    dr_workspace_bytes() report 0 and AUTO run the plain kernels -- same results, no error -- and those keep what fits in LDS:
    TF + double-precision d_tf table up to R = 3392, the TF alone up to 10176 (d_tf through float atomics), nothing beyond
    (the double table had silently lowered the backward's limit from 5120 to 3413 entries; ADVICE r05: the tiers
    switched at a full 160 KiB, which the runtime does not grant -- R = 3409..3413 and 10227..10240 failed at launch; they switch
    at 159 KiB now, and both sides of every boundary, old and new, are exercised here)."""
    from differender_amd import functional as Fn
    vol, _, cam = scene(oracle, N=24)
    tf = oracle.bench_tf(R, 0.03)
    tf[:, 3] = np.linspace(0.01, 0.06, R)
    WH = (16, 16)
    ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
    assert (ws is not None) == (R <= 2030)   # the fast kernels' LDS budget (csrc/march_flat.hip: brick_path_supported)
    e, x, r, n = Fn.ray_setup(T(cam[None]), WH, vol.shape, 1.0)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 4096, 1.0, 0)
    g = np.random.RandomState(3).randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 4096, 1.0, g)
    for variant in ((0, 1) if ws is not None else (0,)):
        out, steps = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 4096, 1.0, variant=variant, workspace=ws)
        assert np.array_equal(steps[0].cpu().numpy(), sref)
        assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
        dv, dt = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 4096, 1.0, T(g[None]), out, variant=variant, workspace=ws)
        assert grad_close(dv.cpu().numpy(), dv0)[0] and grad_close(dt.cpu().numpy(), dt0)[0]
    for mode in (1,):   # the non-differentiable march with the oversize table
        refn, _ = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 4096, 1.0, mode)
        outn, _ = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 4096, 1.0, mode, workspace=ws)
        assert np.abs(outn[0].cpu().numpy() - refn).max() <= FWD_TOL


@pytest.mark.parametrize("mode", [0, 1])
def test_high_sampling_rate_with_termination(oracle, F, mode):
    """sr = 8 (the reference renders its ground truth at 8x, OPT.py:67) with a terminating TF: long segments that span
    many 64-sample chunks, the alpha pre-pass and the exact termination sample."""
    vol, tf, cam = scene(oracle, N=64, tf="peaks", R=128, cam_i=2.6)
    tf[:, 3] = np.linspace(0.0, 0.5, 128)
    WH = (96, 80)
    ref, sref, out, steps, (e0, x0, r0, n0), _ = _fwd_both(oracle, F, vol, tf, cam, WH, sr=8.0, mode=mode)
    assert n0.max() > 1500 and (sref < n0)[n0 > 100].mean() > 0.5
    assert np.array_equal(steps, sref), int((steps != sref).sum())
    assert np.abs(out - ref).max() <= FWD_TOL


def test_huge_strides_take_the_64bit_path(oracle, hiplib):
    """The flat kernels address the brick box with 32-bit in-box offsets (flat_strides_ok); a volume view whose
    strides exceed that (a slice of a much larger allocation) must still render and differentiate correctly."""
    from differender_amd import functional as Fn
    vol_h, tf_h, cam_h = scene(oracle, N=16, R=16, alpha=0.05)
    WH = (24, 24)
    big = 48_000_000                                     # elements between x-planes: 3 * 14 * big >= 2^31
    store = torch.zeros(16 * big, dtype=torch.float32, device=dev())
    vol_v = store.as_strided((16, 16, 16), (big, 16, 1))
    vol_v.copy_(T(vol_h))
    tf, cam = T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = Fn.ray_setup(cam, WH, vol_v.shape, 1.0)
    ws = Fn.alloc_workspace(1, WH, vol_v.shape, tf.shape[0], dev())
    out_v, _ = Fn.march_fwd(vol_v, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws)
    g = torch.ones_like(out_v)
    dv_v, dt_v = Fn.march_bwd(vol_v, tf, cam, e, x, r, n, 4096, 1.0, g, out_v, workspace=ws)
    vol_c = T(vol_h)
    ws2 = Fn.alloc_workspace(1, WH, vol_c.shape, tf.shape[0], dev())
    out_c, _ = Fn.march_fwd(vol_c, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws2)
    dv_c, dt_c = Fn.march_bwd(vol_c, tf, cam, e, x, r, n, 4096, 1.0, g, out_c, workspace=ws2)
    assert float((out_v - out_c).abs().max()) <= FWD_TOL
    ok, err = grad_close(dv_v.cpu().numpy(), dv_c.cpu().numpy())
    assert ok, err
    ok, err = grad_close(dt_v.cpu().numpy(), dt_c.cpu().numpy())
    assert ok, err


def test_stale_workspace_and_mixed_variants(oracle, hiplib):
    """The workspace is scratch: whatever bytes an earlier call left in it must not matter; and the baseline backward
    (which needs no workspace) may follow a forward of the fast path."""
    from differender_amd import functional as Fn
    vol_h, tf_h, cam_h = scene(oracle, N=40, R=32, tf="peaks")
    WH = (48, 40)
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, 1.0)
    ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev())
    ws.fill_(0x5A)                                        # stale bytes from "another call"
    out, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws)
    g = T(np.random.default_rng(3).standard_normal(out.shape).astype(np.float32))
    eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
    dv_o, dt_o = oracle.march_bwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, g[0].cpu().numpy())
    for vb in (0, 1):
        dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, variant=vb, workspace=ws)
        ok, err = grad_close(dv.cpu().numpy(), dv_o)
        assert ok, (vb, err)
        ok, err = grad_close(dt.cpu().numpy(), dt_o)
        assert ok, (vb, err)


def test_backward_does_not_trust_a_workspace_it_did_not_fill(oracle, hiplib):
    """The fast backward takes brick records, live flags, ray flags and work items from the workspace as the forward left
    them. A workspace that does NOT hold this call's forward -- garbage bytes under a forward served by the plain kernels,
    or the coarse tape of another call -- must never be indexed: the fingerprint in the header does not match, B1 does
    nothing and B2 marches every ray (slow, and the oracle's result)."""
    from differender_amd import functional as Fn
    import differender_amd._native as N
    vol_h, tf_h, cam_h = scene(oracle, N=40, R=32, tf="peaks")
    WH = (48, 40)
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, 1.0)
    eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
    g_h = np.random.default_rng(5).standard_normal((1, *WH, 4)).astype(np.float32)
    g = T(g_h)
    dv_o, dt_o = oracle.march_bwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, g_h[0])

    # (1) garbage workspace, forward by the plain kernels (which clear the mark), backward asks for the fast path
    ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev())
    ws.fill_(0x5A)
    out, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, variant=N.DR_VARIANT_BASELINE, workspace=ws)
    assert int(Fn.workspace_stats(ws)[3]) == 0            # the mark word: nobody's
    dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, workspace=ws)
    torch.cuda.synchronize()
    assert int(Fn.workspace_stats(ws)[9]) == 0x5A5A5A5A + 1   # counted (the header was garbage: the counter started there)
    ok, err = grad_close(dv.cpu().numpy(), dv_o)
    assert ok, err
    ok, err = grad_close(dt.cpu().numpy(), dt_o)
    assert ok, err

    # (2) garbage workspace and no forward at all on it
    ws.fill_(0xA5)
    dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, workspace=ws)
    torch.cuda.synchronize()
    ok, err = grad_close(dv.cpu().numpy(), dv_o)
    assert ok, err

    # (3) the workspace holds ANOTHER call's forward (other camera, other ray buffers): same sizes, wrong tape
    cam2 = T(np.atleast_2d(oracle.in_circles(2.1)))
    e2, x2, r2, n2 = Fn.ray_setup(cam2, WH, vol.shape, 1.0)
    Fn.march_fwd(vol, tf, cam2, e2, x2, r2, n2, 4096, 1.0, workspace=ws)
    dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, workspace=ws)
    torch.cuda.synchronize()
    ok, err = grad_close(dv.cpu().numpy(), dv_o)
    assert ok, err
    ok, err = grad_close(dt.cpu().numpy(), dt_o)
    assert ok, err

    assert int(Fn.workspace_stats(ws)[9]) == 1            # ... and counted, for the host layer's warning
    # (4) and the matching pair still takes the fast path: B2 marches nothing but irregular rays
    out4, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws)
    dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out4, workspace=ws)
    assert int(Fn.workspace_stats(ws)[9]) == 0
    # the camera may be handed over as a fresh copy every call (hosts do `cam.contiguous()` on an expanded look_from):
    # its address is not part of the fingerprint
    out5, _ = Fn.march_fwd(vol, tf, cam.clone(), e, x, r, n, 4096, 1.0, workspace=ws)
    dv, dt = Fn.march_bwd(vol, tf, cam.clone(), e, x, r, n, 4096, 1.0, g, out5, workspace=ws)
    assert int(Fn.workspace_stats(ws)[9]) == 0
    ok, err = grad_close(dv.cpu().numpy(), dv_o)
    assert ok, err


def test_termination_hints_choose_a_path_never_a_result(oracle, hiplib):
    """DR_HINT_* (include/differender_hip.h) only choose between ways of computing the same image: a right hint skips or
    refines the alpha pre-pass, a WRONG "no early termination" is detected on the device and repaired (every ray of the
    view marched whole: the oracle's image and gradients, workspace header word 8 counts the view)."""
    from differender_amd import functional as Fn
    import differender_amd._native as N
    WH = (40, 32)
    cam_h = oracle.in_circles(0.9)
    for tfkind in ("bench", "peaks"):
        vol_h, tf_h, _ = scene(oracle, N=40, R=32, tf=tfkind, alpha=0.004)
        vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
        e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, 1.0)
        eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
        ref, ref_steps = oracle.march_fwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, 0)
        g_h = np.random.default_rng(9).standard_normal((1, *WH, 4)).astype(np.float32)
        dv_o, dt_o = oracle.march_bwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, g_h[0])
        terminates = bool((ref_steps < nh).any())
        assert terminates == (tfkind == "peaks")
        for hint in (0, N.DR_HINT_NO_EARLY_TERMINATION, N.DR_HINT_EARLY_TERMINATION):
            ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev())
            out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws, hints=hint)
            assert np.array_equal(steps[0].cpu().numpy(), ref_steps), (tfkind, hint)
            assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL, (tfkind, hint)
            dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_h), out, workspace=ws)
            ok, err = grad_close(dv.cpu().numpy(), dv_o)
            assert ok, (tfkind, hint, err)
            ok, err = grad_close(dt.cpu().numpy(), dt_o)
            assert ok, (tfkind, hint, err)
            wrong = terminates and hint == N.DR_HINT_NO_EARLY_TERMINATION
            st = Fn.workspace_stats(ws)
            assert int(st[8]) == (1 if wrong else 0), (tfkind, hint, st[:10])
            if wrong:
                assert int(st[2]) == int((nh > 0).sum())      # every ray that hits was marched whole
    # unknown hint bits and contradicting hints are rejected
    for bad in (0x800, N.DR_HINT_NO_EARLY_TERMINATION | N.DR_HINT_EARLY_TERMINATION):
        with pytest.raises(RuntimeError):
            Fn.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, hints=bad)


@pytest.mark.parametrize("sr", [1.0, 3.0, 8.0])
def test_nondiff_render_under_the_termination_hint(oracle, hiplib, sr):
    """DR_HINT_EARLY_TERMINATION on a non-differentiable render: the alpha pre-pass runs front to back in groups of brick
    layers at every sampling rate, rays that have crossed alpha 0.99 drop out of later groups. Image and step counts must be
    what the sequential oracle gives -- also for a view whose TF cannot terminate at all, and from inside the volume.
    (Round 3 also built the "one-pass" variant -- colour in the grouped pass, a fix-up launch for the crossing segments;
    this test passed on it, it was 7 % slower and is gone: profiles/r03_ab_experiments.txt.)"""
    from differender_amd import functional as Fn
    import differender_amd._native as N
    vol_h = oracle.synth_volume(48)
    WH = (40, 48)
    tf_a = oracle.peaks_tf(32)
    tf_b = oracle.bench_tf(32, 0.001)                       # this view's TF never reaches alpha 0.99
    tf_c = oracle.peaks_tf(32); tf_c[:, 3] = np.minimum(1.0, 3.0 * tf_c[:, 3])
    cams_h = np.stack([oracle.in_circles(0.4), oracle.in_circles(2.0), np.array([0.2, 0.1, -0.3], np.float32)]).astype(np.float32)
    tfs_h = np.stack([tf_a, tf_b, tf_c]).astype(np.float32)
    vol, tfs, cams = T(vol_h), T(tfs_h), T(cams_h)
    e, x, r, n = Fn.ray_setup(cams, WH, vol.shape, sr)
    outs = {}
    for hint in (0, N.DR_HINT_EARLY_TERMINATION):
        ws = Fn.alloc_workspace(3, WH, vol.shape, 32, dev())
        out, steps = Fn.march_fwd(vol, tfs, cams, e, x, r, n, 1 << 20, sr, N.DR_MODE_NONDIFF, workspace=ws, hints=hint)
        outs[hint] = (out.cpu().numpy(), steps.cpu().numpy())
        st = Fn.workspace_stats(ws)
        assert int(st[0]) == 0, st[:10]                      # no ray needed the count-check repair
    terminated = 0
    for v in range(3):
        eh, xh, rh, nh = (t[v].cpu().numpy() for t in (e, x, r, n))
        ref, ref_steps = oracle.march_fwd(vol_h, tfs_h[v], cams_h[v], eh, xh, rh, nh, 1 << 20, sr, 1)
        terminated += int((ref_steps < nh).sum())
        for hint, (o, s) in outs.items():
            reg = nh != 1
            assert np.array_equal(s[v][reg], ref_steps[reg]), (v, hint)
            assert np.abs(o[v] - ref).max(-1)[reg].max() <= FWD_TOL, (v, hint)
    assert terminated > 500


def test_automatic_hints_follow_the_tensor_not_its_address(oracle, hiplib):
    """hints="auto": the largest alpha of a TF tensor is learnt asynchronously once the same tensor VERSION has been seen
    twice, and forgotten when the tensor is written to; a temporary that merely reuses an address never inherits it."""
    from differender_amd import functional as Fn
    import differender_amd._native as N
    vol_h, tf_h, cam_h = scene(oracle, N=32, R=16, tf="bench", alpha=0.004)
    WH = (24, 24)
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = Fn.ray_setup(cam, WH, vol.shape, 1.0)
    H = Fn._TerminationHints()
    args = (vol.shape, 1.0, 4096, N.DR_MODE_DIFF)
    assert H.hints(tf, *args) == 0                     # first sighting
    assert H.hints(tf, *args) == 0                     # second: the copy is started
    torch.cuda.synchronize()
    assert H.hints(tf, *args) == N.DR_HINT_NO_EARLY_TERMINATION
    tf[:, 3] = 0.9                                     # in-place write: new version -- "no termination" is withdrawn at once
    assert H.hints(tf, *args) == 0 and H.hints(tf, *args) == 0
    torch.cuda.synchronize()
    assert H.hints(tf, *args) == N.DR_HINT_EARLY_TERMINATION
    # a TF that is written to before every call (OPT.py:87 clamps it in place each iteration): the harmless hint survives on
    # the last reading, which is refreshed every few calls; the other hint never comes from a stale reading
    for k in range(20):
        tf.clamp_(0.0, 1.0)
        assert H.hints(tf, *args) == N.DR_HINT_EARLY_TERMINATION
    tf[:, 3] = 0.001
    got = []
    for k in range(20):
        tf.clamp_(0.0, 1.0)
        got.append(H.hints(tf, *args)); torch.cuda.synchronize()
    assert N.DR_HINT_NO_EARLY_TERMINATION not in got and got[-1] == 0       # stale readings never say "no termination"
    for k in range(2):                                                       # the tensor comes to rest: read at this version
        H.hints(tf, *args); torch.cuda.synchronize()
    assert H.hints(tf, *args) == N.DR_HINT_NO_EARLY_TERMINATION             # ... an exact reading does say it
    tf[3, 3] = float("nan")                            # a NaN alpha: anything can happen, no hint
    H.hints(tf, *args); H.hints(tf, *args); torch.cuda.synchronize()
    assert H.hints(tf, *args) == 0
    # temporaries: a fresh tensor per call never inherits anything (the entry keeps its predecessor's memory alive, so the
    # allocator cannot hand the same address out again) ...
    for k in range(4):
        t = T((tf_h * 1.0).astype(np.float32))
        assert H.hints(t, *args) == 0
        del t
    # ... while `.detach()` and other views of ONE tensor -- new Python objects, same storage and version counter -- do
    tf3 = T(tf_h)
    assert H.hints(tf3.detach(), *args) == 0 and H.hints(tf3.detach(), *args) == 0
    torch.cuda.synchronize()
    assert H.hints(tf3.detach(), *args) == N.DR_HINT_NO_EARLY_TERMINATION
    tf3.mul_(0.5)                                      # a write through the base is seen by every view
    assert H.hints(tf3.detach(), *args) == 0
    # writes torch's version counter does NOT see (`tf.data`, raw pointers): the reading is refreshed every REFRESH_EVERY-th
    # call all the same, so a stale "no termination" dies out by itself ...
    tf4 = T(tf_h)
    for k in range(3):
        H.hints(tf4, *args); torch.cuda.synchronize()
    assert H.hints(tf4, *args) == N.DR_HINT_NO_EARLY_TERMINATION
    v0 = tf4._version
    tf4.data[:, 3] = 0.9
    assert tf4._version == v0
    got = []
    for k in range(H.REFRESH_EVERY + 3):
        got.append(H.hints(tf4, *args)); torch.cuda.synchronize()
    assert got[-1] == N.DR_HINT_EARLY_TERMINATION, got
    # ... and the device's verdict (workspace header word 8, fed back by VolumeRaycaster._watch_workspace) withdraws it from the
    # TF that was given the hint -- for a while (tests/test_gpu_boundary.py: scope and expiry), not from TFs seen afterwards
    tf5 = T(tf_h)
    for k in range(3):
        H.hints(tf5, *args); torch.cuda.synchronize()
    assert H.hints(tf5, *args) == N.DR_HINT_NO_EARLY_TERMINATION
    with pytest.warns(RuntimeWarning, match="DR_HINT_NO_EARLY_TERMINATION"):
        H.report_wrong_hint()
    for k in range(4):
        assert H.hints(tf5, *args) == 0; torch.cuda.synchronize()
    tf6 = T(tf_h)                                      # a tensor first seen afterwards is judged on its own readings
    for k in range(3):
        H.hints(tf6, *args); torch.cuda.synchronize()
    assert H.hints(tf6, *args) == N.DR_HINT_NO_EARLY_TERMINATION
    # and through march_fwd the automatic hint reproduces the unhinted image bit for bit
    tf2 = T(tf_h)
    outs = []
    for k in range(4):
        out, _ = Fn.march_fwd(vol, tf2, cam, e, x, r, n, 4096, 1.0)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert Fn._hints.amax(tf2) == (pytest.approx(0.004), True)
    base, _ = Fn.march_fwd(vol, tf2, cam, e, x, r, n, 4096, 1.0, hints=0)
    for o in outs:
        assert np.array_equal(o, base.cpu().numpy())


def test_many_views_without_prepass(oracle, hiplib):
    """More than 48 views in one call (the per-view termination flags once lived in the 2 KiB workspace header and the
    alpha pre-pass was skipped beyond 48 views; they now have their own array behind the header and the pre-pass runs
    for any number of views). Results must match the oracle view by view."""
    from differender_amd import functional as Fn
    vol_h = oracle.synth_volume(24)
    tf_h = oracle.peaks_tf(32)
    V, WH = 50, (12, 10)
    cams_h = np.stack([oracle.in_circles(0.37 * v) for v in range(V)]).astype(np.float32)
    vol, tf, cams = T(vol_h), T(tf_h), T(cams_h)
    e, x, r, n = Fn.ray_setup(cams, WH, vol.shape, 1.0)
    ws = Fn.alloc_workspace(V, WH, vol.shape, tf.shape[0], dev())
    out, steps = Fn.march_fwd(vol, tf, cams, e, x, r, n, 4096, 1.0, workspace=ws)
    g = torch.ones_like(out)
    dv, dt = Fn.march_bwd(vol, tf, cams, e, x, r, n, 4096, 1.0, g, out, workspace=ws)
    dv_ref = np.zeros_like(vol_h); dt_ref = np.zeros_like(tf_h)
    for v in (0, 17, 49):
        eh, xh, rh, nh = (t[v].cpu().numpy() for t in (e, x, r, n))
        ref, sref = oracle.march_fwd(vol_h, tf_h, cams_h[v], eh, xh, rh, nh, 4096, 1.0, 0)
        assert np.abs(out[v].cpu().numpy() - ref).max() <= FWD_TOL
        assert np.array_equal(steps[v].cpu().numpy(), sref)
    for v in range(V):
        eh, xh, rh, nh = (t[v].cpu().numpy() for t in (e, x, r, n))
        a, b = oracle.march_bwd(vol_h, tf_h, cams_h[v], eh, xh, rh, nh, 4096, 1.0, np.ones((WH[0], WH[1], 4), np.float32))
        dv_ref += a; dt_ref += b
    ok, err = grad_close(dv.cpu().numpy(), dv_ref)
    assert ok, err
    ok, err = grad_close(dt.cpu().numpy(), dt_ref)
    assert ok, err


def test_image_bands_reproduce_the_whole_image(oracle, F):
    """One view rendered as three bands of rows (one per GPU in a strong-scaling run): ray buffers are bit-identical to
    the corresponding rows of the whole-image call (jitter included), RGBA agrees, and the bands' gradients add up
    to the whole image's."""
    vol_h, tf_h, cam_h = scene(oracle, N=40, R=32, tf="peaks")
    W, H = 50, 36
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    fn = F._f
    seed = 777
    e, x, r, n = fn.ray_setup(cam, (W, H), vol.shape, 1.0, jitter_seed=seed)
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0)
    g = T(np.random.default_rng(9).standard_normal(out.shape).astype(np.float32))
    dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out)
    # against the oracle, whole image (the bands are then compared with these rows)
    eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
    ref, _ = oracle.march_fwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, 0)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    from differender_amd.distributed import shard_rows
    dv_sum = torch.zeros_like(dv); dt_sum = torch.zeros_like(dt)
    for rank in range(3):
        row0, nr = shard_rows(W, rank, 3)
        rows = (row0, W)
        eb, xb, rb, nb = fn.ray_setup(cam, (nr, H), vol.shape, 1.0, jitter_seed=seed, rows=rows)
        sl = slice(row0, row0 + nr)
        assert torch.equal(eb, e[:, sl]) and torch.equal(xb, x[:, sl]) and torch.equal(rb, r[:, sl]) and torch.equal(nb, n[:, sl])
        ws = fn.alloc_workspace(1, (nr, H), vol.shape, tf.shape[0], dev()) if F.variant != 1 else None
        ob, sb = fn.march_fwd(vol, tf, cam, eb, xb, rb, nb, 4096, 1.0, variant=F.variant, workspace=ws, rows=rows)
        assert torch.equal(sb, steps[:, sl])
        assert float((ob - out[:, sl]).abs().max()) <= 2e-6
        dvb, dtb = fn.march_bwd(vol, tf, cam, eb, xb, rb, nb, 4096, 1.0, g[:, sl].contiguous(), ob, variant=F.variant,
                                workspace=ws, rows=rows)
        dv_sum += dvb; dt_sum += dtb
    ok, err = grad_close(dv_sum.cpu().numpy(), dv.cpu().numpy(), 2e-5)
    assert ok, err
    ok, err = grad_close(dt_sum.cpu().numpy(), dt.cpu().numpy(), 2e-5)
    assert ok, err


@pytest.mark.parametrize("seed", range(12))
def test_random_configurations(oracle, hiplib, seed):
    """Fuzz: random volume shapes, image shapes, TF sizes and contents, cameras (any direction and distance, sometimes
    inside the volume), sampling rates, jitter, modes -- fast path against the oracle on the same ray buffers."""
    from differender_amd import functional as Fn
    rng = np.random.default_rng(1000 + seed)
    vshape = tuple(int(v) for v in rng.integers(9, 45, 3))
    WH = (int(rng.integers(5, 40)), int(rng.integers(5, 40)))
    R = int(rng.choice([2, 5, 16, 64, 200]))
    sr = float(rng.choice([0.6, 1.0, 1.5, 2.0, 3.5]))
    mode = int(rng.integers(0, 2))
    vol_h = rng.random(vshape, dtype=np.float32)
    # smooth it a little so that normals are not pure noise, keep it in [0, 1]
    for ax in range(3):
        vol_h = (vol_h + np.roll(vol_h, 1, ax) + np.roll(vol_h, -1, ax)) / 3.0
    vol_h = np.ascontiguousarray(vol_h.astype(np.float32))
    tf_h = rng.random((R, 4), dtype=np.float32)
    tf_h[:, 3] *= float(rng.choice([0.02, 0.2, 0.9]))          # from "never terminates" to "terminates at once"
    d = rng.standard_normal(3); d /= np.linalg.norm(d)
    if abs(d[1]) > 0.97:                                       # looking (anti)parallel to the up vector is degenerate in VR.py:143
        d = np.array([0.6, 0.3, 0.74]); d /= np.linalg.norm(d)
    cam_h = (d * float(rng.choice([0.5, 1.2, 2.5, 6.0]))).astype(np.float32)
    jitter = int(rng.integers(0, 2)) * 12345
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = Fn.ray_setup(cam, WH, vshape, sr, jitter_seed=jitter)
    eh, xh, rh, nh = (t[0].cpu().numpy() for t in (e, x, r, n))
    ws = Fn.alloc_workspace(1, WH, vshape, R, dev())
    out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, 5000, sr, mode, workspace=ws)
    ref, sref = oracle.march_fwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 5000, sr, mode)
    assert np.isfinite(ref).all()
    assert np.array_equal(steps[0].cpu().numpy(), sref), (vshape, WH, R, sr, mode, cam_h)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL, (vshape, WH, R, sr, mode, cam_h)
    if mode == 0:
        g = rng.standard_normal(out.shape).astype(np.float32)
        dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 5000, sr, T(g), out, workspace=ws)
        dv_o, dt_o = oracle.march_bwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 5000, sr, g[0])
        ok, err = grad_close(dv.cpu().numpy(), dv_o)
        assert ok, (err, vshape, WH, R, sr, cam_h)
        ok, err = grad_close(dt.cpu().numpy(), dt_o)
        assert ok, (err, vshape, WH, R, sr, cam_h)


def _support(O, vol_h, tf_h, cam_h, rays, WH, mask):
    """Voxels / texels that receive ANY contribution from the pixels in `mask` (two probe gradients, so that a
    cancellation in one of them cannot hide a voxel)."""
    eh, xh, rh, nh = rays
    sv = np.zeros(vol_h.shape, bool); st = np.zeros(tf_h.shape, bool)
    for seed in (0, 1):
        g = np.zeros((*WH, 4), np.float32)
        g[mask] = np.random.default_rng(seed).uniform(0.5, 1.5, size=(int(mask.sum()), 4)).astype(np.float32)
        a, b = O.march_bwd(vol_h, tf_h, cam_h, eh, xh, rh, nh, 4096, 1.0, g)
        sv |= a != 0; st |= b != 0
    return sv, st


def test_backward_dynamic_range_of_upstream_gradient(oracle, F):
    """Half the image carries an upstream gradient 1e-5 times the other half's. The voxels that only the small-gradient
    rays touch must still come out right ELEMENTWISE. Round 1's fixed point with ONE global scale flushed them to zero;
    now every brick has its own scale, and bricks whose pixel footprint holds both kinds of rays accumulate in double."""
    vol_h, tf_h, cam_h = scene(oracle, N=48, R=32, alpha=0.03, cam_i=0.3)
    tf_h[:, 3] = np.linspace(0.01, 0.08, 32)
    WH = (64, 64)
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0, 30.0, 0.1, 0, 0)
    rays = tuple(t[0].cpu().numpy() for t in (e, x, r, n))
    out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0)
    g = np.random.default_rng(21).standard_normal((*WH, 4)).astype(np.float32)
    small = np.zeros(WH, bool); small[WH[0] // 2:] = True
    g[small] *= 1e-5
    dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g[None]), out)
    dv, dt = dv.cpu().numpy(), dt.cpu().numpy()
    dv0, dt0 = oracle.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, g)
    assert grad_close(dv, dv0)[0] and grad_close(dt, dt0)[0]                    # the usual whole-tensor criterion
    big_sup, _ = _support(oracle, vol_h, tf_h, cam_h, rays, WH, ~small)
    only_small = (dv0 != 0) & ~big_sup
    assert only_small.sum() > 2000
    # the oracle restricted to the small rays is the cleaner reference for those voxels (no f32 absorption)
    g_s = g.copy(); g_s[~small] = 0.0
    dvs, _ = oracle.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, g_s)
    err = np.abs(dv - dvs)[only_small]
    ref = np.abs(dvs)[only_small]
    # relative to the voxel's own value, plus a floor for sums that cancel -- relative to the SMALL half's own maximum,
    # i.e. 2e-11 of the tensor's: five orders of magnitude below anything one global fixed-point scale could resolve
    bound = 1e-4 * ref + 2e-6 * np.abs(dvs).max()
    bad = err > bound
    assert not bad.any(), (int(bad.sum()), int(bad.size), float((err / bound).max()))
    if F.variant == 0:
        from differender_amd.functional import workspace_stats
        assert int(workspace_stats(F._ws)[4]) > 0, "the bricks along the boundary must have switched to double accumulators"
    # d_tf: every texel sees both halves; the small half's share is 1e-5 of it and must at least not disturb the sum
    assert np.abs(dt - dt0).max() <= 1e-4 * np.abs(dt0).max()


def test_backward_with_non_finite_upstream_gradient(oracle, F):
    """NaN / inf / absurdly large entries in grad_out: the fast path drops NaN adjoints, clamps the rest and keeps such
    pixels from setting any brick's fixed-point scale, so every voxel and texel the bad rays do NOT touch is what the
    oracle computes -- elementwise -- and the bad rays' own voxels are finite. (The reference turns a NaN pixel into NaN
    in everything that ray touches and then zeroes those entries with nan_to_num, VR.py:463-475; the plain kernels keep
    that behaviour.)"""
    vol_h, tf_h, cam_h = scene(oracle, N=32, R=32, tf="peaks")
    WH = (24, 24)
    vol, tf, cam = T(vol_h), T(tf_h), T(np.atleast_2d(cam_h))
    e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0, 30.0, 0.1, 0, 0)
    rays = tuple(t[0].cpu().numpy() for t in (e, x, r, n))
    out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0)
    g = np.random.default_rng(4).standard_normal((1, *WH, 4)).astype(np.float32)
    g_ok = g.copy()
    bad = np.zeros(WH, bool)
    for (i, j) in ((3, 4), (10, 11), (17, 5)):
        g_ok[0, i, j] = 0.0; bad[i, j] = True
    dv_o, dt_o = oracle.march_bwd(vol_h, tf_h, cam_h, *rays, 4096, 1.0, g_ok[0])
    # 1) a NaN pixel alone: its ray is dropped, everything else is untouched
    g_nan = g.copy(); g_nan[0, 3, 4, 0] = np.nan; g_nan[0, 10, 11] = 0.0; g_nan[0, 17, 5] = 0.0
    dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_nan), out)
    if F.variant == 0:
        assert torch.isfinite(dv).all() and torch.isfinite(dt).all()
        # components of the NaN pixel that are finite still contribute; compare where that ray has no say
        sv, st = _support(oracle, vol_h, tf_h, cam_h, rays, WH, bad)
        assert np.abs(dv.cpu().numpy() - dv_o)[~sv].max() <= 1e-4 * np.abs(dv_o).max()
        # elementwise: relative to the voxel's own value plus 5e-6 of the tensor's maximum (the f32 oracle itself is ~5e-4
        # of the maximum away from its f64 twin on this opaque scene: tools/elem_probe.py)
        d = np.abs(dv.cpu().numpy() - dv_o)[~sv]
        assert (d <= 1e-4 * np.abs(dv_o)[~sv] + 5e-6 * np.abs(dv_o).max()).all(), float(d.max() / np.abs(dv_o).max())
    # 2) NaN, inf and -3e30 together
    g_bad = g.copy()
    g_bad[0, 3, 4, 0] = np.nan; g_bad[0, 10, 11, 3] = np.inf; g_bad[0, 17, 5, 1] = -3e30
    dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_bad), out)
    if F.variant == 0:  # (the baseline kernels propagate NaN like the reference; RaycastFunction applies nan_to_num)
        assert torch.isfinite(dv).all() and torch.isfinite(dt).all()
        sv, st = _support(oracle, vol_h, tf_h, cam_h, rays, WH, bad)
        assert (~sv & (dv_o != 0)).sum() > 1000
        d = np.abs(dv.cpu().numpy() - dv_o)[~sv]
        assert (d <= 1e-4 * np.abs(dv_o)[~sv] + 5e-6 * np.abs(dv_o).max()).all(), float(d.max() / np.abs(dv_o).max())
    # 2b) an overflowed loss: EVERY pixel's upstream gradient at +-3e38 (clamped contributions of ~1e38 meet in the float atomics
    # that combine bricks and work items: they saturate at +-FLT_MAX, as the reference's nan_to_num would have left them)
    if F.variant == 0:
        g_huge = np.sign(g).astype(np.float32) * np.float32(3e38)
        dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_huge), out)
        assert torch.isfinite(dv).all() and torch.isfinite(dt).all()
        assert float(dv.abs().max()) > 1e30 and float(dt.abs().max()) > 1e30      # (it did overflow the ordinary range)
        # ... also when the backward is served by the per-ray second pass alone (a workspace that is not this forward's)
        from differender_amd import functional as Fn
        ws_other = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev()); ws_other.zero_()
        dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_huge), out, workspace=ws_other)
        assert int(Fn.workspace_stats(ws_other)[9]) == 1
        assert torch.isfinite(dv).all() and torch.isfinite(dt).all()
        g_nan_all = g.copy(); g_nan_all[0, ::2, :, 1] = np.nan
        dv, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_nan_all), out, workspace=ws_other)
        assert torch.isfinite(dv).all() and torch.isfinite(dt).all()
    # 3) the same call with the three pixels zeroed matches the oracle in the usual sense
    dv2, dt2 = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, T(g_ok), out)
    ok, err = grad_close(dv2.cpu().numpy(), dv_o)
    assert ok, err
    ok, err = grad_close(dt2.cpu().numpy(), dt_o)
    assert ok, err


def test_module_gradients_are_finite_with_a_nan_pixel(oracle, hiplib):
    """Through RaycastFunction (which skips nan_to_num on the fast path): one NaN in the upstream gradient must not
    produce a non-finite or absurd step for the optimiser."""
    from differender_amd.volume_raycaster import Raycaster
    N_ = 24
    vol_f = oracle.synth_volume(N_); tf_f = oracle.peaks_tf(16)
    vol_u = T(vol_f).permute(1, 2, 0).contiguous()[None].requires_grad_(True)
    tf_u = T(tf_f).t().contiguous().requires_grad_(True)
    rc = Raycaster((N_, N_, N_), (24, 24), 16, jitter=False, max_samples=4096)
    img = rc(vol_u, tf_u, T(oracle.in_circles(0.5)))
    w = torch.ones_like(img); w[0, 5, 7] = float("nan")
    (img * w).sum().backward()
    assert torch.isfinite(vol_u.grad).all() and torch.isfinite(tf_u.grad).all()
    clean = Raycaster((N_, N_, N_), (24, 24), 16, jitter=False, max_samples=4096)
    v2 = vol_u.detach().clone().requires_grad_(True); t2 = tf_u.detach().clone().requires_grad_(True)
    clean(v2, t2, T(oracle.in_circles(0.5))).sum().backward()
    assert float(vol_u.grad.abs().max()) <= 1.5 * float(v2.grad.abs().max())


def _ct_like_scene(N=72, R=64, poison=None):
    """An object in air: voxels 0 outside a ball, 0.35 .. 0.9 inside; the TF is exactly transparent below intensity 0.3."""
    ax = np.linspace(-1, 1, N, dtype=np.float32)
    z, y, x = np.meshgrid(ax, ax, ax, indexing="ij")
    r = np.sqrt(x * x + y * y + z * z)
    vol = np.where(r < 0.45, 0.6 + 0.25 * np.sin(9 * x) * np.cos(7 * y) + 0.05 * z, 0.0).astype(np.float32)
    if poison is not None:   # one voxel far out in the air
        vol[3, 4, 5] = poison
    tf = np.zeros((R, 4), np.float32)
    k = np.arange(R) / (R - 1)
    tf[:, 0] = 0.5 + 0.5 * np.sin(6 * k); tf[:, 1] = k; tf[:, 2] = 1 - k
    tf[:, 3] = np.where(k > 0.3, 0.05 + 0.1 * k, 0.0)
    return vol, tf


@pytest.mark.parametrize("mode,sr", [(0, 1.0), (0, 2.0), (1, 4.0)], ids=["diff", "diff_sr2", "nondiff_sr4"])
def test_object_in_air(oracle, hiplib, mode, sr):
    """A CT-like scene: most bricks hold nothing but exactly transparent samples (whole passes of the forward take the
    unlit path, the lazily evaluated opacity and normal-tap coordinates are never needed there), the rays that hit the
    ball terminate inside it. Same image, step counts and gradients as the oracle, which marches every sample in full."""
    from differender_amd import functional as Fn
    vol, tf = _ct_like_scene()
    cam = oracle.in_circles(0.7)
    WH = (64, 64)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape, sr=sr)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, mode)
    vt, tt, ct = T(vol), T(tf), T(cam[None])
    e, x, r, n = Fn.ray_setup(ct, WH, vol.shape, sr)
    ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev())
    out, steps = Fn.march_fwd(vt, tt, ct, e, x, r, n, 1 << 20, sr, mode, workspace=ws)
    assert int(Fn.workspace_stats(ws)[0]) == 0                  # no ray failed its sample count
    assert np.array_equal(steps[0].cpu().numpy(), sref)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    assert ref[..., 3].max() > 0.5 and (ref[..., 3] == 0).mean() > 0.3   # the ball is there, and so is the air around it
    if mode == 0:
        g = np.random.RandomState(3).randn(*WH, 4).astype(np.float32)
        dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, g)
        dv, dt = Fn.march_bwd(vt, tt, ct, e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws)
        ok, err = grad_close(dt.cpu().numpy(), dt0); assert ok, f"d_tf rel err {err}"
        ok, err = grad_close(dv.cpu().numpy(), dv0); assert ok, f"d_vol rel err {err}"


@pytest.mark.parametrize("poison", [float("nan"), 5.0, -1.0, 0.31], ids=["nan", "above_one", "negative", "just_visible"])
def test_object_in_air_with_odd_voxels(oracle, hiplib, poison):
    """One odd voxel out in the air: a NaN sends its samples to TF entries 0 and 1, one above 1 to the last entry, a negative
    one clamps to entry 0, and one just inside the visible range must be rendered. (An infinite or > 2^31 / R voxel is outside
    the oracle's domain: its float -> int conversion is undefined in C.)"""
    from differender_amd import functional as Fn
    vol, tf = _ct_like_scene(N=48, poison=poison)
    cam = oracle.in_circles(2.1)
    WH = (48, 48)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape, sr=1.0)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, 0)
    vt, tt, ct = T(vol), T(tf), T(cam[None])
    e, x, r, n = Fn.ray_setup(ct, WH, vol.shape, 1.0)
    ws = Fn.alloc_workspace(1, WH, vol.shape, tf.shape[0], dev())
    out, steps = Fn.march_fwd(vt, tt, ct, e, x, r, n, 1 << 20, 1.0, 0, workspace=ws)
    got = out[0].cpu().numpy()
    assert np.array_equal(steps[0].cpu().numpy(), sref)
    assert np.array_equal(np.isnan(got), np.isnan(ref))
    fin = ~np.isnan(ref)
    assert np.abs(got[fin] - ref[fin]).max() <= FWD_TOL


@pytest.mark.parametrize("vshape,cam", [((1100, 20, 16), (0.4, 0.3, 2.4)), ((24, 1300, 20), (2.2, 0.2, 0.9)), ((16, 28, 1990), (1.9, 0.4, 1.6))],
                         ids=["long-x", "long-y", "long-z"])
def test_long_volume_both_normal_taps_leave_the_cell(oracle, hiplib, vshape, cam):
    """Edges beyond 991 voxels: delta = 1e-3 world units exceeds half a voxel along the long axis, so the +delta AND the -delta tap
    of one sample can both sit outside the centre's cell. The shared-lerp taps (dr_brick_common.h) then read both extra rows /
    planes (template flavour NARROW = false; every other test of the suite runs NARROW = true). Forward and gradients against
    the oracle, fast path and baseline kernels."""
    from differender_amd import functional as Fn
    rng = np.random.RandomState(8)
    vol = (0.5 + 0.45 * np.sin(0.37 * np.arange(vshape[0]))[:, None, None] * np.cos(0.9 * np.arange(vshape[1]))[None, :, None]
           * np.sin(0.5 + 0.21 * np.arange(vshape[2]))[None, None, :]).astype(np.float32)
    vol += rng.uniform(-0.02, 0.02, size=vshape).astype(np.float32)
    tf = oracle.bench_tf(32, 0.02); tf[:, 3] = np.linspace(0.004, 0.05, 32)
    cam = np.array(cam, np.float32)
    WH = (40, 32)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vshape)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, 0)
    g = rng.randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, 1.0, g)
    vt, tt, ct = T(vol), T(tf), T(cam[None])
    e, x, r, n = Fn.ray_setup(ct, WH, vshape, 1.0)
    assert np.array_equal(n[0].cpu().numpy(), n0) and int(n0.max()) > 1000
    ws = Fn.alloc_workspace(1, WH, vshape, 32, dev())
    assert ws is not None
    out, steps = Fn.march_fwd(vt, tt, ct, e, x, r, n, 1 << 20, 1.0, workspace=ws)
    assert int(Fn.workspace_stats(ws)[0]) == 0
    assert np.array_equal(steps[0].cpu().numpy(), sref)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    dv, dt = Fn.march_bwd(vt, tt, ct, e, x, r, n, 1 << 20, 1.0, T(g[None]), out, workspace=ws)
    ok, err = grad_close(dv.cpu().numpy(), dv0); assert ok, f"d_vol rel err {err}"
    ok, err = grad_close(dt.cpu().numpy(), dt0); assert ok, f"d_tf rel err {err}"
    _, dt_only = Fn.march_bwd(vt, tt, ct, e, x, r, n, 1 << 20, 1.0, T(g[None]), out, want_vol=False, workspace=ws)
    ok, err = grad_close(dt_only.cpu().numpy(), dt0); assert ok, f"d_tf (TF-only backward) rel err {err}"


def _ct_like(O, N):
    """The field inside a ball of radius 0.6, air (exactly 0) outside it -- most bricks hold nothing but transparent samples."""
    vol = O.synth_volume(N)
    ax = np.linspace(-1.0, 1.0, N, dtype=np.float32)
    r2 = ax[:, None, None] ** 2 + ax[None, :, None] ** 2 + ax[None, None, :] ** 2
    return np.where(r2 < 0.36, vol, np.float32(0.0)).astype(np.float32)


@pytest.mark.parametrize("mode,sr", [(1, 1.0), (1, 4.0), (1, 8.0), (0, 1.0), (0, 2.0)])
def test_empty_bricks_and_unlit_segments(oracle, hiplib, mode, sr):
    """Round 5: bricks in which no sample composites anything (air under the reference's tf1 preset) deliver their rays' sample
    counts from the ends of each segment and evaluate nothing (march_flat.hip: brick_empty_*), and in non-differentiable renders
    the alpha pre-pass tells the colour march which (ray, layer) segments hold no sample with alpha > 1e-3. Both are ways of
    NOT computing zeros: images, sample counts and -- through the untouched backward -- gradients stay the oracle's."""
    from differender_amd import functional as Fn
    from differender_amd.utils import get_tf
    N, WH, R = 96, (72, 64), 64
    vol = _ct_like(oracle, N)
    tf = get_tf("tf1", R).t().contiguous().numpy()
    cam = oracle.in_circles(0.8)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape, sr=sr)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, mode)
    assert (sref < n0)[n0 > 10].mean() > 0.05 and (sref == n0)[n0 > 10].mean() > 0.05   # terminating rays and rays through air only
    e, x, r, n = Fn.ray_setup(T(cam[None]), WH, vol.shape, sr)
    for hint in (0, Fn.N.DR_HINT_EARLY_TERMINATION):
        ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
        out, steps = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, mode, workspace=ws, hints=hint)
        st = Fn.workspace_stats(ws)
        assert int(st[0]) == 0, "rays failed their sample count and were marched one by one"
        # ... and both mechanisms DID run (workspace header words 12 / 13): this scene is mostly air, its rays cross unlit TF ranges
        assert int(st[13]) > 5, f"only {int(st[13])} of the sampled workgroups (1 in 64) took the empty-brick path"
        assert int(st[12]) > 5, f"only {int(st[12])} segments were skipped by the colour march (sampled: 1 workgroup in 64)"
        assert np.array_equal(steps[0].cpu().numpy(), sref)
        assert np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    if mode == 0:   # the backward re-marches every live sample itself (alpha = 0 has a slope): d_tf of the air's texels included
        g = np.random.RandomState(9).randn(*WH, 4).astype(np.float32)
        dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, g)
        dv, dt = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws)
        assert grad_close(dv.cpu().numpy(), dv0)[0] and grad_close(dt.cpu().numpy(), dt0)[0]
        assert np.abs(dt0[0]).max() > 0   # texel 0 (air) does receive a gradient
        # a backward that wants d_volume ONLY skips the bricks the forward decided empty (opacity 0 and a flat alpha: every d_volume
        # term of their samples vanishes): same d_volume
        dv_only, none = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws, want_tf=False)
        assert none is None and grad_close(dv_only.cpu().numpy(), dv0)[0]
        assert float((dv_only - dv).abs().max()) <= 1e-6 * float(dv.abs().max())


def test_empty_brick_test_is_exact_at_its_edges(oracle, hiplib):
    """The empty-brick decision rests on the staged voxel range and a texel of slack: a brick that is air except for ONE voxel
    (an interior one, an apron one, a NaN) must not be taken for empty, and a TF whose first lit texel sits right above the air's
    must be honoured. Fast path against the sequential kernels (the oracle's twin), image and counts."""
    from differender_amd import functional as Fn
    N, WH, R = 48, (48, 48), 32
    cam = oracle.in_circles(0.5)
    tf = np.zeros((R, 4), np.float32)
    tf[:, :3] = 0.7
    tf[2:, 3] = 0.2          # texels 0 and 1 composite nothing, texel 2 and up do
    e, x, r, n = Fn.ray_setup(T(cam[None]), WH, (N, N, N), 2.0)
    for kind in ("interior", "apron", "nan", "inf", "just_lit", "just_unlit"):
        vol = np.zeros((N, N, N), np.float32)
        if kind == "interior":
            vol[18, 19, 20] = 0.9
        elif kind == "apron":
            vol[24, 24, 12] = 0.9          # on the plane shared by two bricks (cells 12.. belong to the next brick)
        elif kind == "nan":
            vol[30, 17, 22] = np.nan
        elif kind == "inf":
            vol[:] = np.inf                # every intensity indexes the LAST texel (lit), whatever the range arithmetic makes of it
        elif kind == "just_lit":
            vol[:] = np.float32(1.02 / (R - 1))  # index 1.02: lerps texel 1 (0) with texel 2 (0.2): alpha 0.004, lit in both modes
        else:
            vol[:] = np.float32(0.98 / (R - 1))  # index 0.98: texels 0 and 1 only: nothing composites
        for mode in (0, 1):
            ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
            out, steps = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 2.0, mode, workspace=ws)
            assert int(Fn.workspace_stats(ws)[0]) == 0
            outb, stepsb = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, 2.0, mode, variant=1)
            assert torch.equal(steps, stepsb), (kind, mode)
            a, b = out.cpu().numpy(), outb.cpu().numpy()
            assert np.array_equal(np.isnan(a), np.isnan(b)), (kind, mode)
            assert np.nanmax(np.abs(a - b), initial=0.0) <= FWD_TOL, (kind, mode)
            if kind in ("interior", "apron", "just_lit"):
                assert float(np.nanmax(b)) > 0.0   # the lit voxel is seen
            if kind == "just_unlit":
                assert float(np.abs(b).max()) == 0.0


@pytest.mark.parametrize("sr", [1.0, 2.0])
def test_volume_only_backward_skips_unlit_segments(oracle, hiplib, sr):
    """A backward that wants d_volume only drops the (ray, layer) segments the forward's alpha pre-pass found unlit -- opacity 0 and
    both TF texels' alphas exactly 0 at every sample: nothing of such a sample reaches d_volume -- and the bricks it found empty.
    Dense scene (no air) under the reference's tf1 preset, whose transparent ranges make a third of the segments unlit."""
    from differender_amd import functional as Fn
    from differender_amd.utils import get_tf
    N, WH, R = 96, (64, 56), 64
    vol = oracle.synth_volume(N)
    tf = get_tf("tf1", R).t().contiguous().numpy()
    cam = oracle.in_circles(2.1)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape, sr=sr)
    ref, sref = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, 0)
    g = np.random.RandomState(4).randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, g)
    e, x, r, n = Fn.ray_setup(T(cam[None]), WH, vol.shape, sr)
    for hint in (0, Fn.N.DR_HINT_EARLY_TERMINATION):
        ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
        out, steps = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, workspace=ws, hints=hint)
        st = Fn.workspace_stats(ws)
        assert int(st[0]) == 0 and int(st[14]) == 1      # no repairs; the forward left one word of masks per ray
        assert np.array_equal(steps[0].cpu().numpy(), sref) and np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
        dv_only, _ = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws, want_tf=False)
        assert grad_close(dv_only.cpu().numpy(), dv0)[0]
        dv, dt = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws)
        assert grad_close(dv.cpu().numpy(), dv0)[0] and grad_close(dt.cpu().numpy(), dt0)[0]
        assert float((dv_only - dv).abs().max()) <= 1e-6 * float(dv.abs().max())
    # a TF without exact zeros (every sample has a slope in alpha): no segment may be dropped -- same gradients again
    tf2 = tf.copy(); tf2[:, 3] = np.maximum(tf2[:, 3], 1e-4)
    ref2, _ = oracle.march_fwd(vol, tf2, cam, e0, x0, r0, n0, 1 << 20, sr, 0)
    dv2, _ = oracle.march_bwd(vol, tf2, cam, e0, x0, r0, n0, 1 << 20, sr, g)
    ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
    out2, _ = Fn.march_fwd(T(vol), T(tf2), T(cam[None]), e, x, r, n, 1 << 20, sr, workspace=ws)
    assert np.abs(out2[0].cpu().numpy() - ref2).max() <= FWD_TOL
    dvo, _ = Fn.march_bwd(T(vol), T(tf2), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out2, workspace=ws, want_tf=False)
    assert grad_close(dvo.cpu().numpy(), dv2)[0]


def test_rays_longer_than_the_per_sample_tape_take_the_per_ray_kernels(oracle, hiplib):
    """DR_TAPE_TF reserves, per ray, the longest march the volume allows at the CALL's sampling rate. Ray buffers made for another rate
    (here: ray_setup at rate 2, march at rate 1 -- legal through the C ABI, the march takes its samples from the buffers) hold rays
    longer than that: they must not write past their slot of the tape. F2 hands them to the per-ray kernels, counted; results as the
    oracle's for the same buffers."""
    from differender_amd import functional as Fn
    N, WH, R = 48, (24, 20), 32
    vol_h = oracle.synth_volume(N)
    tf_h = oracle.bench_tf(R, 0.02)
    tf_h[:, 3] = np.linspace(0.0, 0.03, R)
    cam_h = oracle.in_circles(1.1)
    S = 1 << 20
    vol, tf, cam = T(vol_h), T(tf_h), T(cam_h[None])
    e, x, r, n = Fn.ray_setup(cam, WH, vol_h.shape, 2.0)            # twice the samples the tape of a rate-1 call has room for
    ws = Fn.alloc_workspace(1, WH, vol_h.shape, R, dev(), tape=(S, 1.0))
    stride = (int(np.floor(2.0 * np.sqrt(3.0) * np.sqrt(3.0) * (N - 1))) + 2 + 1) & ~1
    assert int(n.max()) > stride
    out, steps = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, 1.0, workspace=ws, tape=True)
    long_rays = int((n > stride).sum())
    assert int(Fn.workspace_stats(ws)[2]) >= long_rays > 0    # marched by the per-ray kernels, and counted
    eo, xo, ro, no = oracle.ray_setup(cam_h, *WH, vol_h.shape, sr=2.0)
    ref, sref = oracle.march_fwd(vol_h, tf_h, cam_h, eo, xo, ro, no, S, 1.0, 0)
    assert np.array_equal(steps[0].cpu().numpy(), sref) and np.abs(out[0].cpu().numpy() - ref).max() <= FWD_TOL
    g = np.random.RandomState(4).randn(1, *WH, 4).astype(np.float32)
    _, dt = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, 1.0, T(g), out, want_vol=False, workspace=ws, tape=True)
    _, dt_ref = oracle.march_bwd(vol_h, tf_h, cam_h, eo, xo, ro, no, S, 1.0, g[0], want_vol=False)
    ok, err = grad_close(dt.cpu().numpy(), dt_ref, tol=1e-3)   # (rays of the plain kernels' float-atomic d_tf among them: their bar)
    assert ok, err


@pytest.mark.parametrize("sr", [2.0, 4.0, 8.0])
def test_sub_ulp_contributions_behind_opaque_structures_D4(oracle, hiplib, sr):
    """a TF whose transparent ranges carry a tiny alpha (1e-6) instead of 0, around opaque peaks, at sampling rates
    >= 2. Behind an opaque structure every such sample contributes T * L * rgb * op ~ 1e-8 to a composite of ~0.8: below half an
    ulp, so SEQUENTIAL float32 compositing (the reference's loop, VR.py:300-302; the oracle; the baseline kernels) drops it, sample
    after sample, while the brick kernels sum a segment's samples among themselves first -- 1.3e-5 (rate 2) / 2.8e-5 (rate 4) on the
    worst pixel until round 6, and a termination decision now and then (alpha stagnates below 0.99 in the sequential recurrence).
    Round 6: the per-ray passes bound that effect from the partials (d4_risk, dr_brick_common.h); the crossing search widens its
    exact-walk band by the bound, and a ray whose image bound exceeds 2e-6 is recomputed sample by sample (ray_exact_kernel). The
    bar is the ordinary one: sample counts bit-exact, RGBA within 1e-5 of the float32 oracle."""
    from differender_amd import functional as Fn
    from differender_amd.utils import get_tf
    N, WH, R = 96, (64, 56), 64
    vol = oracle.synth_volume(N)
    tf = get_tf("tf1", R).t().contiguous().numpy()
    tf[:, 3] = np.where(tf[:, 3] == 0, np.float32(1e-6), tf[:, 3])
    cam = oracle.in_circles(2.1)
    e0, x0, r0, n0 = oracle.ray_setup(cam, *WH, vol.shape, sr=sr)
    ref32, s32 = oracle.march_fwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, 0)
    e, x, r, n = Fn.ray_setup(T(cam[None]), WH, vol.shape, sr)
    ws = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
    out, steps = Fn.march_fwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, workspace=ws)
    st = Fn.workspace_stats(ws)
    assert np.array_equal(steps[0].cpu().numpy(), s32)
    assert np.abs(out[0].cpu().numpy() - ref32).max() <= FWD_TOL
    assert int(st[15]) > 0 and int(st[0]) == 0          # the exact pass ran (and nothing was "repaired")
    # the gradients of the same call: B1 works from F2's own final composite (`fin`), not from the rewritten image
    g = np.random.RandomState(5).randn(*WH, 4).astype(np.float32)
    dv0, dt0 = oracle.march_bwd(vol, tf, cam, e0, x0, r0, n0, 1 << 20, sr, g)
    dv, dt = Fn.march_bwd(T(vol), T(tf), T(cam[None]), e, x, r, n, 1 << 20, sr, T(g[None]), out, workspace=ws)
    assert grad_close(dv.cpu().numpy(), dv0)[0] and grad_close(dt.cpu().numpy(), dt0)[0]
    # the preset itself has exact zeros there: nothing for the exact pass to do
    tf0 = get_tf("tf1", R).t().contiguous().numpy()
    ws0 = Fn.alloc_workspace(1, WH, vol.shape, R, dev())
    out0, steps0 = Fn.march_fwd(T(vol), T(tf0), T(cam[None]), e, x, r, n, 1 << 20, sr, workspace=ws0)
    ref0, s0 = oracle.march_fwd(vol, tf0, cam, e0, x0, r0, n0, 1 << 20, sr, 0)
    assert np.array_equal(steps0[0].cpu().numpy(), s0) and np.abs(out0[0].cpu().numpy() - ref0).max() <= FWD_TOL
    assert int(Fn.workspace_stats(ws0)[15]) <= 0.02 * WH[0] * WH[1], int(Fn.workspace_stats(ws0)[15])


@pytest.mark.parametrize("case", ["sr1", "sr2_terminating", "sr0.6_jitter_clipped", "views_per_view_tf", "f16", "sr4_noise_R300"])
def test_tf_only_backward_over_the_per_sample_tape(oracle, hiplib, case):
    """DR_TAPE_TF (BASELINE config C3: the gradient w.r.t. the transfer function alone): the forward leaves (intensity, lighting) of
    every marched sample on a tape, the backward is a per-ray pass over it (csrc/tf_tape.hip) -- no brick, no tap. Same image (to an
    ulp or two) as without the flag; d_tf within the bar of the oracle's (VR.py:460-461,470-471 restated) AND of the brick-centric
    TF-only backward it replaces. (sr4_noise_R300: rays of 300-440 samples -- passes of four samples per lane -- through white noise
    under a 300-entry TF: consecutive samples hardly ever share a TF cell, so a lane's four samples are up to four runs of d_tf.)"""
    from differender_amd import functional as Fn
    from differender_amd.utils import get_tf
    N, WH, R = 64, (40, 48), (300 if case == "sr4_noise_R300" else 64)
    vol_h = oracle.synth_volume(N)
    if case == "sr4_noise_R300":
        vol_h = np.random.RandomState(11).rand(N, N, N).astype(np.float32)
    sr, S, seed, V = 1.0, 1 << 20, 0, 1
    tf_h = oracle.bench_tf(R, 0.02)
    tf_h[:, 3] = np.linspace(0.0, 0.05, R)
    if case == "sr2_terminating":
        sr = 2.0
        tf_h = get_tf("tf1", R).t().contiguous().numpy()
    if case == "sr0.6_jitter_clipped":
        sr, S, seed = 0.6, 40, 777
    if case == "sr4_noise_R300":
        sr = 4.0
        tf_h[:, 3] = np.linspace(0.0, 0.01, R)
    cams = np.stack([oracle.in_circles(0.3), oracle.in_circles(2.2)])[: (2 if case == "views_per_view_tf" else 1)]
    V = len(cams)
    tfs_h = np.stack([tf_h, np.clip(tf_h * 1.3, 0, 1)])[:V] if case == "views_per_view_tf" else tf_h
    vol = T(vol_h.astype(np.float16)) if case == "f16" else T(vol_h)
    vol_o = vol_h.astype(np.float16).astype(np.float32) if case == "f16" else vol_h
    tf, cam = T(tfs_h), T(cams)
    e, x, r, n = Fn.ray_setup(cam, WH, vol_h.shape, sr, jitter_seed=seed)
    g = np.random.RandomState(2).randn(V, *WH, 4).astype(np.float32)
    ws_t = Fn.alloc_workspace(V, WH, vol_h.shape, R, dev(), tape=(S, sr))
    ws_b = Fn.alloc_workspace(V, WH, vol_h.shape, R, dev())
    assert ws_t.numel() > ws_b.numel()
    out_t, st_t = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, workspace=ws_t, tape=True)
    out_b, st_b = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, workspace=ws_b)
    # (the same samples in the same arithmetic; where the untaped march drops unlit segments at listing, the flat sample order of a
    #  wave -- hence the association of a few segments' scans -- differs: an ulp or two)
    assert float((out_t - out_b).abs().max()) <= 5e-7 and torch.equal(st_t, st_b)
    _, dt_t = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(g), out_t, want_vol=False, workspace=ws_t, tape=True)
    _, dt_b = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(g), out_b, want_vol=False, workspace=ws_b)
    assert int(Fn.workspace_stats(ws_t)[9]) == 0      # the tape was found: no ray went through the per-ray fallback for lack of it
    dt_ref = np.zeros_like(tfs_h)
    for v in range(V):
        eo, xo, ro, no = oracle.ray_setup(cams[v], *WH, vol_h.shape, sr=sr, jitter_seed=seed, view=v)
        _, b = oracle.march_bwd(vol_o, tfs_h[v] if tfs_h.ndim == 3 else tfs_h, cams[v], eo, xo, ro, no, S, sr, g[v], want_vol=False)
        if tfs_h.ndim == 3:
            dt_ref[v] = b
        else:
            dt_ref += b
    ok, err = grad_close(dt_t.cpu().numpy(), dt_ref)
    assert ok, (case, err)
    ok, err = grad_close(dt_t.cpu().numpy(), dt_b.cpu().numpy())
    assert ok, (case, err)
    # the backward does not trust a workspace without this call's tape: asked for the tape pass on the untaped forward's workspace
    # (too small: refused) and on a taped workspace whose forward ran WITHOUT the flag (every ray through the per-ray pass: slow, right)
    with pytest.raises(RuntimeError):
        Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(g), out_b, want_vol=False, workspace=ws_b, tape=True)
    out_n, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, S, sr, workspace=ws_t)
    _, dt_n = Fn.march_bwd(vol, tf, cam, e, x, r, n, S, sr, T(g), out_n, want_vol=False, workspace=ws_t, tape=True)
    assert int(Fn.workspace_stats(ws_t)[9]) == 1
    assert grad_close(dt_n.cpu().numpy(), dt_ref, tol=1e-3)[0]   # (the plain kernels' float-atomic d_tf: their own bar)
