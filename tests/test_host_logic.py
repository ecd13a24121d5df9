"""Host-side logic that needs no GPU: batching/layout rules of Raycaster (VR.py:551-571), utils."""
import math

import pytest
import torch


@pytest.fixture(scope="module")
def rc(hiplib):
    from differender_amd.volume_raycaster import Raycaster
    return Raycaster((6, 8, 10), (16, 24), 12, max_samples=64)  # volume (D,H,W) = (6,8,10)


def test_ctor_matches_reference_attributes(rc):
    assert rc.volume_shape == (10, 6, 8)  # (W, D, H), VR.py:481
    assert rc.vr.resolution == (16, 24) and rc.vr.max_samples == 64
    assert "Max Samples = 64" in repr(rc)


def test_determine_batch_unbatched(rc):
    vol = torch.rand(1, 6, 8, 10); tf = torch.rand(4, 12); lf = torch.rand(3)
    b, bs, v, t, l = rc._determine_batch(vol, tf, lf)
    assert (b, bs) == (False, 0)
    assert v.shape == (10, 6, 8) and t.shape == (12, 4) and l.shape == (3,)
    assert torch.equal(v, vol.squeeze(0).permute(2, 0, 1)) and torch.equal(t, tf.permute(1, 0))
    assert v.data_ptr() == vol.data_ptr()  # a view, not the .contiguous() copy of VR.py:571


def test_determine_batch_mixed(rc):
    vol = torch.rand(1, 6, 8, 10); tf = torch.rand(4, 12); lf = torch.rand(5, 3)
    b, bs, v, t, l = rc._determine_batch(vol, tf, lf)
    assert b and bs == 5 and v.shape == (10, 6, 8) and t.shape == (12, 4) and l.shape == (5, 3)
    volb = torch.rand(5, 1, 6, 8, 10); tfb = torch.rand(5, 4, 12)
    b, bs, v, t, l = rc._determine_batch(volb, tfb, lf[0])
    assert b and bs == 5 and v.shape == (5, 10, 6, 8) and t.shape == (5, 12, 4) and l.shape == (5, 3)
    assert torch.equal(v[2], volb[2, 0].permute(2, 0, 1))
    with pytest.raises(ValueError):
        rc._determine_batch(torch.rand(6, 8, 10), tf, lf)


def test_no_cpu_fallback(rc):
    with pytest.raises(RuntimeError, match="no CPU path"):
        rc(torch.rand(1, 6, 8, 10), torch.rand(4, 12), torch.tensor([2.5, 0.7, 0.0]))


def test_alias_package_imports():
    from differender.volume_raycaster import Raycaster, RaycastFunction, VolumeRaycaster  # noqa: F401
    from differender.utils import get_tf, in_circles, get_rand_pos
    tf = get_tf("tf1", 128)
    assert tf.shape == (4, 128) and float(tf.min()) >= 0 and float(tf.max()) <= 1
    # control point (0.3601 -> a=0.3904) .. (0.4475 -> 0.3917): plateau value inside
    assert abs(float(tf[3, int(round(0.40 * 127))]) - 0.391) < 2e-3
    assert get_tf("gray", 16)[3, 0] == pytest.approx(0.02) and get_tf("black", 16).max() == pytest.approx(1e-2)
    with pytest.raises(Exception):
        get_tf("nope", 8)
    c = in_circles(0.5)
    assert c.tolist() == pytest.approx([2.5 * math.cos(0.5), 0.7, 2.5 * math.sin(0.5)])
    p = get_rand_pos(7)
    assert p.shape == (7, 3) and torch.allclose(p.norm(dim=1), torch.full((7,), 2.7), atol=1e-5)


def test_tex_from_pts_is_piecewise_linear():
    from differender_amd.utils import tex_from_pts
    pts = torch.tensor([[0.0, 0, 0, 0, 0.0], [0.5, 1, 0, 0, 1.0], [1.0, 0, 1, 0, 0.0]])
    t = tex_from_pts(pts, 5)
    assert t.shape == (4, 5)
    assert t[0].tolist() == pytest.approx([0, 0.5, 1, 0.5, 0]) and t[3].tolist() == pytest.approx([0, 0.5, 1, 0.5, 0])
    assert t[1].tolist() == pytest.approx([0, 0, 0, 0.5, 1])


def test_reciprocal_division_identity(oracle):
    """The fast kernels compute s/(n-1) from a per-ray reciprocal with one fma correction (dr_device.h
    sample_pos_rcp); that must be the IEEE quotient bit for bit, or sample positions would drift from VR.py:279."""
    import ctypes
    fn = oracle.lib().dro_check_rcp_division
    fn.restype = ctypes.c_long
    fn.argtypes = [ctypes.c_int, ctypes.c_long]
    assert fn(3072, 20_000_000) == 0


def test_shard_rows_tiles_the_image():
    from differender_amd.distributed import shard_rows
    for rows, world in ((512, 8), (100, 3), (7, 8), (1, 1)):
        bands = [shard_rows(rows, r, world) for r in range(world)]
        assert sum(n for _, n in bands) == rows
        pos = 0
        for row0, n in bands:
            assert row0 == pos and n >= 0
            pos += n
        assert max(n for _, n in bands) - min(n for _, n in bands) <= 1


def test_work_balanced_row_bands():
    """shard_rows(weights=...): bands tile the image, every band has a row, and the bands carry equal estimated work; the
    estimate itself is the pinhole + slab test of compute_entry_exit (VR.py:127-151, 28-53) on a coarse grid, checked
    here against the oracle's sample counts."""
    import numpy as np
    from differender_amd.distributed import row_work_estimate, shard_rows
    from oracle import oracle as O
    cam = O.in_circles(0.3)
    W = H = 96
    w = row_work_estimate(cam, W, H)
    e, x, r, n = O.ray_setup(cam, W, H, (64, 64, 64), 1.0)
    true = n.sum(axis=1).astype(np.float64)
    assert np.corrcoef(w, true)[0, 1] > 0.999
    for G in (2, 3, 4, 8):
        bands = [shard_rows(W, k, G, weights=w) for k in range(G)]
        assert bands[0][0] == 0 and all(b[1] >= 1 for b in bands)
        assert all(bands[k][0] + bands[k][1] == bands[k + 1][0] for k in range(G - 1)) and bands[-1][0] + bands[-1][1] == W
        work = np.array([true[r0:r0 + nr].sum() for r0, nr in bands])
        even = np.array([true[r0:r0 + nr].sum() for r0, nr in (shard_rows(W, k, G) for k in range(G))])
        assert work.max() / work.mean() < 1.12, (G, work)
        assert work.max() <= even.max()
    # degenerate weights still give every rank a band
    assert [shard_rows(5, k, 5, weights=np.array([0, 0, 9.0, 0, 0])) for k in range(5)] == [(k, 1) for k in range(5)]
    # fewer rows than ranks: no weighted cut has a row for everybody (the unweighted split hands out empty bands instead)
    with pytest.raises(ValueError, match="fewer rows than ranks"):
        shard_rows(3, 0, 8, weights=np.ones(3))
    assert [shard_rows(3, k, 8) for k in range(8)] == [(0, 1), (1, 1), (2, 1)] + [(3, 0)] * 5
    # the bands of a full node (8 ranks) on the headline image, cut by estimated work
    w8 = row_work_estimate([2.5, 0.7, 0.0], 512, 512)
    bands8 = [shard_rows(512, k, 8, weights=w8) for k in range(8)]
    assert bands8[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(bands8, bands8[1:])) and sum(b[1] for b in bands8) == 512
    work8 = np.array([w8[r0:r0 + nr].sum() for r0, nr in bands8])
    assert work8.max() <= 1.05 * work8.mean()
    assert sum(nr for _, nr in (shard_rows(10, k, 4, weights=np.zeros(10)) for k in range(4))) == 10
    with pytest.raises(ValueError):
        shard_rows(10, 0, 2, weights=np.ones(9))


def test_ssim2d_restatement_against_a_direct_evaluation():
    """differender_amd.utils.ssim2d (restating pytorch_msssim.ssim as examples/test_opt_tf.py:70 calls it; [mem], unpinned) against a
    direct float64 numpy evaluation of the same definition; identical images give 1, the loss is differentiable."""
    import numpy as np
    import torch
    from differender_amd.utils import ssim2d, dssim_mse_loss
    rng = np.random.default_rng(3)
    X = rng.random((2, 3, 24, 20)); Y = np.clip(X + 0.1 * rng.standard_normal(X.shape), 0, 1)
    k = np.arange(11) - 5.0
    w = np.exp(-k * k / (2 * 1.5 ** 2)); w /= w.sum()

    def filt(a):   # valid separable convolution
        a = np.stack([np.tensordot(w, a[:, :, i:i + 11, :], axes=(0, 2)) for i in range(a.shape[2] - 10)], axis=2)
        return np.stack([np.tensordot(a[:, :, :, j:j + 11], w, axes=(3, 0)) for j in range(a.shape[3] - 10)], axis=3)

    mu1, mu2 = filt(X), filt(Y)
    s1, s2, s12 = filt(X * X) - mu1 ** 2, filt(Y * Y) - mu2 ** 2, filt(X * Y) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    m = ((2 * mu1 * mu2 + C1) / (mu1 ** 2 + mu2 ** 2 + C1)) * ((2 * s12 + C2) / (s1 + s2 + C2))
    want = np.maximum(m.reshape(2, 3, -1).mean(-1), 0.0).mean()
    got = ssim2d(torch.from_numpy(X), torch.from_numpy(Y), data_range=1.0, size_average=True, nonnegative_ssim=True)
    assert abs(float(got) - want) < 1e-12
    assert abs(float(ssim2d(torch.from_numpy(X), torch.from_numpy(X), data_range=1.0)) - 1.0) < 1e-12
    xt = torch.from_numpy(X).float().requires_grad_(True)
    loss, dssim, mse = dssim_mse_loss(xt, torch.from_numpy(Y).float())
    loss.backward()
    assert torch.isfinite(xt.grad).all() and float(xt.grad.abs().max()) > 0 and 0 < float(dssim.detach()) < 1


def test_get_tf_generate_returns_a_random_peaked_tf():
    """UT.py:67-70: get_tf('generate', res) = tex_from_pts(TFGenerator(max_num_peaks=2).generate(), res). torchvtk's generator is
    not available; the stand-in keeps the contract (control points in the presets' format, 1-2 opacity peaks, values in [0, 1])."""
    from differender_amd.utils import get_tf
    from differender_amd.utils.utils import TFGenerator
    torch.manual_seed(5)
    seen = set()
    for _ in range(20):
        pts = TFGenerator(peakgen_kwargs={"max_num_peaks": 2}).generate()
        assert pts.shape[1] == 5 and float(pts[0, 0]) == 0.0 and float(pts[-1, 0]) == 1.0
        assert bool((pts[1:, 0] >= pts[:-1, 0]).all()) and float(pts.min()) >= 0.0 and float(pts.max()) <= 1.0
        seen.add((pts.shape[0] - 2) // 4)
        tf = get_tf("generate", 128)
        assert tf.shape == (4, 128) and tf.dtype == torch.float32 and 0.0 < float(tf[3].max()) <= 0.9
        assert float(tf[3, 0]) == 0.0 and float(tf[3, -1]) == 0.0
    assert seen == {1, 2}
    with pytest.raises(Exception, match="Invalid Transfer function identifier"):
        get_tf("nope", 16)


def test_tape_fits_its_memory_cap(hiplib, monkeypatch):
    """The TF-only forward's per-sample tape has a fixed stride per ray, known before any ray is (host arithmetic of the C ABI:
    dr_workspace_bytes_tape). RaycastFunction asks for it only below DIFFERENDER_TAPE_MAX_GIB; above, the brick-centric TF-only
    backward serves the call."""
    from differender_amd import functional as Fn
    from differender_amd.volume_raycaster import tape_fits
    one = Fn.tape_workspace_bytes(1, (512, 512), (512, 512, 512), 256, 1 << 20, 1.0)
    plain = int(Fn.N.lib().dr_workspace_bytes(1, 512, 512, 512, 512, 512, 256))
    stride = (int(math.floor(2.0 * math.sqrt(3.0) * math.sqrt(3.0) * 511)) + 2 + 1) & ~1
    assert 0 <= one - plain - 512 * 512 * stride * 8 < 4096         # the plain workspace + 8 B per ray and possible sample (+ alignment)
    assert Fn.tape_workspace_bytes(1, (512, 512), (512, 512, 512), 256, 1024, 1.0) < one     # max_samples caps the stride
    assert tape_fits(1, (512, 512), (512, 512, 512), 256, 1 << 20, 1.0)                     # 6.4 GB
    assert not tape_fits(8, (1024, 1024), (1024, 1024, 1024), 256, 1 << 20, 1.0)            # 8 x 51 GB
    monkeypatch.setenv("DIFFERENDER_TAPE_MAX_GIB", "1")
    assert not tape_fits(1, (512, 512), (512, 512, 512), 256, 1 << 20, 1.0)
    assert not tape_fits(1, (16, 16), (4000, 8, 8), 16, 64, 1.0)                            # no fast path at all: no tape either
