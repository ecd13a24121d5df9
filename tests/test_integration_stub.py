"""INTEGRATION.md's reference-side ctypes stub, extracted from the document and executed VERBATIM.

CPU part: the block parses, binds every symbol it names and its argtypes agree with differender_amd/_native.py.
GPU part: its render() / render_backward() reproduce the oracle (so the document cannot rot)."""
import ctypes
import os
import re
import types

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stub_module(hiplib):
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, flags=re.S)
    assert len(blocks) == 1, "INTEGRATION.md must hold exactly one python block (the stub)"
    from differender_amd import _native as N
    # the stub opens the library by its bare soname; make that name resolve to the in-tree build
    ctypes.CDLL(N.LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    mod = types.ModuleType("differender_hip_stub")
    exec(compile(blocks[0], "INTEGRATION.md", "exec"), mod.__dict__)
    return mod


def test_stub_binds_the_declared_signatures(hiplib):
    from differender_amd import _native as N
    mod = _stub_module(hiplib)
    for name in ("dr_ray_setup", "dr_march_fwd", "dr_march_bwd", "dr_mse_loss_grad", "dr_tf_momentum_step",
                 "dr_workspace_bytes"):
        fn = getattr(mod._lib, name)
        assert list(fn.argtypes) == list(N.SIGNATURES[name][1]), name
    assert mod._lib.dr_error_string(-1) == b"differender_hip: invalid argument"
    with pytest.raises(RuntimeError, match="invalid argument"):
        mod._check(-1)


@pytest.mark.gpu
def test_stub_renders_and_differentiates_like_the_oracle(hiplib, oracle):
    import torch
    mod = _stub_module(hiplib)
    dev = torch.device("cuda:0")
    N_, R, WH = 32, 32, (40, 32)
    vol_h = oracle.synth_volume(N_)
    tf_h = oracle.bench_tf(R, 0.03); tf_h[:, 3] = np.linspace(0.01, 0.06, R)
    cam_h = oracle.in_circles(0.8)
    vr = types.SimpleNamespace(resolution=WH, fov_rad=np.radians(30.0), near=0.1, max_samples=4096)  # VR.py:77-78
    volume = torch.from_numpy(vol_h).to(dev).permute(1, 2, 0).contiguous().permute(2, 0, 1)  # (W,D,H) view, VR.py:571
    tf = torch.from_numpy(tf_h).to(dev)
    out, saved = mod.render(vr, volume, tf, torch.from_numpy(cam_h).to(dev), 1.0)
    ref, _, (e, x, r, n) = oracle.render(vol_h, tf_h, cam_h, WH, S=4096)
    assert np.abs(out.cpu().numpy() - ref).max() <= 1e-5
    g = np.random.default_rng(0).standard_normal((*WH, 4)).astype(np.float32)
    d_vol, d_tf = mod.render_backward(vr, volume, tf, 1.0, saved, torch.from_numpy(g).to(dev))
    dv0, dt0 = oracle.march_bwd(vol_h, tf_h, cam_h, e, x, r, n, 4096, 1.0, g)
    assert d_vol.stride() == volume.stride()
    assert np.abs(d_vol.cpu().numpy() - dv0).max() <= 1e-4 * np.abs(dv0).max()
    assert np.abs(d_tf.cpu().numpy() - dt0).max() <= 1e-4 * np.abs(dt0).max()
    nd, _ = mod.render(vr, volume, tf, torch.from_numpy(cam_h).to(dev), 4.0, mode=1)
    refn, _, _ = oracle.render(vol_h, tf_h, cam_h, WH, sr=4.0, mode=1)
    assert np.abs(nd.cpu().numpy() - refn).max() <= 1e-5
