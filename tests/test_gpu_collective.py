"""The C-ABI collective shim (dr_comm_*, dr_allreduce_*): RCCL is loaded on demand and a one-rank communicator
reduces in place (the N-rank exchange itself is covered by the gloo tests on CPU and by the driver's multi-GPU run)."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_one_rank_allreduce_through_the_c_abi(hiplib):
    from differender_amd import _native as N
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    ident = ctypes.create_string_buffer(128)
    N.check(hiplib.dr_comm_unique_id(ident), "dr_comm_unique_id")
    comm = ctypes.c_void_p()
    N.check(hiplib.dr_comm_init_rank(ctypes.byref(comm), 1, ident, 0), "dr_comm_init_rank")
    assert comm.value
    d_vol = torch.randn(32, 24, 40, device=dev).permute(2, 0, 1)          # dense, not contiguous: as the product's d_volume
    d_tf = torch.randn(64, 4, device=dev)
    want_vol, want_tf = d_vol.clone(), d_tf.clone()
    s = torch.cuda.current_stream().cuda_stream
    N.check(hiplib.dr_allreduce_gradients_f32(comm, d_vol.data_ptr(), d_vol.numel(), d_tf.data_ptr(), d_tf.numel(), s),
            "dr_allreduce_gradients_f32")
    N.check(hiplib.dr_allreduce_f32(comm, d_tf.data_ptr(), d_tf.numel(), s), "dr_allreduce_f32")
    torch.cuda.synchronize()
    assert torch.equal(d_vol, want_vol) and torch.equal(d_tf, want_tf)     # the sum over one rank
    N.check(hiplib.dr_comm_destroy(comm), "dr_comm_destroy")


def test_calls_run_on_the_device_of_their_buffers(hiplib, oracle):
    """The library switches to the device that owns the buffers (and back): with one GPU this can only be checked for
    being harmless -- a render issued while another thread-local 'current device' state is in effect still matches."""
    import numpy as np
    from differender_amd import functional as F
    dev = torch.device("cuda:0")
    vol = torch.from_numpy(oracle.synth_volume(16)).to(dev); tf = torch.from_numpy(oracle.bench_tf(8, 0.05)).to(dev)
    cam = torch.from_numpy(np.atleast_2d(oracle.in_circles(0.2))).to(dev)
    e, x, r, n = F.ray_setup(cam, (16, 16), vol.shape, 1.0)
    out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0)
    ref, _, _ = oracle.render(vol.cpu().numpy(), tf.cpu().numpy(), oracle.in_circles(0.2), (16, 16), S=4096)
    assert np.abs(out[0].cpu().numpy() - ref).max() <= 1e-5
    assert torch.cuda.current_device() == 0
