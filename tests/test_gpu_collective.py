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


def test_single_process_init_all_with_one_device(hiplib):
    """dr_comm_init_all -- ONE process driving N devices (ncclCommInitAll), the set-up SURVEY 8(e) names first -- had never
    executed, not even with N = 1. One device: both spellings of the device list (NULL = devices 0..N-1, and an
    explicit list), then the grouped gradient all-reduce on the communicator it returns."""
    from differender_amd import _native as N
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    s = torch.cuda.current_stream().cuda_stream
    for devices in (None, (ctypes.c_int * 1)(0)):
        comms = (ctypes.c_void_p * 1)()
        N.check(hiplib.dr_comm_init_all(comms, 1, devices), "dr_comm_init_all")
        assert comms[0]
        d_vol = torch.randn(40, 32, 24, device=dev)
        d_tf = torch.randn(128, 4, device=dev)
        want_vol, want_tf = d_vol.clone(), d_tf.clone()
        N.check(hiplib.dr_allreduce_gradients_f32(comms[0], d_vol.data_ptr(), d_vol.numel(), d_tf.data_ptr(), d_tf.numel(), s),
                "dr_allreduce_gradients_f32")
        N.check(hiplib.dr_allreduce_gradients_f32(comms[0], None, 0, d_tf.data_ptr(), d_tf.numel(), s), "dr_allreduce_gradients_f32 (TF only)")
        torch.cuda.synchronize()
        assert torch.equal(d_vol, want_vol) and torch.equal(d_tf, want_tf)
        N.check(hiplib.dr_comm_destroy(comms[0]), "dr_comm_destroy")
    # argument errors are reported, not crashed on
    assert hiplib.dr_comm_init_all(None, 1, None) != 0
    assert hiplib.dr_comm_init_all((ctypes.c_void_p * 1)(), 0, None) != 0


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


_NCCL_ONE_RANK = r"""
import os, sys, socket
sys.path.insert(0, os.environ["DR_ROOT"])
import torch, torch.distributed as dist
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), DR_ALLREDUCE_SINGLE_RANK="1")
dev = torch.device("cuda", 0); torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)      # "nccl" IS RCCL on ROCm
from differender_amd.distributed import GradientReducer, all_reduce_gradients
big = torch.arange(96 * 64 * 80, dtype=torch.float32, device=dev).reshape(96, 64, 80).permute(2, 0, 1)   # dense, not contiguous, > 1 MiB
small = torch.linspace(0, 1, 64, device=dev).reshape(16, 4)
ref_b, ref_s = big.clone(), small.clone()
all_reduce_gradients([big, small, None])
torch.cuda.synchronize()
assert torch.equal(big, ref_b) and torch.equal(small, ref_s)             # a sum over one rank is the identity
red = GradientReducer()
for step in range(3):                                                    # bench.py's pattern: at most one reduction in flight
    g = (big * float(step + 1)).contiguous().permute(1, 2, 0)
    want = g.clone()
    done = red.submit([g, small.clone()])
    last = (g, want)
out = red.wait(); torch.cuda.synchronize()
assert torch.equal(out[0], last[1])
dist.barrier(); dist.destroy_process_group()
print("NCCL_ONE_RANK_OK")
"""


def test_torch_rccl_backend_one_rank_gradient_reduction():
    """torch.distributed's "nccl" backend (= RCCL) on this box, the code path of bench.py --gpus N and of
    differender_amd.distributed -- with the one rank a one-GPU box allows: communicator set-up, the sum all-reduce of a dense
    non-contiguous gradient through its flat storage view, the coalesced small message, the overlapped reducer."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, DR_ROOT=root, HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-c", _NCCL_ONE_RANK], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "NCCL_ONE_RANK_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
