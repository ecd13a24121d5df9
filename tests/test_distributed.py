"""N > 1 path on CPU: two gloo ranks shard the views, accumulate local gradients (computed with the
oracle here, since there is no GPU) and all-reduce them; the result must equal the single-process sum."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_views, q):
    try:
        _worker_body(rank, world, port, n_views, q)
    except Exception as exc:  # surface the reason in the parent instead of a bare non-zero exit code
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


def _worker_body(rank, world, port, n_views, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS=("2" if world <= 2 else "1"))
    torch.set_num_threads(1)
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from differender_amd.distributed import shard_views, all_reduce_gradients
    from oracle import oracle as O
    vol = O.synth_volume(20); tf = O.bench_tf(16, 0.03); tf[:, 3] = np.linspace(0.01, 0.06, 16)
    dv = torch.zeros(20, 20, 20).permute(2, 0, 1)  # dense but not contiguous, as the product's d_vol is
    dt = torch.zeros(16, 4)
    mine = shard_views(n_views)
    for v in mine:
        cam = O.in_circles(0.4 * v)
        e, x, r, n = O.ray_setup(cam, 16, 16, vol.shape)
        g = np.random.RandomState(v).randn(16, 16, 4).astype(np.float32)
        a, b = O.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g)
        dv += torch.from_numpy(a); dt += torch.from_numpy(b)
    all_reduce_gradients([dv, dt])
    if rank == 0:
        q.put((mine, dv.contiguous().numpy(), dt.numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_view_sharding_is_a_partition():
    from differender_amd.distributed import shard_views
    for n, w in [(8, 2), (5, 2), (7, 4), (1, 2)]:
        parts = [shard_views(n, r, w) for r in range(w)]
        assert sorted(sum(parts, [])) == list(range(n))


@pytest.mark.parametrize("world,n_views", [(2, 3), (4, 9), (8, 11)], ids=["world2", "world4", "world8"])
def test_gradient_allreduce_matches_single_process(oracle, world, n_views):
    """World sizes 2, 4 and 8 (the driver's scaling run is the first time 8 ranks meet on hardware: the control flow -- view
    sharding with uneven shares, local accumulation, the coalesced small bucket + the dense-but-not-contiguous large one --
    is rehearsed here over gloo)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_views, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    assert got[0] != "error", got
    mine, dv, dt = got
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert mine == list(range(0, n_views, world))
    O = oracle
    vol = O.synth_volume(20); tf = O.bench_tf(16, 0.03); tf[:, 3] = np.linspace(0.01, 0.06, 16)
    dv_ref = np.zeros_like(vol); dt_ref = np.zeros_like(tf)
    for v in range(n_views):
        cam = O.in_circles(0.4 * v)
        e, x, r, n = O.ray_setup(cam, 16, 16, vol.shape)
        g = np.random.RandomState(v).randn(16, 16, 4).astype(np.float32)
        a, b = O.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g)
        dv_ref += a; dt_ref += b
    assert np.abs(dv - dv_ref).max() <= 1e-5 * np.abs(dv_ref).max()
    assert np.abs(dt - dt_ref).max() <= 1e-5 * np.abs(dt_ref).max()


def _reducer_worker(rank, world, port, q):
    try:
        sys.path.insert(0, ROOT)
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        torch.set_num_threads(1)
        import torch.distributed as dist
        dist.init_process_group("gloo", rank=rank, world_size=world)
        from differender_amd.distributed import GradientReducer
        red = GradientReducer()
        results = []
        for step in range(4):
            # what bench.py's step does after the backward: fresh gradient tensors every step, handed to the reducer;
            # the previous step's reduction is collected when the next one is submitted (at most one in flight)
            big = torch.full((64, 64, 64), float((rank + 1) * (step + 1))).permute(2, 0, 1)   # >= 1 MiB, dense, not contiguous
            small = torch.full((16, 4), float(rank + 10 * step))
            done = red.submit([big, small, None])
            if done:
                results.append([t.clone() for t in done])
        results.append([t.clone() for t in red.wait()])
        assert red.wait() == []
        if rank == 0:
            q.put([(float(b.flatten()[0]), float(b.min()), float(b.max()), float(s_.flatten()[0])) for b, s_ in results])
        dist.barrier()
        dist.destroy_process_group()
    except Exception:
        import traceback
        q.put(("error", rank, traceback.format_exc()))
        raise


@pytest.mark.parametrize("world", [2, 8], ids=["world2", "world8"])
def test_overlapped_gradient_reducer(world):
    """The N > 1 control flow of bench.py: asynchronous all-reduce of each step's gradients, at most one in flight,
    buffers kept alive until it completes, every step's result correct -- with 2 ranks and with the 8 of a full node."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    assert not (isinstance(got, tuple) and got[0] == "error"), got
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert len(got) == 4
    tri = world * (world + 1) // 2
    for step, (b0, bmin, bmax, s0) in enumerate(got):
        assert b0 == bmin == bmax == float(tri * (step + 1))                  # sum over ranks of (rank + 1) * (step + 1)
        assert s0 == float(world * (world - 1) // 2 + 10 * step * world)      # sum over ranks of (rank + 10 step)
