"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by VIEW (or, for one view, by bands of image rows: shard_rows): every rank holds a full replica of the volume and the transfer function
(512 MiB at 512^3 f32 -- small against 288 GB of HBM), renders its own views, and accumulates a local
d_volume / d_tf over them. The only exchange step is one sum all-reduce of those two shared gradients
(SURVEY section 8(e)); the forward path needs no collective at all. The reference has no counterpart.
"""
import torch

__all__ = ["shard_views", "shard_rows", "all_reduce_gradients", "GradientReducer"]


def shard_views(n_views, rank=None, world_size=None):
    """Indices of the views rank `rank` renders: v = rank (mod world_size), in ascending order."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
    return list(range(rank, n_views, world_size))


def shard_rows(image_rows, rank=None, world_size=None):
    """A single view split into `world_size` bands of image rows: returns (row0, n_rows) of
    rank `rank`; the bands differ by at most one row and tile the image. Pass `rows=(row0, image_rows)` and an output
    shape of (n_rows, H) to functional.ray_setup / march_fwd / march_bwd; the bands' gradients are summed with
    all_reduce_gradients exactly like those of different views."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
    base, extra = divmod(int(image_rows), int(world_size))
    row0 = rank * base + min(rank, extra)
    return row0, base + (1 if rank < extra else 0)


def all_reduce_gradients(grads, group=None, async_op=False):
    """Sum-all-reduce the shared gradients in place (d_volume: one large message, so the ring is
    bandwidth-bound on the xGMI links; d_tf: a few KiB, latency-bound). Large tensors go as they are --
    they are already one contiguous bucket each; small ones are coalesced into a single message."""
    import os
    import torch.distributed as dist
    if not dist.is_available() or not dist.is_initialized():
        return []
    if dist.get_world_size(group) == 1 and os.environ.get("DR_ALLREDUCE_SINGLE_RANK") != "1":
        return []   # (the variable makes a one-rank group go through RCCL all the same: the only way to test this path on one GPU)
    handles = []
    small = [g for g in grads if g is not None and g.numel() * g.element_size() < (1 << 20)]
    large = [g for g in grads if g is not None and g.numel() * g.element_size() >= (1 << 20)]
    # The small bucket goes FIRST: it is reduced synchronously (a few KiB: latency-bound, not worth deferring), and the
    # collectives of a process group run in order on RCCL's stream -- issued behind the large asynchronous reduction, its
    # implicit wait would hold the caller's stream until that one had finished too, and nothing would overlap.
    if small:
        flat = torch.cat([g.reshape(-1) for g in small])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        off = 0
        for g in small:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
    for g in large:
        t = g if g.is_contiguous() or _dense(g) else None
        if t is None:
            c = g.contiguous()
            dist.all_reduce(c, op=dist.ReduceOp.SUM, group=group)
            g.copy_(c)
        else:
            h = dist.all_reduce(_flat_view(t), op=dist.ReduceOp.SUM, group=group, async_op=async_op)
            if async_op:
                handles.append(h)
    return handles  # async_op: call .wait() on each before reading the large tensors


def _dense(t):
    """True if t covers a contiguous block of memory exactly once (e.g. a permuted contiguous tensor)."""
    sizes_strides = sorted(((st, sz) for sz, st in zip(t.shape, t.stride()) if sz > 1), key=lambda p: p[0])
    expect = 1
    for st, sz in sizes_strides:
        if st != expect:
            return False
        expect *= sz
    return True


def _flat_view(t):
    """1-D view over the storage block of a dense tensor (order is irrelevant for an elementwise sum)."""
    return torch.as_strided(t, (t.numel(),), (1,))


class GradientReducer:
    """At most ONE gradient all-reduce in flight, overlapped with the next step's rendering.

    `submit(grads)` first waits for the reduction submitted before (so its buffers may be recycled), then starts the sum
    all-reduce of `grads` asynchronously -- on RCCL's own stream, i.e. concurrent with the kernels the caller enqueues
    next -- and keeps the tensors alive until it has completed. `wait()` blocks the current stream on the reduction in
    flight and returns the reduced tensors of that step (read them only after it). One instance per training loop."""

    def __init__(self, group=None):
        self.group = group
        self._handles, self._grads = [], []

    def submit(self, grads):
        done = self.wait()
        self._grads = [g for g in grads if g is not None]
        self._handles = all_reduce_gradients(self._grads, group=self.group, async_op=True)
        return done

    def wait(self):
        for h in self._handles:
            h.wait()
        done, self._handles, self._grads = self._grads, [], []
        return done
