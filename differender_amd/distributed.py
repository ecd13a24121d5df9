"""Multi-GPU helpers (one process per GPU, torch.distributed; backend "nccl" is RCCL on ROCm).

The hot path shards by VIEW (or, for one view, by bands of image rows: shard_rows): every rank holds a full replica of the volume and the transfer function
(512 MiB at 512^3 f32 -- small against 288 GB of HBM), renders its own views, and accumulates a local
d_volume / d_tf over them. The only exchange step is one sum all-reduce of those two shared gradients
(SURVEY section 8(e)); the forward path needs no collective at all. The reference has no counterpart.
"""
import torch

__all__ = ["shard_views", "shard_rows", "row_work_estimate", "all_reduce_gradients", "GradientReducer"]


def shard_views(n_views, rank=None, world_size=None):
    """Indices of the views rank `rank` renders: v = rank (mod world_size), in ascending order."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
    return list(range(rank, n_views, world_size))


def shard_rows(image_rows, rank=None, world_size=None, weights=None):
    """A single view split into `world_size` bands of image rows: returns (row0, n_rows) of
    rank `rank`; the bands tile the image. Pass `rows=(row0, image_rows)` and an output shape of (n_rows, H) to
    functional.ray_setup / march_fwd / march_bwd; the bands' gradients are summed with all_reduce_gradients exactly like
    those of different views.
    weights: per-row work estimates (row_work_estimate): the bands then carry equal WORK instead of equal rows -- the rows
    through the middle of the volume hold several times the samples of the rows at the image's edge (equal rows: the
    slowest of 8 bands has 1.4 x the mean, profiles/r03_band_prediction_*.txt). Every rank must pass the same weights."""
    if rank is None or world_size is None:
        import torch.distributed as dist
        rank, world_size = dist.get_rank(), dist.get_world_size()
    image_rows, world_size = int(image_rows), int(world_size)
    if weights is None:
        base, extra = divmod(image_rows, world_size)
        row0 = rank * base + min(rank, extra)
        return row0, base + (1 if rank < extra else 0)
    import numpy as np
    w = np.asarray(weights, np.float64)
    if w.shape != (image_rows,) or not np.isfinite(w).all() or (w < 0).any():
        raise ValueError("weights must be one finite, non-negative number per image row")
    if image_rows < world_size:
        raise ValueError(f"cannot cut {image_rows} image rows into {world_size} non-empty weighted bands "
                         "(fewer rows than ranks: use the unweighted split, which allows empty bands)")
    w = w + max(float(w.sum()), 1.0) * 1e-3 / image_rows        # every row costs something (ray setup, image I/O)
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, world_size):                              # first row at which the cumulated work reaches k / G
        c = int(np.searchsorted(cum, cum[-1] * k / world_size, side="left"))
        cuts.append(min(max(c, cuts[-1] + 1), image_rows - (world_size - k)))   # at least one row per band
    cuts.append(image_rows)
    return cuts[rank], cuts[rank + 1] - cuts[rank]


def row_work_estimate(look_from, image_rows, image_cols, fov_deg=30.0, near=0.1, cols=48):
    """Marched samples per image row, up to a factor: the chord through the volume [-1, 1]^3 of `cols` rays per row, from the
    pinhole model of compute_entry_exit (VR.py:127-151, 28-53, 221-259) evaluated on the HOST in numpy (the sample count
    of a ray is its chord length x the sampling rate x the volume diagonal, VR.py:251-253). look_from: three host numbers
    (a camera that only exists on the device would need a synchronising copy: pass its host original). An estimate for
    load balancing only -- it decides who renders which rows, never what is rendered."""
    import numpy as np
    W, H = int(image_rows), int(image_cols)
    o = np.asarray(look_from, np.float64).reshape(3)
    vd = -o / np.linalg.norm(o)
    right = np.cross(vd, [0.0, 1.0, 0.0]); right /= np.linalg.norm(right)
    up = np.cross(right, vd); up /= np.linalg.norm(up)
    near_h = 2.0 * np.tan(np.radians(fov_deg)) * near
    near_w = near_h * (W / H)
    x = (np.arange(W) + 0.5) / W - 0.5
    y = (np.linspace(0, H - 1, min(cols, H)) + 0.5) / H - 0.5
    d = near * vd + x[:, None, None] * near_w * right + y[None, :, None] * near_h * up
    d /= np.linalg.norm(d, axis=-1, keepdims=True)
    with np.errstate(divide="ignore", invalid="ignore"):
        t1 = (-1.0 - o) / d
        t2 = (1.0 - o) / d
    tmin = np.nanmax(np.minimum(t1, t2), axis=-1)
    tmax = np.nanmin(np.maximum(t1, t2), axis=-1)
    hit = ~((tmax < 0.0) | (tmin > tmax))
    return np.where(hit, tmax - tmin, 0.0).sum(axis=1)


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
