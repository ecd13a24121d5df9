"""Tensor-level wrappers over the C ABI (include/differender_hip.h).

Buffers are torch tensors on a ROCm device; the HIP library allocates nothing and runs on the
current torch stream. Volume tensors live in the reference's field index space (VX, VY, VZ) =
(W, D, H) of the user's (1, D, H, W) tensor (VR.py:481) and may have arbitrary strides, so the
permuted view of VR.py:566,571 is consumed without a copy.
"""
import math
import warnings

import numpy as np
import torch

from . import _native as N

__all__ = ["ray_setup", "march_fwd", "march_bwd", "new_jitter_seed", "alloc_workspace", "workspace_stats", "evaluated_samples",
           "mse_loss_grad", "tf_momentum_step", "as_volume", "bwd_is_sanitised", "termination_hints"]


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _require_gpu(t, name):
    if not t.is_cuda:
        raise RuntimeError(f"differender_amd: `{name}` must live on a ROCm GPU (got {t.device}); "
                           "there is no CPU path")


def as_volume(vol):
    """The reference's setters call `.float()` on whatever they are given (VR.py:118-125): float64, bfloat16 or
    integer volumes are converted; float32 and float16 (storage extension) pass through untouched."""
    return vol if vol.dtype in (torch.float32, torch.float16) else vol.float()


def bwd_is_sanitised(vol, tf, d_vol, workspace, n):
    """True if march_bwd(vol, tf, ..., n, ..., workspace=workspace) is served by the fast kernels, which sanitise their
    gradients themselves (dr_march_bwd_variant -- the function dr_march_bwd_rows itself asks; finite by construction, the float
    atomics that combine bricks saturate at +-FLT_MAX: include/differender_hip.h) -- also when the per-ray second pass ends up
    serving the whole call (a stale workspace, a repaired wrong hint): it sanitises whenever it runs for this path; False for the plain kernels, which propagate NaN like the reference and want the
    reference's nan_to_num. n: the (views, W, H) sample-count buffer of the call."""
    if workspace is None:
        return False
    if workspace.numel() == 0:
        return False
    sv = vol.stride()[-3:]
    sd = d_vol.stride()[-3:] if d_vol is not None else (0, 0, 0)
    V, W, H = (int(v) for v in n.shape)
    return N.lib().dr_march_bwd_variant(V, W, H, *(int(v) for v in vol.shape[-3:]), int(tf.shape[-2]), *sv, *sd,
                                        int(d_vol is not None), N.DR_VARIANT_AUTO, 1) == N.DR_VARIANT_AUTO


def _vol_args(vol, n_views):
    """-> (ptr, dtype_tag, VX, VY, VZ, sx, sy, sz, view_stride)"""
    if vol.dtype == torch.float32:
        tag = N.DR_F32
    elif vol.dtype == torch.float16:
        tag = N.DR_F16
    else:
        raise TypeError(f"volume dtype must be float32 or float16, got {vol.dtype}")
    if vol.ndim == 3:
        vs, (VX, VY, VZ), (sx, sy, sz) = 0, vol.shape, vol.stride()
    elif vol.ndim == 4:
        if vol.shape[0] != n_views:
            raise ValueError(f"batched volume has {vol.shape[0]} items, expected {n_views}")
        vs, (VX, VY, VZ), (sx, sy, sz) = vol.stride(0), vol.shape[1:], vol.stride()[1:]
    else:
        raise ValueError("volume must be (VX,VY,VZ) or (views,VX,VY,VZ)")
    return vol.data_ptr(), tag, VX, VY, VZ, sx, sy, sz, vs


def _tf_args(tf, n_views):
    if tf.dtype != torch.float32 or not tf.is_contiguous():
        raise TypeError("tf must be a contiguous float32 tensor")
    if tf.ndim == 2 and tf.shape[1] == 4:
        return tf.data_ptr(), tf.shape[0], 0
    if tf.ndim == 3 and tf.shape[2] == 4 and tf.shape[0] == n_views:
        return tf.data_ptr(), tf.shape[1], tf.stride(0)
    raise ValueError("tf must be (R,4) or (views,R,4)")


def new_jitter_seed():
    """Non-zero 32-bit seed drawn from torch's CPU generator (so torch.manual_seed controls jitter)."""
    return int(torch.randint(1, 2 ** 31 - 1, (1,)).item())


def alloc_workspace(n_views, out_shape, vol_shape, R, device, tape=None):
    """Scratch buffer of the fast (brick-centric) kernels for one forward(+backward) pair, or None when only
    the baseline kernels can serve this problem (dr_workspace_bytes() == 0). The forward leaves its coarse
    tape here; hand the same buffer to march_bwd.
    tape=(max_samples, sampling_rate): room for the per-sample tape of a DR_TAPE_TF forward as well (march_fwd(tape=True) /
    march_bwd(tape=True): the backward w.r.t. the transfer function alone) -- 8 B per ray and possible sample."""
    W, H = int(out_shape[0]), int(out_shape[1])
    VX, VY, VZ = (int(s) for s in vol_shape)
    if tape is not None:
        nbytes = N.lib().dr_workspace_bytes_tape(int(n_views), W, H, VX, VY, VZ, int(R), int(tape[0]), float(tape[1]))
    else:
        nbytes = N.lib().dr_workspace_bytes(int(n_views), W, H, VX, VY, VZ, int(R))
    if nbytes == 0:
        return None
    return torch.empty(nbytes, dtype=torch.uint8, device=device)


def tape_workspace_bytes(n_views, out_shape, vol_shape, R, max_samples, sampling_rate):
    """Bytes alloc_workspace(..., tape=(max_samples, sampling_rate)) would ask for (0: only the baseline kernels serve this problem).
    The tape has a fixed stride per ray -- the longest march the volume allows at this rate -- so it is known before any ray is:
    6.4 GB for one 512^2 view of a 512^3 volume at rate 1, 49 GB for a 1024^2 view of a 1024^3 volume."""
    W, H = int(out_shape[0]), int(out_shape[1])
    VX, VY, VZ = (int(s) for s in vol_shape)
    return int(N.lib().dr_workspace_bytes_tape(int(n_views), W, H, VX, VY, VZ, int(R), int(max_samples), float(sampling_rate)))


def workspace_stats(workspace):
    """Diagnostics the last forward left in the workspace header: [0] = rays whose segments failed the
    sample-count check and were marched individually (expected 0), [2] = all rays the per-ray fallback marched
    (those plus the irregular ones: single-sample rays), [4] = bricks whose d_volume box accumulated in double (last
    backward), [5] = overflow work items heavy bricks were cut into, [8] = views for which a DR_HINT_NO_EARLY_TERMINATION
    hint turned out wrong (their rays were marched one by one), [9] = backward calls that did not find their forward's
    fingerprint in this workspace and marched every ray one by one, [3] = the forward's fingerprint (0: nobody's),
    [12] = (ray, layer) segments the colour march skipped because the alpha pre-pass had found them unlit, [13] = brick workgroups
    (of all forward passes) that took the empty-brick path -- both SAMPLED (every 64th workgroup reports), 0 when the mechanisms have
    nothing to do; [15] = rays whose pixel was recomputed sample by sample in the reference's sequential float32 order
    (ray_exact_kernel, DESIGN.md D4: runs of contributions of the size of an ulp of the running composite)."""
    return workspace[:64].view(torch.int32).cpu()


def evaluated_samples(workspace):
    """(alpha pre-pass, colour march, backward): samples whose taps the brick kernels actually EVALUATED in the calls that were
    given DR_COUNT_EVALUATED (march_fwd(hints=... | N.DR_COUNT_EVALUATED), march_bwd(count_evaluated=True)) since the last forward
    -- the work-skipping paths (empty bricks, unlit segments) evaluate nothing for samples that still count as marched. bench.py
    prices its roofline with these, not with the marched count. Synchronises."""
    w = workspace[58 * 4:64 * 4].view(torch.int64).cpu()
    return int(w[0]), int(w[1]), int(w[2])


def _ws_args(workspace):
    if workspace is None:
        return None, 0
    return workspace.data_ptr(), workspace.numel()


def _rows(rows, W):
    """rows = None (whole image) or (row0, image_rows): the W buffer rows are image rows [row0, row0 + W)."""
    if rows is None:
        return int(W), 0
    row0, img_w = int(rows[0]), int(rows[1])
    if row0 < 0 or img_w < W or row0 > img_w - W:
        raise ValueError(f"band rows [{row0}, {row0 + W}) do not fit an image of {img_w} rows")
    return img_w, row0


def ray_setup(cam, out_shape, vol_shape, sampling_rate, fov_deg=30.0, near=0.1, jitter_seed=0, view_base=0, rows=None):
    """compute_entry_exit (VR.py:221-259) for cam (views,3) -> entry, exit (views,W,H), rays (views,W,H,3),
    n (views,W,H) int32. rows=(row0, image_rows) renders a band of a taller image (see distributed.shard_rows)."""
    _require_gpu(cam, "look_from")
    cam = cam.to(torch.float32).contiguous()
    V = cam.shape[0]
    W, H = int(out_shape[0]), int(out_shape[1])
    dev = cam.device
    entry = torch.empty((V, W, H), dtype=torch.float32, device=dev)
    exit_ = torch.empty((V, W, H), dtype=torch.float32, device=dev)
    rays = torch.empty((V, W, H, 3), dtype=torch.float32, device=dev)
    n = torch.empty((V, W, H), dtype=torch.int32, device=dev)
    VX, VY, VZ = (int(s) for s in vol_shape)
    with torch.cuda.device(dev):
        rc = N.lib().dr_ray_setup_rows(cam.data_ptr(), V, W, H, *_rows(rows, W), VX, VY, VZ,
                                       float(np.radians(fov_deg)), float(near),
                                       float(sampling_rate), int(jitter_seed) & 0xFFFFFFFF, int(view_base),
                                       entry.data_ptr(), exit_.data_ptr(), rays.data_ptr(), n.data_ptr(), _stream())
    N.check(rc, "dr_ray_setup_rows")
    return entry, exit_, rays, n


class _TerminationHints:
    """DR_HINT_* for march_fwd, derived from the transfer function WITHOUT ever synchronising with the device.

    Whether a ray can reach alpha 0.99 at all is a property of the TF's largest alpha (and of the longest possible ray): the
    library decides it on the device at the start of every forward and gates its alpha pre-pass off -- but the gated
    launches themselves (five per forward: ~45 us at 512^3, ~25 us at 256^3) are still issued, and how finely a pre-pass
    that DOES run should proceed front to back is a host decision too. A TF that is rendered again and again unchanged (a
    fixed TF under volume optimisation, OPT.py; the ground-truth TF; every benchmark) lets the host know: the largest alpha
    of a TF tensor is computed once per tensor VERSION (torch bumps `_version` on every in-place write, for every view and
    `.detach()` of the tensor alike), copied to pinned memory asynchronously, and used by later calls once the copy has landed
    (event.query(), never a wait). A TF seen for the first time gets no hint; one that is written to between calls gets
    "no early termination" never, and "early termination" from its last reading (refreshed every few calls). A stale answer
    (data changed behind the version counter's back) cannot produce a wrong image: the device re-checks and repairs
    (include/differender_hip.h).

    Writes the version counter does NOT see (`tf.data.add_()`, a raw-pointer write: `_version` stays put) would leave a
    reading "exact" for ever, and a TF whose alphas grow under such updates would keep getting "no early termination" -- which
    the device repairs correctly, but by marching every ray of the view with the per-ray kernels (10-40 x slower). Two
    safeguards: the largest alpha is re-read every REFRESH_EVERY-th call even when the version has not moved, and the device's
    own verdict feeds back -- when a workspace header reports a wrong hint (word 8, `report_wrong_hint`), the TFs that were
    recently GIVEN that hint (not every cached TF, and nothing at all if the hint was the caller's own: ADVICE r04) lose it for
    DISTRUST_REFRESHES re-readings of their largest alpha, twice as long after every further report (a wrong "no termination"
    is the only hint that costs anything)."""

    REFRESH_EVERY = 8   # the largest alpha of a TF is re-read on every 8th call at most (whether or not its version moved)
    DISTRUST_REFRESHES = 4   # clean re-readings before a TF that was given a wrong "no termination" hint may get it again

    def __init__(self, capacity=8):
        # key -> entry. The key is WHERE the data lives (storage address, offset, shape, strides) and the entry holds a
        # reference to the tensor, so that the memory cannot be handed to anybody else while the entry exists (a TF is a few
        # KB; at most `capacity` of them are kept): `tf.detach()` -- a new Python object every call, same storage, same
        # version counter -- matches; a temporary such as `tf.permute(1, 0).contiguous()` gets another address every call
        # and never does.
        self._seen = {}
        self._capacity = capacity
        self._handed = []            # keys of the TFs that were most recently given DR_HINT_NO_EARLY_TERMINATION (newest last)
        self._warned = False

    def report_wrong_hint(self):
        """The device found DR_HINT_NO_EARLY_TERMINATION wrong for some view of a recent render (workspace header word 8).
        If that hint came from here, the TFs it was recently derived from had their data moved behind the version counter's
        back: they lose it for a while (see the class docstring). A hint the CALLER passed explicitly is the caller's business:
        nothing is withdrawn. Returns the number of cached TFs that were blamed."""
        blamed = [self._seen[k] for k in self._handed if k in self._seen]
        self._handed = []
        for ent in blamed:
            ent["strikes"] += 1
            ent["distrust"] = min(self.DISTRUST_REFRESHES << (ent["strikes"] - 1), 1 << 16)
        if blamed and not self._warned:
            self._warned = True
            warnings.warn("differender_amd: the hint DR_HINT_NO_EARLY_TERMINATION (derived from the transfer function's largest "
                          "alpha) was wrong for a recent render -- the TF was written to without torch's version counter seeing "
                          "it (`tf.data`, a raw pointer). The device repaired the render (every ray marched one by one: slow); "
                          "the hint is withheld from that TF until its largest alpha has been re-read a few times.",
                          RuntimeWarning, stacklevel=4)
        return len(blamed)

    @staticmethod
    def _key(tf):
        return (tf.untyped_storage().data_ptr(), tf.storage_offset(), tuple(tf.shape), tuple(tf.stride()), tf.dtype)

    def invalidate(self, tf):
        base = tf.untyped_storage().data_ptr()
        for k in [k for k in self._seen if k[0] == base]:
            del self._seen[k]

    def _start_read(self, ent, tf, alpha):
        host = ent["host"]   # one pinned scalar per entry (at most one read of an entry is in flight)
        if host is None:
            host = ent["host"] = torch.empty((), dtype=torch.float32, pin_memory=True)
        with torch.no_grad(), torch.cuda.device(tf.device):   # (the event must sit on the stream the copy runs on)
            host.copy_(torch.nan_to_num(alpha(tf.detach()).float(), nan=float("inf")).max(), non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
        ent["pending"] = (host, ev, tf._version)
        ent["since"] = 0

    def amax(self, tf, alpha=lambda t: t[..., 3]):
        """(value, exact): the largest alpha of `tf` (inf if any alpha is NaN) as last read, and whether that reading is of
        the tensor's CURRENT version; (None, False) while nothing is known. Never waits for the device."""
        key = self._key(tf)
        ent = self._seen.get(key)
        if ent is None:
            if len(self._seen) >= self._capacity:
                self._seen.pop(next(iter(self._seen)))
            self._seen[key] = {"tensor": tf.detach(), "seen": tf._version, "value": None, "value_version": None,
                               "pending": None, "since": 0, "host": None, "distrust": 0, "strikes": 0}
            # A LEAF that requires grad is a parameter somebody optimises: it WILL be seen again (unlike a temporary -- and unlike
            # the result of a differentiable op, e.g. torch.sigmoid(logits): a new tensor every iteration that also "requires
            # grad"), so its largest alpha is read at once -- the harmless "many rays terminate" hint is then there from the second
            # or third iteration of a training loop instead of the ninth (OPT.py's TF: forward 1.81 -> 1.63 ms per iteration).
            if tf.requires_grad and tf.is_leaf:
                self._start_read(self._seen[key], tf, alpha)
            return None, False
        if ent["pending"] is not None and ent["pending"][1].query():
            host, _, ver = ent["pending"]
            ent["value"], ent["value_version"], ent["pending"] = float(host.item()), ver, None
            ent["distrust"] = max(ent["distrust"] - 1, 0)   # one more re-reading since the last wrong hint
        v = tf._version
        ent["since"] += 1
        if ent["value_version"] == v:
            # (re-read now and then all the same: `tf.data` writes do not bump the version)
            if ent["pending"] is None and ent["since"] >= self.REFRESH_EVERY:
                self._start_read(ent, tf, alpha)
            return ent["value"], ent["distrust"] == 0
        if ent["pending"] is None:
            # an unchanged tensor is read on the second sighting of its version; one that is written to between calls on
            # every REFRESH_EVERY-th call (its last reading keeps serving the harmless hint meanwhile)
            if ent["seen"] == v or ent["since"] >= self.REFRESH_EVERY:
                self._start_read(ent, tf, alpha)
        ent["seen"] = v
        return ent["value"], False

    def hints(self, tf, vol_shape, sampling_rate, max_samples, mode, alpha=lambda t: t[..., 3]):
        a, exact = self.amax(tf, alpha)
        if a is None:
            return 0
        # may_terminate() of the device (csrc/dr_brick_common.h), in double, with a margin on either side of its 0.02
        VX, VY, VZ = (int(v) for v in vol_shape)
        diag = math.sqrt((VX - 1) ** 2 + (VY - 1) ** 2 + (VZ - 1) ** 2)
        n_max = math.floor(float(sampling_rate) * 2.0 * math.sqrt(3.0) * diag) + 1.0
        if mode == N.DR_MODE_DIFF and n_max > max_samples:
            n_max = float(max_samples)
        n_max = max(n_max, 1.0)
        if not (a == a) or a == float("inf"):
            return 0
        base = max(1.0 - min(a, 1.0), 0.0)
        op = 1.0 - base ** (1.0 / float(sampling_rate))
        remain = (1.0 - op) ** n_max if op < 1.0 else 0.0
        # "no ray terminates" only from a reading of the CURRENT version (a wrong one costs a slow repair on the device);
        # "many rays terminate" also from the last reading of a tensor that has been written to since (OPT.py clamps its TF
        # in place every iteration): that hint only chooses how finely the pre-pass proceeds
        if exact and remain > 0.03:   # (the device's threshold is 0.02, evaluated in float: 1.5 x is a wide margin for rounding)
            key = self._key(tf)
            if not self._handed or self._handed[-1] != key:
                self._handed = [k for k in self._handed if k != key][-3:] + [key]
            return N.DR_HINT_NO_EARLY_TERMINATION
        if remain < 1e-4:
            return N.DR_HINT_EARLY_TERMINATION
        return 0


_hints = _TerminationHints()


def termination_hints(tf, vol_shape, sampling_rate, max_samples, mode, alpha=lambda t: t[..., 3]):
    """DR_HINT_* bits for march_fwd(hints=...) from a TF tensor the CALLER holds on to (see _TerminationHints); `alpha`
    selects its alpha entries when the layout is not (..., R, 4) -- e.g. `lambda t: t[..., 3, :]` for the user-facing
    ([BS,] 4, R) layout of Raycaster."""
    return _hints.hints(tf, vol_shape, sampling_rate, max_samples, mode, alpha)


def march_fwd(vol, tf, cam, entry, exit_, rays, n, max_samples, sampling_rate, mode=N.DR_MODE_DIFF,
              variant=N.DR_VARIANT_AUTO, want_steps=True, fov_deg=30.0, near=0.1, workspace="auto", rows=None,
              hints="auto", tape=False):
    """raycast + get_final_image (VR.py:261-306,363-372) or the nondiff pair (VR.py:308-361).
    Returns out (views,W,H,4) and steps (views,W,H) int32 (or None).
    workspace: a buffer from alloc_workspace() (keep it for march_bwd), "auto" to allocate a throw-away one,
    or None to force the baseline kernels.
    hints: "auto" (DR_HINT_* from the TF's largest alpha once it is known, see _TerminationHints), 0 / None, or explicit bits.
    tape: DR_TAPE_TF -- the caller will ask for the TF gradient only (march_bwd(want_vol=False, tape=True)); the workspace must come
    from alloc_workspace(..., tape=(max_samples, sampling_rate)). Differentiable mode only; same image."""
    _require_gpu(vol, "volume")
    V, W, H = n.shape
    dev = vol.device
    cam = cam.to(torch.float32).contiguous()
    out = torch.empty((V, W, H, 4), dtype=torch.float32, device=dev)
    steps = torch.empty((V, W, H), dtype=torch.int32, device=dev) if want_steps else None
    vargs = _vol_args(vol, V)
    targs = _tf_args(tf, V)
    if isinstance(workspace, str):
        workspace = alloc_workspace(V, (W, H), vargs[2:5], targs[1], dev) if variant == N.DR_VARIANT_AUTO else None
    if isinstance(hints, str):
        hints = _hints.hints(tf, vargs[2:5], sampling_rate, max_samples, mode) if variant == N.DR_VARIANT_AUTO else 0
    with torch.cuda.device(dev):
        rc = N.lib().dr_march_fwd_rows(*vargs, targs[0], targs[1], targs[2], cam.data_ptr(), entry.data_ptr(),
                                       exit_.data_ptr(), rays.data_ptr(), n.data_ptr(), V, W, H, int(max_samples),
                                       float(sampling_rate), float(np.radians(fov_deg)), float(near), int(mode),
                                       int(variant) | int(hints or 0) | (N.DR_TAPE_TF if tape else 0), out.data_ptr(),
                                       steps.data_ptr() if want_steps else None,
                                       *_ws_args(workspace), *_rows(rows, W), _stream())
    N.check(rc, "dr_march_fwd_rows")
    return out, steps


def march_bwd(vol, tf, cam, entry, exit_, rays, n, max_samples, sampling_rate, grad_out, out, want_vol=True,
              want_tf=True, variant=N.DR_VARIANT_AUTO, fov_deg=30.0, near=0.1, workspace=None, rows=None, count_evaluated=False,
              tape=False):
    """Adjoint of the differentiable march w.r.t. vol and tf (replaces raycast.grad, VR.py:460-461,470-471).
    Shared (un-batched) vol / tf receive one gradient accumulated over all views.
    workspace: the buffer the matching march_fwd filled (fast path); None runs the baseline kernels.
    count_evaluated: measurement only (DR_COUNT_EVALUATED; see evaluated_samples()).
    tape: the forward was run with tape=True and only d_tf is wanted: the per-ray pass over the tape (csrc/tf_tape.hip)."""
    if tape and want_vol:
        raise ValueError("tape=True serves the backward w.r.t. the transfer function alone (want_vol=False)")
    _require_gpu(vol, "volume")
    V, W, H = n.shape
    cam = cam.to(torch.float32).contiguous()
    grad_out = grad_out.to(torch.float32).contiguous()
    out = out.contiguous()
    vargs = _vol_args(vol, V)
    targs = _tf_args(tf, V)
    d_vol = d_tf = None
    dv = (None, 0, 0, 0, 0)
    if want_vol:
        d_vol = torch.zeros_like(vol, dtype=torch.float32, memory_format=torch.preserve_format)
        if vol.ndim == 3:
            dv = (d_vol.data_ptr(), *d_vol.stride(), 0)
        else:
            dv = (d_vol.data_ptr(), *d_vol.stride()[1:], d_vol.stride(0))
    dt = (None, 0)
    if want_tf:
        d_tf = torch.zeros_like(tf)
        dt = (d_tf.data_ptr(), d_tf.stride(0) if tf.ndim == 3 else 0)
    if not (want_vol or want_tf):
        return None, None
    with torch.cuda.device(vol.device):
        rc = N.lib().dr_march_bwd_rows(*vargs, targs[0], targs[1], targs[2], cam.data_ptr(), entry.data_ptr(),
                                       exit_.data_ptr(), rays.data_ptr(), n.data_ptr(), V, W, H, int(max_samples),
                                       float(sampling_rate), float(np.radians(fov_deg)), float(near),
                                       int(variant) | (N.DR_COUNT_EVALUATED if count_evaluated else 0) | (N.DR_TAPE_TF if tape else 0),
                                       grad_out.data_ptr(), out.data_ptr(), *dv, *dt, *_ws_args(workspace),
                                       *_rows(rows, W), _stream())
    N.check(rc, "dr_march_bwd_rows")
    return d_vol, d_tf


def mse_loss_grad(out, reference, inv_norm=None, want_grad=True, loss=None):
    """Image loss and its gradient in one pass over the render (dr_mse_loss_grad; replaces compute_loss,
    EX.py:368-373, and the torch mse_loss round trip of EX.py:439-443).
    Returns (loss, grad_out): loss is a 0-d float64 tensor on the device (accumulated into `loss` if given),
    grad_out has the shape of `out`. inv_norm defaults to 1/numel (= torch.nn.functional.mse_loss)."""
    _require_gpu(out, "out")
    if reference.shape != out.shape or reference.device != out.device:
        raise ValueError("reference must match out in shape and device")
    if out.dtype != torch.float32 or reference.dtype != torch.float32:
        raise TypeError("out and reference must be float32")
    out = out.contiguous()
    reference = reference.contiguous()
    if loss is None:
        loss = torch.zeros((), dtype=torch.float64, device=out.device)
    grad = torch.empty_like(out) if want_grad else None
    inv_norm = 1.0 / out.numel() if inv_norm is None else float(inv_norm)
    with torch.cuda.device(out.device):
        rc = N.lib().dr_mse_loss_grad(out.data_ptr(), reference.data_ptr(), out.numel(), inv_norm,
                                      grad.data_ptr() if want_grad else None, loss.data_ptr(), _stream())
    N.check(rc, "dr_mse_loss_grad")
    return loss, grad


def tf_momentum_step(tf, d_tf, momentum, lr, gamma, max_grad):
    """In-place momentum step on the transfer function (dr_tf_momentum_step; apply_grad, EX.py:375-381):
    momentum = gamma*momentum + lr*clamp(d_tf, +-max_grad); tf = max(tf - momentum, 0)."""
    _require_gpu(tf, "tf")
    for t, name in ((tf, "tf"), (d_tf, "d_tf"), (momentum, "momentum")):
        if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != tf.shape or t.device != tf.device:
            raise ValueError(f"{name} must be a contiguous float32 tensor of tf's shape on tf's device")
    with torch.cuda.device(tf.device):
        rc = N.lib().dr_tf_momentum_step(tf.data_ptr(), d_tf.data_ptr(), momentum.data_ptr(), tf.numel(), float(lr),
                                         float(gamma), float(max_grad), _stream())
    N.check(rc, "dr_tf_momentum_step")
    _hints.invalidate(tf)   # written through its raw pointer: torch's version counter did not see it
    return tf, momentum
