"""Image losses of the reference's demo loop that are not part of the raycasting path.

`ssim2d` restates `pytorch_msssim.ssim` as examples/test_opt_tf.py:14,70 calls it
(`ssim2d(res, gt, data_range=1.0, size_average=True, nonnegative_ssim=True)`): pytorch_msssim is not installed here and is
not vendored by the reference, so its semantics are taken from memory of the package [mem: parity unpinned] -- 11-tap
gaussian window (sigma 1.5), separable, VALID convolution per channel, K = (0.01, 0.03), the per-channel mean of the SSIM map,
`relu` on it when nonnegative_ssim, mean over channels and batch when size_average. Plain torch ops (differentiable through
autograd); used by the synthetic counterparts of the demo (examples/test_opt_synthetic.py, bench.py --workload opt)."""
import torch
import torch.nn.functional as F

__all__ = ["ssim2d", "dssim_mse_loss"]


def _gauss_window(size, sigma, dtype, device):
    x = torch.arange(size, dtype=dtype, device=device) - size // 2
    g = torch.exp(-(x * x) / (2.0 * sigma * sigma))
    return g / g.sum()


def _filter(x, win):
    c = x.shape[1]
    k = win.numel()
    x = F.conv2d(x, win.view(1, 1, k, 1).expand(c, 1, k, 1), groups=c) if x.shape[2] >= k else x
    x = F.conv2d(x, win.view(1, 1, 1, k).expand(c, 1, 1, k), groups=c) if x.shape[3] >= k else x
    return x


def ssim2d(X, Y, data_range=255.0, size_average=True, win_size=11, win_sigma=1.5, K=(0.01, 0.03), nonnegative_ssim=False):
    """SSIM of two (N, C, H, W) image batches; see the module docstring for what is restated and from where."""
    if X.shape != Y.shape or X.ndim != 4:
        raise ValueError("ssim2d expects two (N, C, H, W) tensors of the same shape")
    win = _gauss_window(win_size, win_sigma, X.dtype, X.device)
    C1, C2 = (K[0] * data_range) ** 2, (K[1] * data_range) ** 2
    c = X.shape[1]
    # the five windowed moments in ONE pair of grouped convolutions (channels stacked): two launches forward, two backward
    m = _filter(torch.cat([X, Y, X * X, Y * Y, X * Y], dim=1), win)
    mu1, mu2 = m[:, :c], m[:, c:2 * c]
    mu1_sq, mu2_sq, mu12 = mu1 * mu1, mu2 * mu2, mu1 * mu2
    s1 = m[:, 2 * c:3 * c] - mu1_sq
    s2 = m[:, 3 * c:4 * c] - mu2_sq
    s12 = m[:, 4 * c:] - mu12
    cs = (2.0 * s12 + C2) / (s1 + s2 + C2)
    ssim_map = ((2.0 * mu12 + C1) / (mu1_sq + mu2_sq + C1)) * cs
    per_channel = ssim_map.flatten(2).mean(-1)
    if nonnegative_ssim:
        per_channel = torch.relu(per_channel)
    return per_channel.mean() if size_average else per_channel.mean(1)


def dssim_mse_loss(res, gt):
    """The loss of examples/test_opt_tf.py:70-72: nan_to_num(1 - ssim(res, gt, data_range=1, nonnegative)) + mse."""
    dssim = 1.0 - ssim2d(res, gt, data_range=1.0, size_average=True, nonnegative_ssim=True)
    mse = F.mse_loss(res, gt)
    return torch.nan_to_num(dssim) + mse, dssim, mse
