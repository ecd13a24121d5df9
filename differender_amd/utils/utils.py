"""Helpers the example scripts import from `differender.utils` (reference: differender/utils/utils.py,
"UT.py"): transfer-function presets and camera paths. The reference builds its presets with
torchvtk.utils.tex_from_pts (an undeclared dependency that is not installed here); `tex_from_pts`
below restates its published behaviour: piecewise-linear interpolation of (x, r, g, b, a) control
points onto a regular grid over [0,1], returned channels-first (4, res)."""
import math

import torch
import torch.nn.functional as F

__all__ = ["get_tf", "in_circles", "get_rand_pos", "tex_from_pts", "TFGenerator"]


def tex_from_pts(pts, res):
    """pts (N,5) rows (x, r, g, b, a) with x ascending in [0,1] -> (4, res) float32 texture."""
    pts = torch.as_tensor(pts, dtype=torch.float32)
    xs = pts[:, 0].contiguous()
    q = torch.linspace(0.0, 1.0, res)
    hi = torch.searchsorted(xs, q, right=True).clamp(1, len(xs) - 1)
    lo = hi - 1
    x0, x1 = xs[lo], xs[hi]
    w = ((q - x0) / (x1 - x0).clamp_min(1e-12)).clamp(0.0, 1.0)
    return (pts[lo, 1:] * (1.0 - w)[:, None] + pts[hi, 1:] * w[:, None]).t().contiguous()


# Control points (x, r, g, b, a) of the reference's presets (data, UT.py:9-66).
_PRESETS = {
    "tf1": [[0.0000, 0.0000, 0.0000, 0.0000, 0.0000], [0.0840, 0.8510, 0.7230, 0.4672, 0.0000],
            [0.0850, 0.8510, 0.7230, 0.4672, 0.0831], [0.1844, 0.8510, 0.7230, 0.4672, 0.0801],
            [0.1890, 0.8510, 0.7230, 0.4672, 0.0000], [0.2444, 0.8667, 0.5166, 0.6566, 0.0000],
            [0.2528, 0.7176, 0.0675, 0.3276, 0.0782], [0.2621, 0.8667, 0.5166, 0.6566, 0.0000],
            [0.3407, 0.9843, 0.9843, 0.9843, 0.0000], [0.3601, 0.9843, 0.9843, 0.9843, 0.3904],
            [0.4475, 0.9843, 0.9843, 0.9843, 0.3917], [0.4655, 0.9843, 0.9843, 0.9843, 0.0000],
            [1.0000, 0.0000, 0.0000, 0.0000, 0.0000]],
    "tf2": [[0.0000, 0.0000, 0.0000, 0.0000, 0.0000], [0.0178, 0.5333, 0.3597, 0.1861, 0.0000],
            [0.0206, 0.5333, 0.3597, 0.1861, 0.1834], [0.0361, 0.5333, 0.3597, 0.1861, 0.1804],
            [0.0388, 0.5333, 0.3597, 0.1861, 0.0000], [0.2224, 0.6902, 0.0839, 0.1951, 0.0000],
            [0.2274, 0.6902, 0.0839, 0.1951, 0.0880], [0.2479, 0.6902, 0.0839, 0.1951, 0.0831],
            [0.2515, 0.6902, 0.0839, 0.1951, 0.0000], [0.2857, 0.9843, 0.9843, 0.9843, 0.0000],
            [0.3042, 0.9843, 0.9843, 0.9843, 0.8240], [0.4540, 0.9843, 0.9843, 0.9843, 0.8172],
            [0.4916, 0.9843, 0.9843, 0.9843, 0.0000], [1.0000, 0.0000, 0.0000, 0.0000, 0.0000]],
    "tf3": [[0.0000, 0.0000, 0.0000, 0.0000, 0.0000], [0.0279, 0.5991, 0.6235, 0.1345, 0.0000],
            [0.0477, 0.5991, 0.6235, 0.1345, 0.1736], [0.1090, 0.5991, 0.6235, 0.1345, 0.1779],
            [0.1304, 0.5991, 0.6235, 0.1345, 0.0000], [0.3654, 0.9843, 0.9843, 0.9843, 0.0000],
            [0.3991, 0.9843, 0.9843, 0.9843, 0.3912], [0.7440, 0.9843, 0.9843, 0.9843, 0.3893],
            [0.7850, 0.9843, 0.9843, 0.9843, 0.0000], [1.0000, 0.0000, 0.0000, 0.0000, 0.0000]],
    "tf4": [[0.0000, 0.0000, 0.0000, 0.0000, 0.0000], [0.0916, 0.5059, 0.1627, 0.1627, 0.0000],
            [0.1204, 0.5059, 0.1627, 0.1627, 0.1932], [0.1865, 0.5059, 0.1627, 0.1627, 0.1956],
            [0.2120, 0.5059, 0.1627, 0.1627, 0.0000], [0.4841, 0.9176, 0.9176, 0.9176, 0.0000],
            [0.5195, 0.9176, 0.9176, 0.9176, 0.6406], [0.6609, 0.9176, 0.9176, 0.9176, 0.6362],
            [0.6968, 0.9176, 0.9176, 0.9176, 0.0000], [1.0000, 0.0000, 0.0000, 0.0000, 0.0000]],
    "tf5": [[0.0000, 0.0000, 0.0000, 0.0000, 0.0000], [0.1300, 0.5000, 0.5000, 0.5000, 0.0000],
            [0.1350, 0.5000, 0.5000, 0.5000, 0.7500], [0.1600, 0.5000, 0.5000, 0.5000, 0.7500],
            [0.1700, 0.5000, 0.5000, 0.5000, 0.0000], [1.0000, 0.0000, 0.0000, 0.0000, 0.0000]],
}


class TFGenerator:
    """Stand-in for torchvtk.utils.TFGenerator as UT.py:67-70 uses it (`TFGenerator(peakgen_kwargs={'max_num_peaks': 2}).generate()`
    -> control points for tex_from_pts): torchvtk is neither vendored by the reference nor installed here, so only the call's
    CONTRACT is restated -- random control points (x, r, g, b, a) of 1 .. max_num_peaks trapezoid opacity peaks in the presets'
    own format (x ascending in [0, 1], transparent between the peaks, one random colour per peak) -- not its random stream or its
    parameter distributions [mem: unpinned; a random TF has no reference value to match]. Drawn from torch's global generator."""

    def __init__(self, peakgen_kwargs=None, **_ignored):
        kw = dict(peakgen_kwargs or {})
        self.max_num_peaks = int(kw.get("max_num_peaks", 5))
        self.height_range = tuple(kw.get("height_range", (0.1, 0.9)))
        self.width_range = tuple(kw.get("width_range", (0.02, 0.2)))

    def generate(self):
        n = int(torch.randint(1, self.max_num_peaks + 1, (1,)))
        slot = 1.0 / n                                                   # one peak per equal slice of [0, 1]: ascending x, no overlap
        pts = [[0.0, 0.0, 0.0, 0.0, 0.0]]
        for k in range(n):
            w = float(torch.empty(1).uniform_(*self.width_range).clamp(max=0.8 * slot))
            c = k * slot + 0.1 * slot + float(torch.rand(1)) * (0.8 * slot - w) + 0.5 * w
            ramp = 0.1 * w + 1e-3
            h = float(torch.empty(1).uniform_(*self.height_range))
            r, g, b = (float(v) for v in torch.rand(3))
            x0, x1 = max(c - 0.5 * w, 1e-3), min(c + 0.5 * w, 1.0 - 1e-3)
            pts += [[x0, r, g, b, 0.0], [x0 + ramp, r, g, b, h], [max(x1 - ramp, x0 + ramp), r, g, b, h], [x1, r, g, b, 0.0]]
        pts.append([1.0, 0.0, 0.0, 0.0, 0.0])
        t = torch.tensor(pts, dtype=torch.float32)
        t[:, 0] = torch.cummax(t[:, 0], 0).values                          # (guards x ascending against rounding)
        return t


def get_tf(id, res):
    """UT.py:7-79: (4, res) transfer-function texture for a preset name."""
    if id in _PRESETS:
        return tex_from_pts(torch.tensor(_PRESETS[id]), res)
    if id == "black":
        return torch.zeros((4, res)) + 1e-2
    if id == "gray":
        tf = torch.full((4, res), 0.5)
        tf[3, :] = 0.02
        return tf
    if id == "rand":
        return torch.rand(4, res)
    if id == "generate":   # UT.py:67-70
        tfgen = TFGenerator(peakgen_kwargs={"max_num_peaks": 2})
        return tex_from_pts(tfgen.generate(), res)
    raise Exception(f"Invalid Transfer function identifier given ({id}).")


def in_circles(i, y=0.7, dist=2.5):
    """UT.py:80-83: orbit camera."""
    return torch.tensor([math.cos(i) * dist, y, math.sin(i) * dist], dtype=torch.float32)


def get_rand_pos(bs=None, dist=2.7):
    """UT.py:86-90: random camera(s) on a sphere of radius `dist`."""
    if bs is None:
        return F.normalize(torch.randn(3), dim=0) * dist
    return F.normalize(torch.randn(bs, 3), dim=1) * dist
