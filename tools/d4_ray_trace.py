"""Where along ONE ray do the fast composite and the sequential kernels part company? Renders the view with max_samples = k for a
ladder of k (the differentiable march stops after min(n, S) samples, VR.py:268) and prints both prefixes of pixel (i, j).
    python tools/d4_ray_trace.py N WH SR TF i j [step]      (run under DIFFERENDER_HIP_LIB=ab_libs/d4off.so: no exact pass)"""
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
import bench  # noqa: E402
from differender_amd import functional as Fn  # noqa: E402
from differender_amd.utils import get_tf  # noqa: E402

N, wh, sr, tfname, pi, pj = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4], int(sys.argv[5]), int(sys.argv[6])
step = int(sys.argv[7]) if len(sys.argv) > 7 else 32
dev = torch.device("cuda:0")
R = 128
vol = bench.synth_volume_torch(N, dev)
cam = torch.tensor([bench.in_circles(2.1)], dtype=torch.float32, device=dev)
tf = get_tf("tf1", R).t().contiguous().to(dev)
if tfname == "d4":
    tf[:, 3] = torch.where(tf[:, 3] == 0, torch.full_like(tf[:, 3], 1e-6), tf[:, 3])
e, x, r, n = Fn.ray_setup(cam, (wh, wh), vol.shape, sr)
full, st = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, hints=0)
nst = int(st[0, pi, pj])
print("planned", int(n[0, pi, pj]), "live", nst)
prev = None
for k in list(range(step, nst, step)) + [nst]:
    f, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, k, sr, hints=0)
    s, _ = Fn.march_fwd(vol, tf, cam, e, x, r, n, k, sr, variant=1)
    fv, sv = f[0, pi, pj].cpu().numpy(), s[0, pi, pj].cpu().numpy()
    d = fv - sv
    print(k, "fast", fv, "seq", sv, "diff", d, "step of diff", (d - prev) if prev is not None else "")
    prev = d
