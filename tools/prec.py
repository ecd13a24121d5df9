import sys, os
import torch, numpy as np
sys.path.insert(0, "/root/repo")
from differender_amd import functional as F
from oracle import oracle as O
from bench import synth_volume_torch, bench_tf_torch, in_circles
dev = torch.device("cuda:0")
N, IMG, R = 256, 256, 256
vol = synth_volume_torch(N, dev)
vol_h = vol.cpu().numpy(); vol_d = vol_h.astype(np.float64)
cam = torch.tensor([in_circles(0.3)], dtype=torch.float32, device=dev)
cam_h = cam[0].cpu().numpy()
ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, R, dev)
for sr in (1.0, 4.0, 8.0, 16.0):
    tf = bench_tf_torch(R, 2e-3, dev); tf_h = tf.cpu().numpy()
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, sr)
    out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, 0, workspace=ws)
    outb, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, 0, variant=1)
    sl = (slice(112, 144), slice(112, 144))
    eh, xh, rh, nh = (t[0].cpu().numpy()[sl] for t in (e, x, r, n))
    # float64 evaluation of the SAME rays (ray buffers promoted): the exact-arithmetic reference
    ref64, _ = O.march_fwd(vol_d, tf_h.astype(np.float64), cam_h.astype(np.float64), eh.astype(np.float64), xh.astype(np.float64), rh.astype(np.float64), nh, 1 << 20, sr, 0)
    a = out[0].cpu().numpy()[sl]; b = outb[0].cpu().numpy()[sl]
    print("sr %4.1f  n~%d  |fast - seq| %.2e   |fast - f64| %.2e   |seq - f64| %.2e" % (sr, int(nh.max()), np.abs(a - b).max(), np.abs(a - ref64).max(), np.abs(b - ref64).max()), flush=True)
