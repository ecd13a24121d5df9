"""Which term of the D4 bound flags rays? (library built with -DDR_D4_DEBUG: the image holds, per ray, the linear part, the quadrature
part, the total and the index of the channel with the largest bound)   python tools/d4_terms_probe.py N WH SR TF MODE"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from differender_amd import functional as Fn
from differender_amd.utils import get_tf
N, wh, sr, tfname, mode = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), sys.argv[4], int(sys.argv[5])
dev = torch.device("cuda:0"); R = 128
vol = bench.synth_volume_torch(N, dev)
cam = torch.tensor([bench.in_circles(2.1)], dtype=torch.float32, device=dev)
tf = get_tf("tf1", R).t().contiguous().to(dev) if tfname != "bench" else bench.bench_tf_torch(R, 1e-3, dev)
e, x, r, n = Fn.ray_setup(cam, (wh, wh), vol.shape, sr)
ws = Fn.alloc_workspace(1, (wh, wh), vol.shape, R, dev)
ws[64:68].view(torch.int32)[0] = -1
o, st = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws, hints=0)
o = o[0].cpu().numpy(); st = st[0].cpu().numpy()
fl = o[..., 2] > 3e-6
print(f"{N} {wh} sr {sr} {tfname} mode {mode}: flagged {fl.sum()} rays; by channel {np.bincount(o[..., 3][fl].astype(int), minlength=4)}")
if fl.any():
    lin, qd = o[..., 0][fl], o[..., 1][fl]
    print(f"  linear part: median {np.median(lin):.2e} max {lin.max():.2e}; quadrature part: median {np.median(qd):.2e} max {qd.max():.2e}")
    print(f"  flagged by the linear part alone {(lin > 3e-6).sum()}, by the quadrature part alone {(qd > 3e-6).sum()}; steps of flagged rays: median {np.median(st[fl])}")

    worst = int(np.argmax(o[..., 2]))
    print("  worst ray: pixel", np.unravel_index(worst, o.shape[:2]), "bound", o[..., 2].ravel()[worst], "steps", st.ravel()[worst])
    ws[64:68].view(torch.int32)[0] = worst
    torch.cuda.synchronize()
    Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws, hints=0)
    torch.cuda.synchronize()
