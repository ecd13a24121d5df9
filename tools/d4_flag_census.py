"""How many rays does F2 send to the exact pass on the bench's own cameras?  python tools/d4_flag_census.py [N WH SR TF]
(under ab_libs/d4dbg.so also: which term of the bound flags them)"""
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
import bench
from differender_amd import functional as Fn
from differender_amd.utils import get_tf
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
wh = int(sys.argv[2]) if len(sys.argv) > 2 else 512
sr = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
tfname = sys.argv[4] if len(sys.argv) > 4 else "bench"
dev = torch.device("cuda:0"); R = 256
vol = bench.synth_volume_torch(N, dev)
n_max = 2.0 * 3.0 * (N - 1)
tf = bench.bench_tf_torch(R, 3.0 / n_max, dev) if tfname == "bench" else get_tf(tfname, R).t().contiguous().to(dev)
dbg = bool(Fn.N.lib().dr_build_flags() & 1)
for k in range(14):
    cam = torch.tensor([bench.in_circles(0.1 * k)], dtype=torch.float32, device=dev)
    e, x, r, n = Fn.ray_setup(cam, (wh, wh), vol.shape, sr)
    ws = Fn.alloc_workspace(1, (wh, wh), vol.shape, R, dev)
    ws[64:68].view(torch.int32)[0] = -1
    o, st = Fn.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, workspace=ws, hints=0)
    line = f"camera {0.1 * k:.1f}: exact rays {int(Fn.workspace_stats(ws)[15])}"
    if dbg:
        o = o[0].cpu().numpy(); reg = st[0].cpu().numpy() > 1
        fl = (o[..., 2] > 3e-6) & reg
        line += f"; bound > 3e-6 on {fl.sum()} rays, by channel {np.bincount(o[..., 3][fl].astype(int), minlength=4)}, linear part alone {(o[..., 0][fl] > 3e-6).sum()}, max bound {o[..., 2][reg].max():.2e}"
    print(line, flush=True)
