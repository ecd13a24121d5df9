"""Time the non-differentiable render (sr given) and the differentiable forward for the loaded library build."""
import sys, os, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F, _native as N
from bench import synth_volume_torch, bench_tf_torch, in_circles
from differender_amd.utils import get_tf
dev = torch.device("cuda:0")
Nv, IMG, R = 512, 512, 256
vol = synth_volume_torch(Nv, dev)
ws = F.alloc_workspace(1, (IMG, IMG), (Nv,) * 3, R, dev)
cam = torch.tensor([in_circles(0.3)], dtype=torch.float32, device=dev)
for tfname in ("bench", "tf1"):
    tf = bench_tf_torch(R, 1e-3, dev) if tfname == "bench" else get_tf("tf1", R).t().contiguous().to(dev)
    for mode, sr in ((0, 1.0), (0, 1.5), (1, 2.0), (1, 3.0), (1, 4.0), (1, 8.0)):
        e, x, r, n = F.ray_setup(cam, (IMG, IMG), (Nv,) * 3, sr)
        for it in range(6):
            if it == 2:
                torch.cuda.synchronize(); t = time.perf_counter()
            out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode=mode, workspace=ws)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t) / 4 * 1e3
        print(os.path.basename(N.LIB_PATH), tfname, "mode", mode, "sr", sr, "%.3f ms" % ms, "planned %.3g" % float(n.sum()))
