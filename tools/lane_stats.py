"""Lane utilisation of the forward brick kernel (library built with -DDR_LANE_STATS)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from bench import synth_volume_torch, bench_tf_torch, in_circles
dev = torch.device("cuda:0")
for N, IMG in ((512, 512), (256, 256)):
    vol = synth_volume_torch(N, dev); tf = bench_tf_torch(256, 1e-3, dev)
    ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, 256, dev)
    cam = torch.tensor([in_circles(0.3)], dtype=torch.float32, device=dev)
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, 1.0)
    ws[:2048].zero_()
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
    torch.cuda.synchronize()
    st = ws[:2048].view(torch.int64).cpu()
    slots, listed, valid = int(st[24]), int(st[25]), int(st[26])
    print(f"{N}^3/{IMG}^2: samples {int(steps.sum())}  lane slots {slots}  listed {listed} ({100*listed/slots:.1f} %)  in-brick {valid} ({100*valid/slots:.1f} %)")
