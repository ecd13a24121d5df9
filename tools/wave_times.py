"""Sample-loop time of each of the 8 waves of the backward's workgroups (library built with -DDR_PHASE_TIMING=2)."""
import sys, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from bench import synth_volume_torch, bench_tf_torch, in_circles
dev = torch.device("cuda:0")
N, IMG, R = 512, 512, 256
vol = synth_volume_torch(N, dev); tf = bench_tf_torch(R, 1e-3, dev)
ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, R, dev)
cam = torch.tensor([in_circles(0.3)], dtype=torch.float32, device=dev)
e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, 1.0)
out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, torch.ones_like(out), out, workspace=ws)
torch.cuda.synchronize()
t = ws[:256].view(torch.int64).cpu().numpy().astype(float)[8:16]
print("backward sample-loop ticks per wave, summed over the grid (relative to wave 0):", [round(v / t[0], 3) for v in t])
