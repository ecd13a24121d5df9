"""Host-side cost of each call of one optimisation step (no synchronisation inside the loop)."""
import sys, time, math, os
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from bench import synth_volume_torch, bench_tf_torch, in_circles

dev = torch.device("cuda:0")
N, IMG, R = 512, 512, 256
vol = synth_volume_torch(N, dev); tf = bench_tf_torch(R, 1e-3, dev)
ws = F.alloc_workspace(1, (IMG, IMG), (N, N, N), R, dev)
cam = torch.tensor([in_circles(0.3)], dtype=torch.float32, device=dev)
target = torch.rand((1, IMG, IMG, 4), device=dev)
acc = {}
def T(name, fn):
    t = time.perf_counter(); r = fn(); acc[name] = acc.get(name, 0.0) + time.perf_counter() - t; return r
for it in range(12):
    if it == 2:
        torch.cuda.synchronize(); acc.clear(); t_all = time.perf_counter()
    e, x, r, n = T("ray_setup", lambda: F.ray_setup(cam, (IMG, IMG), (N, N, N), 1.0))
    out, steps = T("march_fwd", lambda: F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws))
    _, g = T("loss", lambda: F.mse_loss_grad(out, target))
    dv, dt = T("march_bwd", lambda: F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g, out, workspace=ws))
host = time.perf_counter() - t_all
torch.cuda.synchronize()
total = time.perf_counter() - t_all
print("host enqueue per step: %.3f ms; wall per step %.3f ms" % (host / 10 * 1e3, total / 10 * 1e3))
for k, v in acc.items(): print("  %-10s %.3f ms" % (k, v / 10 * 1e3))
