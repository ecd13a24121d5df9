"""Per-workgroup phase times of the flat kernels (library built with -DDR_PHASE_TIMING=1: tools/mkvariant.sh)."""
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
for it in range(3):
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
    g = torch.ones_like(out)
    dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g, out, workspace=ws)
torch.cuda.synchronize()
t = ws[:256].view(torch.int64).cpu().numpy().astype(float)
nb = (N - 1 + 11) // 12
nb = nb ** 3
print("clock ticks per workgroup (sum over the grid / bricks)")
B = 8   # stats word ST_TIMING = 16 -> 64-bit slot 8
print("fwd: staging+listing %.0f  wave split %.0f  sample loop %.0f" % tuple(t[B:B + 3] / nb))
print("bwd: wave 0 waits for the slowest wave %.0f" % (t[B + 7] / nb))
print("bwd: staging+listing %.0f  wave split %.0f  sample loop %.0f  | whole workgroup incl. flush %.0f" % (t[B + 3] / nb, t[B + 4] / nb, t[B + 5] / nb, t[B + 6] / nb))
print("(library built with -DDR_PHASE_TIMING=3 instead: forward prologue of thread 0 = brick record %.0f, candidates loaded + listed %.0f, box arrived + stored %.0f, barrier %.0f)" % tuple(t[B:B + 4] / nb))
