import os, sys, math
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from bench import synth_volume_torch, bench_tf_torch
dev = torch.device("cuda:0")
for N, IMG in ((64, 64), (128, 128), (256, 256), (512, 512)):
    vol = synth_volume_torch(N, dev); tf = bench_tf_torch(64, 1e-3, dev)
    cam = torch.tensor([[0.45, 0.2, 0.0]], dtype=torch.float32, device=dev)
    ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, 64, dev)
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, 1.0)
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
    torch.cuda.synchronize()
    st = F.workspace_stats(ws)
    outb, stepsb = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, variant=1)
    print(N, IMG, "repaired", int(st[0]), "fallback", int(st[2]), "items", int(st[5]), "max|diff|", float((out - outb).abs().max()), "steps equal", bool(torch.equal(steps, stepsb)))
