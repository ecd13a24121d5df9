import sys, os, torch
sys.path.insert(0, os.getcwd())
import bench
from differender_amd import functional as F
from differender_amd.utils import get_tf
dev = torch.device("cuda:0")
N, IMG, R = 512, 512, 256
vol = bench.synth_volume_torch(N, dev)
tf = get_tf("tf1", R).t().contiguous().float().to(dev)
tf[:, 3] = torch.where(tf[:, 3] == 0, torch.full_like(tf[:, 3], 1e-6), tf[:, 3])
cam = torch.tensor([bench.in_circles(0.3)], dtype=torch.float32, device=dev)
for sr in (1.0, 2.0):
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), vol.shape, sr)
    ws = F.alloc_workspace(1, (IMG, IMG), vol.shape, R, dev)
    g = torch.randn(1, IMG, IMG, 4, device=dev)
    for it in range(3):
        t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True); t2 = torch.cuda.Event(enable_timing=True)
        t0.record()
        out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, workspace=ws)
        t1.record()
        dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, g, out, workspace=ws)
        t2.record(); torch.cuda.synchronize()
    print(os.environ.get("DIFFERENDER_HIP_LIB", "shipped")[-12:], "sr", sr, "fwd %.2f ms bwd %.2f ms exact rays %d" % (t0.elapsed_time(t1), t1.elapsed_time(t2), int(F.workspace_stats(ws)[15])))
