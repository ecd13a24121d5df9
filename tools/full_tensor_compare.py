import sys, os, torch
sys.path.insert(0, os.getcwd())
from differender_amd import functional as F
from differender_amd.utils import get_tf
from bench import synth_volume_torch, bench_tf_torch, in_circles
dev = torch.device("cuda:0")
# usage: full_tensor_compare.py [N=512] [IMG=512] [f32|f16] [jitter seed=0]
N = int(sys.argv[1]) if len(sys.argv) > 1 else 512
IMG = int(sys.argv[2]) if len(sys.argv) > 2 else 512
F16 = len(sys.argv) > 3 and sys.argv[3] == "f16"
JIT = int(sys.argv[4]) if len(sys.argv) > 4 else 0
R = 256
vol = synth_volume_torch(N, dev)
if F16:
    vol = vol.half()
for tfname in ("bench", "tf1"):
    tf = bench_tf_torch(R, 1e-3, dev) if tfname == "bench" else get_tf("tf1", R).t().contiguous().to(dev)
    for ci in (0.3, 1.7):
        cam = torch.tensor([in_circles(ci)], dtype=torch.float32, device=dev)
        ws = F.alloc_workspace(1, (IMG, IMG), (N,) * 3, R, dev)
        e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, 1.0, jitter_seed=JIT)
        out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, workspace=ws)
        ob, sb = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, variant=1)
        g = torch.randn(out.shape, generator=torch.Generator().manual_seed(5)).to(dev)
        dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g, out, workspace=ws)
        db, dtb = F.march_bwd(vol, tf, cam, e, x, r, n, 1 << 20, 1.0, g, ob, variant=1)
        d = (dv - db).abs()
        print(tfname, ci, "steps equal", bool(torch.equal(steps, sb)), "fwd max diff %.2e" % float((out - ob).abs().max()),
              "d_vol max diff / max %.3e" % (float(d.max()) / float(db.abs().max())), "voxels beyond 1e-4 of max:", int((d > 1e-4 * db.abs().max()).sum()),
              "beyond 3e-5:", int((d > 3e-5 * db.abs().max()).sum()), "d_tf %.3e" % (float((dt - dtb).abs().max()) / float(dtb.abs().max())), flush=True)
