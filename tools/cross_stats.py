"""How often ray_cross_kernel needs its exact restart (library built with -DDR_CROSS_STATS: tools/mkvariant.sh)."""
import sys, os, math
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from differender_amd.utils import get_tf, in_circles, get_rand_pos
from bench import synth_volume_torch
dev = torch.device("cuda:0")
for (N, IMG, R, V, sr, mode, noise) in ((256, 256, 128, 8, 8.0, 1, True), (256, 256, 128, 8, 1.0, 0, True), (512, 512, 256, 1, 1.0, 0, False)):
    torch.manual_seed(0)
    vol = synth_volume_torch(N, dev)
    if noise:
        m = torch.rand_like(vol) < 0.05; vol[m] = torch.rand_like(vol[m])
    tf = get_tf("tf1", R).t().contiguous().to(dev)
    cam = torch.cat([in_circles(0.3)[None], get_rand_pos(V - 1)], 0).float().to(dev) if V > 1 else in_circles(0.3)[None].float().to(dev)
    ws = F.alloc_workspace(V, (IMG, IMG), (N,) * 3, R, dev)
    e, x, r, n = F.ray_setup(cam, (IMG, IMG), (N,) * 3, sr)
    ws[:2048].zero_()
    out, steps = F.march_fwd(vol, tf, cam, e, x, r, n, 1 << 20, sr, mode, workspace=ws)
    torch.cuda.synchronize()
    st = ws[:2048].view(torch.int32).cpu()
    st64 = ws[:2048].view(torch.int64).cpu()
    r0, r1 = int(st[32]), int(st[33]); w0, w1 = int(st64[18]), int(st64[19])
    print(f"   exact walk: passes skipped as stagnant {int(st[34])}, passes walked sample by sample {int(st[35])}; clock ticks per ray "
          f"{int(st64[20]) / max(int(st[33]), 1):.0f} (longest {int(st64[21])})")
    print(f"{N}^3 {IMG}^2 x{V} sr={sr} mode={mode}: rays {V*IMG*IMG}, terminated early {int((steps < n).sum())}, resolved by ray_cross {r0} "
          f"(samples walked {w0}, {w0/max(r0,1):.1f}/ray), exact restarts {r1} ({100*r1/max(r0,1):.1f} %, samples walked {w1}, {w1/max(r1,1):.0f}/ray)")
