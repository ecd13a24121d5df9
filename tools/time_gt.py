"""Time the reference-style ground-truth render (nondiff, sr=8, BS=8, tf1) and the tf1 forward for the loaded build."""
import time, torch, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender.volume_raycaster import Raycaster
from differender.utils import get_tf, in_circles, get_rand_pos
from examples.render_nondiff_synthetic import synthetic_volume
from differender_amd import _native as N
dev = torch.device("cuda")
Nv, BS, R = 256, 8, 128
vol = synthetic_volume(Nv, dev).float()
tf = get_tf("tf1", R).to(dev).float()
rc = Raycaster(vol.shape[-3:], (256, 256), R, jitter=True, max_samples=1024)
torch.manual_seed(0)
lf = torch.cat([in_circles(0.3)[None], get_rand_pos(BS - 1)], dim=0).float().to(dev)
def t(fn, n=5):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for sr in (8.0, 2.0):
    print(os.path.basename(N.LIB_PATH), "nondiff sr=%g BS=8 256^3: %.2f ms" % (sr, t(lambda: rc.raycast_nondiff(vol, tf, lf, sampling_rate=sr))))
print(os.path.basename(N.LIB_PATH), "diff fwd sr=1 BS=8: %.2f ms" % t(lambda: rc(vol, tf, lf)))
