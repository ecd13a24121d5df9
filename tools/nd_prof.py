import time, torch, sys
sys.path.insert(0, ".")
from differender.volume_raycaster import Raycaster
from differender.utils import get_tf, in_circles, get_rand_pos
from examples.render_nondiff_synthetic import synthetic_volume
dev = torch.device("cuda")
N, BS, R = 256, 8, 128
vol = synthetic_volume(N, dev).float()
tf = get_tf("tf1", R).to(dev).float()
rc = Raycaster(vol.shape[-3:], (256, 256), R, jitter=True, max_samples=1024)
torch.manual_seed(0)
lf = torch.cat([in_circles(0.3)[None], get_rand_pos(BS - 1)], dim=0).float().to(dev)
for _ in range(4):
    rc.raycast_nondiff(vol, tf, lf, sampling_rate=8.0)
torch.cuda.synchronize()
