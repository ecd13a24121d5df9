"""Which TF resolutions does the fast backward (LDS = 57 104 + 48 R bytes; its work-item kernel 8 208 more) really launch at?
Prints the device's LDS limits and tries a small fwd + bwd at each R (GPU box)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from differender_amd import functional as F
from oracle import oracle as O

p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if "shared" in k.lower()})
dev = torch.device("cuda:0")
vol = torch.from_numpy(O.synth_volume(24)).to(dev)
cam = torch.tensor([O.in_circles(0.4)], dtype=torch.float32, device=dev)
WH = (16, 16)
e, x, r, n = F.ray_setup(cam, WH, vol.shape, 1.0)
for R in [int(a) for a in sys.argv[1:]] or [256, 1024, 1536, 1800, 1900, 2000, 2040, 2048, 2200, 2223]:
    tf = torch.from_numpy(O.bench_tf(R, 0.03)).to(dev)
    ws = F.alloc_workspace(1, WH, vol.shape, R, dev)
    try:
        out, _ = F.march_fwd(vol, tf, cam, e, x, r, n, 4096, 1.0, workspace=ws)
        g = torch.ones_like(out)
        dv, dt = F.march_bwd(vol, tf, cam, e, x, r, n, 4096, 1.0, g, out, workspace=ws)
        torch.cuda.synchronize()
        print(R, "workspace" if ws is not None else "no workspace", "ok", float(dt.abs().sum()))
    except RuntimeError as ex:
        print(R, "workspace" if ws is not None else "no workspace", "FAILED:", ex)
