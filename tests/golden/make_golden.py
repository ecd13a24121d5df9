#!/usr/bin/env python3
"""Generates tests/golden/raycast_16.npz -- SELF-GENERATED golden vectors.

The reference (nanovis/Differender) ships no test vectors and cannot be executed here (taichi is not
installed), so these are produced by this repo's own CPU oracle (oracle/dr_oracle.c), f32 and f64
instantiations, after the oracle has been pinned by the analytic known-answer and finite-difference tests
of tests/test_oracle_kat.py. They guard against regressions of the oracle and give the HIP path a fixed
target that does not depend on the oracle being rebuilt. Run:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O  # noqa: E402


def main():
    N, R, WH, S = (16, 20, 24), 8, (16, 12), 4096
    out = {}
    vol = O.synth_volume(N)
    rng = np.random.RandomState(7)
    tf = rng.uniform(0.05, 0.9, size=(R, 4)).astype(np.float32)
    tf[:, 3] = np.linspace(0.02, 0.25, R)
    cam = O.in_circles(0.8)
    g = rng.randn(*WH, 4).astype(np.float32)
    out.update(vol=vol, tf=tf, cam=cam, grad_out=g, max_samples=np.int32(S))
    for tag, sr, seed in (("sr1", 1.0, 0), ("sr2_jit", 2.0, 77)):
        e, x, r, n = O.ray_setup(cam, *WH, vol.shape, sr=sr, jitter_seed=seed, view=1)
        rgba, steps = O.march_fwd(vol, tf, cam, e, x, r, n, S, sr, 0)
        rgba_nd, steps_nd = O.march_fwd(vol, tf, cam, e, x, r, n, S, sr, 1)
        dv, dt = O.march_bwd(vol, tf, cam, e, x, r, n, S, sr, g)
        # f64 evaluation of the same formulas on the same (f32) ray buffers
        f8 = np.float64
        rgba64, _ = O.march_fwd(vol.astype(f8), tf.astype(f8), cam.astype(f8), e.astype(f8), x.astype(f8),
                                r.astype(f8), n, S, sr, 0)
        dv64, dt64 = O.march_bwd(vol.astype(f8), tf.astype(f8), cam.astype(f8), e.astype(f8), x.astype(f8),
                                 r.astype(f8), n, S, sr, g.astype(f8))
        out.update({f"{tag}_sr": np.float32(sr), f"{tag}_seed": np.uint32(seed), f"{tag}_entry": e, f"{tag}_exit": x,
                    f"{tag}_rays": r, f"{tag}_n": n, f"{tag}_rgba": rgba, f"{tag}_steps": steps,
                    f"{tag}_rgba_nondiff": rgba_nd, f"{tag}_steps_nondiff": steps_nd, f"{tag}_dvol": dv,
                    f"{tag}_dtf": dt, f"{tag}_rgba_f64": rgba64.astype(np.float32),
                    f"{tag}_dvol_f64": dv64.astype(np.float32), f"{tag}_dtf_f64": dt64.astype(np.float32)})
    path = os.path.join(HERE, "raycast_16.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
