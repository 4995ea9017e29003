#!/usr/bin/env python3
"""Where do replayed solves differ from walked ones?  DR on a tall image, replay on / off, by iteration count."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(21)
X = rng.standard_normal((2400, 300))
xd = device.to_colmajor(torch.from_numpy(X).cuda())
lam = 0.1
for iters in (3, 4, 5, 6, 8, 12, 35):
    lib.proxtv_set_option(b"replay", 0)
    a = device.tv1_2d(xd, lam, max_iters=iters)[0].cpu().numpy()
    lib.proxtv_set_option(b"replay", 1)
    b = device.tv1_2d(xd, lam, max_iters=iters)[0].cpu().numpy()
    d = np.abs(a - b)
    bad = np.argwhere(d > 1e-13)
    print(f"iters {iters}: max diff {d.max():.3e}; {len(bad)} elements differ by more than 1e-13")
    if len(bad):
        r, c = np.unravel_index(np.argmax(d), d.shape)
        rows = np.flatnonzero(d[:, c] > 1e-13)
        print(f"   worst at row {r} (segment {r // 1088}, chunk {r % 1088 // 17}, sample {r % 17}), column {c}; rows differing in that column: {rows[:20]} ... {len(rows)}")
        cols = sorted(set(bad[:, 1].tolist()))
        print(f"   columns affected: {len(cols)}; first: {cols[:10]}")
        print("   a:", a[max(0, r - 3):r + 4, c], "\n   b:", b[max(0, r - 3):r + 4, c])
