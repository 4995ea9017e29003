#!/usr/bin/env python3
"""Run one BASELINE configuration a few times (for rocprofv3 --kernel-trace --stats):  python tools/profile_cases.py c3|c4|c4y|kol"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(0)
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
which = sys.argv[1]
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
if which == "c3":
    X = dev(rng.standard_normal((4096, 4096)))
    W1, W2 = dev(rng.uniform(0.05, 0.15, (4095, 4096))), dev(rng.uniform(0.05, 0.15, (4096, 4095)))
    out = device.colmajor_empty((4096, 4096))
    run = lambda: device.tv1w_2d(X, W1, W2, out=out)
elif which in ("c4", "c4y"):
    V = dev(rng.standard_normal((512, 512, 64)))
    out = device.colmajor_empty((512, 512, 64))
    run = (lambda: device.tvgen(V, [0.1, 0.1, 0.05], [1, 2, 3], out=out)) if which == "c4" else \
          (lambda: device.tvgen(V, [0.1, 0.1, 0.1], [1, 2, 3], method="yang", out=out))
elif which.startswith("dr"):      # dr0.3 -> DR 4096^2 at lambda 0.3
    lam = float(which[2:])
    X = dev(rng.standard_normal((4096, 4096)))
    out = device.colmajor_empty((4096, 4096))
    run = lambda: device.tv1_2d(X, lam, out=out)
else:
    X = dev(rng.standard_normal((4096, 4096)))
    out = device.colmajor_empty((4096, 4096))
    run = lambda: device.tv1_2d(X, 0.1, method="kolmogorov", max_iters=50, out=out)
for _ in range(reps):
    run()
torch.cuda.synchronize()
