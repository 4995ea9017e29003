"""Ablation of the weighted DR solve (C3): which phase of the weighted chunk kernel costs what.  PROXTV_ABLATE bits:
1 = skip the walk, 2 = skip the epilogue, 4 = skip the window loads (8 = baseline without repair launches)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(0)
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
X = dev(rng.standard_normal((4096, 4096)))
W1, W2 = dev(rng.uniform(0.05, 0.15, (4095, 4096))), dev(rng.uniform(0.05, 0.15, (4096, 4095)))
out = device.colmajor_empty((4096, 4096))
for ab in (0, 8, 9, 12, 13, 10):
    lib.proxtv_set_option(b"ablate", ab)
    lib.proxtv_set_option(b"profile", 1)
    device.tv1w_2d(X, W1, W2, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter(); device.tv1w_2d(X, W1, W2, out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print(f"ablate {ab:2d}: {dt*1e3:7.2f} ms  col/row = {lib.proxtv_last_kernel_ms(0):.2f}/{lib.proxtv_last_kernel_ms(1):.2f}", flush=True)
