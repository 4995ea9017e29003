"""Weighted DR 4096^2 under one pinned geometry mode (separate process per mode: a faulting kernel aborts the process)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
mode = int(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
rng = np.random.default_rng(0)
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
X = dev(rng.standard_normal((n, n)))
W1, W2 = dev(rng.uniform(0.05, 0.15, (n - 1, n))), dev(rng.uniform(0.05, 0.15, (n, n - 1)))
out = device.colmajor_empty((n, n))
lib.proxtv_set_option(b"chunk_mode", mode)
t0 = time.perf_counter(); device.tv1w_2d(X, W1, W2, out=out); torch.cuda.synchronize()
print(f"mode {mode} n {n}: ok {1e3*(time.perf_counter()-t0):.1f} ms fixups {lib.proxtv_last_fixups()}", flush=True)
