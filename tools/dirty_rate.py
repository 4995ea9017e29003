#!/usr/bin/env python3
"""How often does a chunked sweep leave ANYTHING to the repair kernel (option "why": what marked sweeps dirty)?  Per case: solves,
sweeps, and the eight counters summed over the solves.  The number that decides whether repair launches can be deferred to the
end of a solve (option "optimistic")."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
import ctypes as C

lib = _lib.require_device()
lib.proxtv_set_option(b"why", 1)
why = (C.c_uint * 8)()
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
print(f"{'case':34s} {'solves':>6s} {'sweeps':>7s}  ran-off in-wg across late 2nd-chance - - -")
for n, lam, reps in ((4096, 0.1, 30), (4096, 0.2, 10), (4096, 0.3, 10), (2048, 0.1, 30), (1024, 0.1, 50), (512, 0.1, 50), (256, 0.1, 50)):
    X = dev(np.random.default_rng(n).standard_normal((n, n)))
    out = device.colmajor_empty((n, n))
    device.tv1_2d(X, lam, out=out)
    lib.proxtv_debug_why(why)
    s0 = lib.proxtv_debug_counter(b"sweep_launches")
    dirty_solves = 0
    tot = np.zeros(8, dtype=np.int64)
    for _ in range(reps):
        device.tv1_2d(X, lam, out=out)
        lib.proxtv_debug_why(why)
        w = np.array(list(why), dtype=np.int64)
        tot += w
        dirty_solves += int(w[:4].sum() > 0)
    print(f"DR {n}^2 lambda {lam:<4}              {reps:6d} {lib.proxtv_debug_counter(b'sweep_launches') - s0:7d}  {' '.join(str(int(v)) for v in tot)}   solves with any mark: {dirty_solves}")
rng = np.random.default_rng(0)
X = dev(rng.standard_normal((4096, 4096)))
W1, W2 = dev(rng.uniform(0.05, 0.15, (4095, 4096))), dev(rng.uniform(0.05, 0.15, (4096, 4095)))
out = device.colmajor_empty((4096, 4096))
device.tv1w_2d(X, W1, W2, out=out)
lib.proxtv_debug_why(why)
tot = np.zeros(8, dtype=np.int64); dirty_solves = 0
for _ in range(10):
    device.tv1w_2d(X, W1, W2, out=out)
    lib.proxtv_debug_why(why)
    w = np.array(list(why), dtype=np.int64); tot += w; dirty_solves += int(w[:4].sum() > 0)
print(f"weighted DR 4096^2                     10          {' '.join(str(int(v)) for v in tot)}   solves with any mark: {dirty_solves}")
