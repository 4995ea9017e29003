#!/usr/bin/env python3
"""Run BASELINE configurations a few times in one process (for rocprofv3 --kernel-trace --stats / --pmc):

    python tools/profile_cases.py c3|c4|c4y|kol|dr<lambda>|calib[+more...] [reps]

calib = the 8-byte-per-lane copy of 1 GiB the FETCH_SIZE / WRITE_SIZE counters are calibrated on (tools/kernel_counters.py)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def case(which):
    rng = np.random.default_rng(0)
    if which == "calib":
        n = 128 * 1024 * 1024
        src = torch.rand(n, dtype=torch.float64, device="cuda")
        dst = torch.empty_like(src)
        return lambda: lib.proxtv_calib_copy_dev(src.data_ptr(), dst.data_ptr(), n, None)
    if which == "c3":
        X = dev(rng.standard_normal((4096, 4096)))
        W1, W2 = dev(rng.uniform(0.05, 0.15, (4095, 4096))), dev(rng.uniform(0.05, 0.15, (4096, 4095)))
        out = device.colmajor_empty((4096, 4096))
        return lambda: device.tv1w_2d(X, W1, W2, out=out)
    if which in ("c4", "c4y"):
        V = dev(rng.standard_normal((512, 512, 64)))
        out = device.colmajor_empty((512, 512, 64))
        return (lambda: device.tvgen(V, [0.1, 0.1, 0.05], [1, 2, 3], out=out)) if which == "c4" else \
               (lambda: device.tvgen(V, [0.1, 0.1, 0.1], [1, 2, 3], method="yang", out=out))
    if which.startswith("dr"):      # dr0.3 -> DR 4096^2 at lambda 0.3
        lam = float(which[2:])
        X = dev(rng.standard_normal((4096, 4096)))
        out = device.colmajor_empty((4096, 4096))
        return lambda: device.tv1_2d(X, lam, out=out)
    if which.startswith("prox"):    # prox0 / prox1: plain 1-D prox sweeps along dimension 0 / 1 of a 4096^2 image
        X = dev(rng.standard_normal((4096, 4096)))
        out = device.colmajor_empty((4096, 4096))
        return lambda: device.tv1_fibres(X, 0.1, int(which[4:]), out=out)
    X = dev(rng.standard_normal((4096, 4096)))
    out = device.colmajor_empty((4096, 4096))
    return lambda: device.tv1_2d(X, 0.1, method="kolmogorov", max_iters=50, out=out)


for which in sys.argv[1].split("+"):
    run = case(which)
    for _ in range(reps):
        run()
    torch.cuda.synchronize()
    del run
    torch.cuda.empty_cache()
