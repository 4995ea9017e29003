"""Small images: wall time per DR solve (device-resident) -- where a solve is a chain of 72 dependent sweeps, each a handful of
workgroups.   python tools/small_images.py [n ...]      (run under rocprofv3 --kernel-trace --stats for the kernel durations)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
for n in [int(a) for a in sys.argv[1:]] or [128, 256, 512, 1024, 2048]:
    X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((n, n))).cuda())
    out = device.colmajor_empty((n, n))
    for _ in range(4):
        device.tv1_2d(X, 0.1, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20):
        device.tv1_2d(X, 0.1, out=out)
    torch.cuda.synchronize()
    print(f"DR {n}^2 lambda=0.1: {(time.perf_counter() - t0) / 20 * 1e3:.3f} ms per solve   mode {lib.proxtv_chunk_mode()}", flush=True)
