"""One long fibre (BASELINE config #1 shape: 10^6 samples) as lambda grows: wall time per call on the device.
    python tools/long_fibre.py [n] [lambdas...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
lams = [float(a) for a in sys.argv[2:]] or [0.5, 3.0, 30.0, 300.0]
x = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((n,))).cuda())
out = device.colmajor_empty((n,))
for lam in lams:
    ts = []
    for _ in range(6):
        torch.cuda.synchronize(); t0 = time.perf_counter(); device.tv1_fibres(x, lam, 0, out=out); torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    print(f"tv1 fibre n={n} lambda={lam:<6} calls 1..6: " + " ".join(f"{t:.2f}" for t in ts) + f" ms   mode {lib.proxtv_chunk_mode()}", flush=True)
