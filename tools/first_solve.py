"""Wall time of the FIRST solves of a fresh process (the geometry policy knows nothing yet) and of the steady state:
    python tools/first_solve.py <lambda> [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
lam = float(sys.argv[1]); n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((n, n))).cuda())
out = device.colmajor_empty((n, n))
small = device.to_colmajor(torch.zeros((64, 64), dtype=torch.float64).cuda())
torch.cuda.synchronize()
ts = []
for k in range(5):
    torch.cuda.synchronize(); t0 = time.perf_counter(); device.tv1_2d(X, lam, out=out); torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"DR {n}^2 lambda={lam}: solves 1..5 took " + " ".join(f"{t:.1f}" for t in ts) + f" ms   mode {lib.proxtv_chunk_mode()}")
