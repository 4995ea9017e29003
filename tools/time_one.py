"""Wall time of ONE configuration, device-resident (for A/B builds):  python tools/time_one.py c2|c3|prox0|prox1|wprox0|wprox1 [lambda]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
which = sys.argv[1]
lam = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
rng = np.random.default_rng(0)
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
X = dev(rng.standard_normal((4096, 4096)))
out = device.colmajor_empty((4096, 4096))
if which in ("c3", "wprox0", "wprox1"):
    W1, W2 = dev(rng.uniform(0.5 * lam, 1.5 * lam, (4095, 4096))), dev(rng.uniform(0.5 * lam, 1.5 * lam, (4096, 4095)))   # (default lambda 0.1: BASELINE config #3)
run = {"c2": lambda: device.tv1_2d(X, lam, out=out), "c3": lambda: device.tv1w_2d(X, W1, W2, out=out),
       "prox0": lambda: device.tv1_fibres(X, lam, 0, out=out), "prox1": lambda: device.tv1_fibres(X, lam, 1, out=out),
       "wprox0": lambda: device.tv1_fibres(X, 0.0, 0, weights=W1, out=out),
       "wprox1": lambda: device.tv1_fibres(X, 0.0, 1, weights=W2, out=out)}[which]
for _ in range(3): run()
torch.cuda.synchronize()
ts = []
for _ in range(7):
    t0 = time.perf_counter(); run(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
print(f"{which} lambda={lam}: min {min(ts):.3f} median {sorted(ts)[3]:.3f} ms  fixups {lib.proxtv_last_fixups()} mode {lib.proxtv_chunk_mode()}")
