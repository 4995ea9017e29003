"""4096^2 DR solves over lambda (piece length grows like (lambda / noise)^2) and the hard block image: wall time per solve
(adaptive policy, third solve), fibres repaired, geometry mode.   python tools/lambda_sweep.py [lambdas...]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
lams = [float(a) for a in sys.argv[1:]] or [0.1, 0.3, 0.5, 0.7, 1.0, 3.0, 10.0, 30.0]
X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
out = device.colmajor_empty((4096, 4096))
def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3
for lam in lams:
    device.tv1_2d(X, lam, out=out)
    ms = timed(lambda: device.tv1_2d(X, lam, out=out))
    print(f"DR 4096^2 lambda={lam:<5} {ms:9.2f} ms   fixups {lib.proxtv_last_fixups():7d}  mode {lib.proxtv_chunk_mode()}", flush=True)
r7 = np.random.default_rng(7)
Xh = device.to_colmajor(torch.from_numpy(np.kron(r7.standard_normal((8, 8)), np.ones((128, 128))) + 0.2 * r7.standard_normal((1024, 1024))).cuda())
hout = device.colmajor_empty((1024, 1024))
device.tv1_2d(Xh, 0.5, out=hout)
ms = timed(lambda: device.tv1_2d(Xh, 0.5, out=hout))
print(f"hard blocks+noise 1024^2 lambda=0.5 {ms:9.2f} ms   fixups {lib.proxtv_last_fixups():7d}  mode {lib.proxtv_chunk_mode()}")
