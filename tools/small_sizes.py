"""DR solve time against image size (device-resident): where the solve stops being bandwidth-bound and becomes launch-bound."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(0)
for n in (128, 256, 512, 1024, 2048, 4096, 8192):
    X = device.to_colmajor(torch.from_numpy(rng.standard_normal((n, n))).cuda())
    out = device.colmajor_empty((n, n))
    for _ in range(3): device.tv1_2d(X, 0.1, out=out)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 10 if n <= 2048 else 3
    for _ in range(reps): device.tv1_2d(X, 0.1, out=out)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"{n:5d}^2: {dt*1e3:8.3f} ms  {n*n/dt/1e6:8.1f} Mpixel/s", flush=True)
B = 64
S = device.to_colmajor(torch.from_numpy(rng.standard_normal((512, 512, B))).cuda())
so = device.colmajor_empty((512, 512, B))
for _ in range(2): device.tv1_2d_batch(S, 0.1, out=so)
torch.cuda.synchronize(); t0 = time.perf_counter(); device.tv1_2d_batch(S, 0.1, out=so); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"batch {B} x 512^2: {dt*1e3:8.3f} ms  {B*512*512/dt/1e6:8.1f} Mpixel/s", flush=True)
