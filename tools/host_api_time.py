"""End-to-end time of the drop-in host-pointer entry point (DR2_TV through ctypes: H2D + solve + D2H) against the
device-resident solve, 4096^2 f64."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
X = np.asfortranarray(np.random.default_rng(0).standard_normal((4096, 4096)))
out = np.zeros_like(X, order="F"); info = np.zeros(3)
def host():
    lib.DR2_TV(4096, 4096, X.ctypes.data, 0.1, 0.1, 1.0, 1.0, out.ctypes.data, 1, 0, info.ctypes.data)
def time_host():
    for _ in range(2): host()
    t0 = time.perf_counter()
    for _ in range(5): host()
    return (time.perf_counter() - t0) / 5
th = time_host()
lib.proxtv_set_option(b"host_register", 1)      # page-lock the caller's arrays around each transfer
thr = time_host()
lib.proxtv_set_option(b"host_register", 0)
# the ceiling: the same two transfers from / to memory that is already page-locked (what a caller who owns pinned buffers gets)
hp = torch.from_numpy(X.ravel(order="K").copy()).pin_memory(); hq = torch.empty_like(hp).pin_memory(); dd = torch.empty(4096 * 4096, dtype=torch.float64, device="cuda")
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5):
    dd.copy_(hp, non_blocking=True); hq.copy_(dd, non_blocking=True); torch.cuda.synchronize()
tp = (time.perf_counter() - t0) / 5
xd = device.to_colmajor(torch.from_numpy(X).cuda()); yd = device.colmajor_empty((4096, 4096))
for _ in range(2): device.tv1_2d(xd, 0.1, out=yd)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): device.tv1_2d(xd, 0.1, out=yd)
torch.cuda.synchronize(); td = (time.perf_counter() - t0) / 5
print(f"host-pointer DR2_TV: {th*1e3:.1f} ms = {16.777216/th:.0f} Mpixel/s ; device-resident: {td*1e3:.2f} ms ; transfers + staging: {(th-td)*1e3:.1f} ms "
      f"({2*134.2/(th-td)/1e3:.1f} GB/s effective over 2 x 134 MB)")
print(f"  with option host_register (hipHostRegister / Unregister around each transfer): {thr*1e3:.1f} ms")
print(f"  the two transfers alone between page-locked host memory and HBM: {tp*1e3:.2f} ms ({2*134.2/tp/1e3:.1f} GB/s)")
