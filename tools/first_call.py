"""What a drop-in caller that makes ONE call pays: wall time of the library's initialisation and of the first calls of a fresh
process -- the first 4096^2 DR solve and the first 10^6-sample fibre -- against the steady state.
    python tools/first_call.py dr|fibre [lambda]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
torch.cuda.init()
what = sys.argv[1] if len(sys.argv) > 1 else "dr"
lam = float(sys.argv[2]) if len(sys.argv) > 2 else (0.1 if what == "dr" else 0.5)
if what == "dr":
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda().t().contiguous().t()
else:
    x = torch.from_numpy(np.random.default_rng(0).standard_normal((1_000_000,))).cuda()
out = torch.empty_like(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
from proxtv_amd import _lib, device
lib = _lib.require_device()
t_init = (time.perf_counter() - t0) * 1e3
ts = []
for k in range(6):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if what == "dr":
        device.tv1_2d(x, lam, out=out)
    else:
        device.tv1_fibres(x, lam, 0, out=out)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
name = "DR 4096^2" if what == "dr" else "fibre 10^6"
print(f"{name} lambda={lam}: load + proxtv_init {t_init:.1f} ms ; calls 1..6: " + " ".join(f"{t:.2f}" for t in ts) + f" ms ; mode {lib.proxtv_chunk_mode()}")
