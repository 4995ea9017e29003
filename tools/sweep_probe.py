#!/usr/bin/env python3
"""Time ONE prox sweep (tv1_fibres, 4096^2 N(0,1) or a smooth+noise image) per lambda, pinned geometry mode and dim."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
lams = [float(v) for v in (sys.argv[1].split(',') if len(sys.argv) > 1 else ['0.1', '1', '3', '10', '30'])]
modes = [int(v) for v in (sys.argv[2].split(',') if len(sys.argv) > 2 else ['0', '1', '2', '3', '4', '5'])]
kind = sys.argv[3] if len(sys.argv) > 3 else "randn"
rng = np.random.default_rng(0)
A = rng.standard_normal((4096, 4096))
if kind == "smooth":
    g = np.linspace(0, 6 * np.pi, 4096)
    A = 3 * np.sin(g)[:, None] * np.cos(0.7 * g)[None, :] + 0.3 * A
X = device.to_colmajor(torch.from_numpy(A).cuda())
out = device.colmajor_empty((4096, 4096))
for lam in lams:
    for dim in (0, 1):
        ref = None
        row = []
        for mode in modes:
            lib.proxtv_set_option(b"chunk_mode", mode)
            device.tv1_fibres(X, lam, dim, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3): device.tv1_fibres(X, lam, dim, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 3
            if ref is None: ref = out.clone()
            row.append(f"m{mode}: {dt*1e3:8.2f} ms (fix {lib.proxtv_last_fixups():5d}, d={float((out-ref).abs().max()):.0e})")
        print(f"{kind} lam={lam:5.1f} dim={dim}  " + "  ".join(row), flush=True)
lib.proxtv_set_option(b"chunk_mode", -1)
