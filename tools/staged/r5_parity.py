#!/usr/bin/env python3
"""r5-prep: option "repair_jobs" (one lane per failed link across workgroups) against the sequential repair, on one box.

Every case is solved twice in this process -- repair_jobs = 0 (the validated path) and 1 -- pinned to rung 1 where the
links fail, and the two results must agree to 1e-9 relative (both are exact solves; they differ by rounding only where
a fibre was rewritten by a different walk).  Prints the fix-up counters of both so that a run in which the jobs kernel
never had anything to do cannot pass for a green one.
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from proxtv_amd import _lib, device

lib = _lib.require_device()
dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
rng = np.random.default_rng(5)
worst = 0.0
bad = 0


def both(label, run):
    global worst, bad
    outs, fixes, ms = [], [], []
    for jobs in (0, 1):
        lib.proxtv_set_option(b"repair_jobs", jobs)
        run()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = run()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) * 1e3)
        outs.append((r[0] if isinstance(r, tuple) else r).clone())
        fixes.append(lib.proxtv_last_fixups())
    a, b = outs
    err = float((a - b).abs().max() / max(1e-300, float(a.abs().max())))
    worst = max(worst, err)
    ok = err <= 1e-9 and bool(torch.isfinite(b).all())
    bad += 0 if ok else 1
    print(f"{'ok  ' if ok else 'FAIL'} {label:44s} rel.diff {err:.2e}  fixups {fixes[0]} / {fixes[1]}  ms {ms[0]:.2f} / {ms[1]:.2f}", flush=True)


lib.proxtv_set_option(b"chunk_mode", 1)
for n, m in ((1024, 1024), (2048, 600), (777, 1500), (4096, 4096)):
    X = dev(rng.standard_normal((n, m)))
    for lam in (0.4, 0.65, 0.7, 0.9):
        both(f"tv1_2d DR {n}x{m} lam {lam}", lambda: device.tv1_2d(X, lam))
    for d in (0, 1):
        both(f"one sweep dim {d} {n}x{m} lam 0.7", lambda: device.tv1_fibres(X, 0.7, d))
for tile in (0, 1):
    lib.proxtv_set_option(b"tile", tile)
    X = dev(rng.standard_normal((1536, 1536)))
    both(f"tv1_2d DR 1536^2 lam 0.7 tile={tile}", lambda: device.tv1_2d(X, 0.7))
    W1, W2 = dev(rng.uniform(0.35, 1.05, (1535, 1536))), dev(rng.uniform(0.35, 1.05, (1536, 1535)))
    both(f"weighted DR 1536^2 w~0.7 tile={tile}", lambda: device.tv1w_2d(X, W1, W2))
lib.proxtv_set_option(b"tile", 1)
V = dev(rng.standard_normal((256, 320, 24)))
both("tvgen PD 256x320x24 lam 0.7", lambda: device.tvgen(V, [0.7, 0.7, 0.7], [1, 2, 3]))
both("tvgen Yang 256x320x24 lam 0.7", lambda: device.tvgen(V, [0.7, 0.7, 0.7], [1, 2, 3], method="yang"))
X = dev(rng.standard_normal((2048, 2048)))
both("tv1_2d PD2 2048^2 lam 0.7", lambda: device.tv1_2d(X, 0.7, method="pd"))
print(f"worst {worst:.2e}  failures {bad}")
sys.exit(1 if bad else 0)
