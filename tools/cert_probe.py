"""Which path leaves the fibre the certifier objected to in a full-size PD2 solve, and what does that fibre look like?  (round 6, session `probe`)"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np
import proxtv_amd as ptv
from proxtv_amd import _lib
lib = _lib.require_device()
rng = np.random.default_rng(0)
X = np.asfortranarray(rng.standard_normal((4096, 4096)))
lam = 0.1
out = "gpurun_out/r6/probe"
os.makedirs(out, exist_ok=True)
for knobs in ({}, {"runs": 0}, {"chunk_mode": 5}):
    for k, v in (("runs", 1), ("chunk_mode", -1)):
        lib.proxtv_set_option(k.encode(), v)
    for k, v in knobs.items():
        lib.proxtv_set_option(k.encode(), v)
    lib.proxtv_set_option(b"certify", 1)
    c0 = lib.proxtv_debug_counter(b"certify_failures")
    ptv.tv1_2d(X, lam, method="pd")
    print(knobs, "PD2 failures:", lib.proxtv_debug_counter(b"certify_failures") - c0, flush=True)
for k, v in (("runs", 1), ("chunk_mode", -1)):
    lib.proxtv_set_option(k.encode(), v)
# the Dykstra loop by hand, column prox through the library (certified), everything else in numpy: the operand of the first failing sweep
x = X.copy(); p = np.zeros_like(X); q = np.zeros_like(X)
for it in range(12):
    yc = np.asfortranarray(x + p)
    c0 = lib.proxtv_debug_counter(b"certify_failures")
    z = ptv.tvgen(yc, [lam], [1], [1])
    bad = lib.proxtv_debug_counter(b"certify_failures") - c0
    print("iteration", it, "column prox failures", bad, flush=True)
    if bad:
        lib.proxtv_set_option(b"certify", 0)
        z0 = ptv.tvgen(yc, [lam], [1], [1])            # what the sweep leaves without the repair
        lib.proxtv_set_option(b"runs", 0)
        z1 = ptv.tvgen(yc, [lam], [1], [1])            # the speculative path
        lib.proxtv_set_option(b"runs", 1)
        lib.proxtv_set_option(b"certify", 1)
        diff = np.abs(z0 - z).max(axis=0)
        cols = np.flatnonzero(diff > 0)
        print("  columns that differ from the certified result:", cols[:10], "max diff", diff.max(), " runs=0 vs certified:", np.abs(z1 - z).max())
        for c in cols[:3]:
            np.savez(os.path.join(out, f"fibre_it{it}_col{c}.npz"), y=yc[:, c], z_runs=z0[:, c], z_plain=z1[:, c], z_cert=z[:, c], lam=lam)
        break
    p += x - z
    yr = np.asfortranarray(z + q)
    xn = ptv.tvgen(yr, [lam], [2], [1])
    q += z - xn
    x = xn
