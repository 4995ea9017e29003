#!/usr/bin/env python3
"""Full-size solves under the certifier (option certify): every sweep of every iteration -- late iterates with their near-ties included --
is checked against the optimality conditions of the prox, fibre by fibre.  Prints one line per solve; exits non-zero if a fibre failed.

    python tools/certified_campaign.py [seconds] [seed] [high]

high: penalties from 0.4 to 4 times the edges' spread -- pieces of tens to hundreds of samples: the pinning rung with its knots known by
windows, and the mid-solve samples of the Dykstra / ADMM operands that send a solve there.

(Round 6: the first full-size certified PD2 solve found an edge of -4.00000006 lambda that the known-runs path mishandled, 3e-9 off in
two rows -- tests/golden/sliver_edge_fibre.npz.)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
high = len(sys.argv) > 3 and sys.argv[3] == "high"
rng = np.random.default_rng(seed)
lib.proxtv_set_option(b"certify", 1)
lib.proxtv_set_option(b"verbose", 1)
t_end = time.time() + budget
total = sweeps0 = 0
n = 0
bad = 0
while time.time() < t_end:
    shape = [(4096, 4096), (2300, 5000), (5000, 2300), (3300, 3300)][int(rng.integers(0, 4))]
    kind = int(rng.integers(0, 4))
    if kind == 0:   X = rng.standard_normal(shape)
    elif kind == 1: X = rng.standard_normal(shape) * 10 ** rng.uniform(-3, 3)
    elif kind == 2: X = np.round(rng.standard_normal(shape) * 4) * 0.25
    else:           X = np.cumsum(rng.standard_normal(shape), axis=int(rng.integers(0, 2))) * 0.1 + rng.standard_normal(shape)
    scale = float(np.std(np.diff(X, axis=0)))
    lam = float(scale * 10 ** (rng.uniform(-0.4, 0.6) if high else rng.uniform(-1.6, -0.2)))      # from "every edge is a bend" to pieces of a few samples
    method = ["dr", "pd", "yang", "dr", "drw", "vol"][int(rng.integers(0, 6))]
    c0 = lib.proxtv_debug_counter(b"certify_failures"); s0 = lib.proxtv_debug_counter(b"certify_sweeps")
    if method == "vol":      # a volume through PD_TV (three terms) or Yang3: prox sweeps along every dimension, short and long fibres
        shape = [(512, 512, 64), (1200, 300, 40), (96, 2300, 60)][int(rng.integers(0, 3))]
        V = rng.standard_normal(shape) * (1.0 if kind != 2 else 1.0)
        if kind == 2: V = np.round(V * 4) * 0.25
        vd = device.to_colmajor(torch.from_numpy(V).cuda())
        lams = [float(lam * f) for f in rng.uniform(0.5, 1.5, 3)]
        device.tvgen(vd, lams, [1, 2, 3], method="yang" if rng.random() < 0.4 else None)
    else:
        xd = device.to_colmajor(torch.from_numpy(X).cuda())
        if method == "drw":
            w1 = device.to_colmajor(torch.from_numpy(rng.uniform(0.5 * lam, 1.5 * lam, (shape[0] - 1, shape[1]))).cuda())
            w2 = device.to_colmajor(torch.from_numpy(rng.uniform(0.5 * lam, 1.5 * lam, (shape[0], shape[1] - 1))).cuda())
            device.tv1w_2d(xd, w1, w2)
        else:
            device.tv1_2d(xd, lam, method=method)
    f = lib.proxtv_debug_counter(b"certify_failures") - c0
    bad += f
    n += 1
    print(f"{method:4s} {shape[0]}x{shape[1]} family {kind} lambda {lam:.4g} (x sd of edges {lam / scale:.3f}): {lib.proxtv_debug_counter(b'certify_sweeps') - s0} sweeps certified, "
          f"{f} fibres failed; rung {lib.proxtv_chunk_mode()}", flush=True)
print(f"certified campaign: {n} solves, {bad} fibres failed")
sys.exit(1 if bad else 0)
