#!/usr/bin/env python3
"""Lively stretches followed by flat ones, through the speculative rungs pinned, against the sequential walk (rung 5).

The repair kernel jumps from a chunk that took a repair walk over to the next link in doubt; the bend it starts the next walk from is
the record of the chunk just before that link -- a chunk the chunk kernel proved, whose record is therefore non-zero and true.
The data that would tell if it were otherwise: fibres whose lively stretches end on a workgroup boundary's worth of flat samples,
where every link across workgroups is in doubt.

    python tools/repair_scan_check.py [seed]        (exit status 1 on a mismatch; honours PROXTV_LIB with PROXTV_DEBUG_ALT_LIB=1)
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from proxtv_amd import _lib, device


def stretches(rng, n, m):
    """n fibres of m samples along axis 0: lively stretches (unit noise around a random level) and flat ones in alternation."""
    X = np.empty((m, n))
    for j in range(n):
        k, col = 0, X[:, j]
        lively = bool(rng.integers(0, 2))
        while k < m:
            span = int(rng.integers(30, 330)) if lively else int(rng.integers(100, 700))
            level = rng.normal() * 2
            col[k:k + span] = level + (rng.standard_normal(min(span, m - k)) if lively else 0.0)
            k += span
            lively = not lively
    return X


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
    lib = _lib.require_device()
    rng = np.random.default_rng(seed)
    dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    worst, bad, cases = 0.0, 0, 0
    for (m, n) in ((2048, 1024), (4096, 512), (1500, 777)):
        A = stretches(rng, n, m)
        for d, X in ((0, dev(A)), (1, dev(A.T))):          # fibres contiguous (along-fibre kernel) / strided (tiles)
            for lam in (0.3, 0.7, 1.0, 1.6):
                lib.proxtv_set_option(b"chunk_mode", 5)
                want = device.tv1_fibres(X, lam, d).clone()
                for mode, tile in ((0, 1), (1, 1), (1, 0), (2, 1)):
                    lib.proxtv_set_option(b"chunk_mode", mode)
                    lib.proxtv_set_option(b"tile", tile)
                    got = device.tv1_fibres(X, lam, d)
                    err = float((got - want).abs().max() / want.abs().max())
                    cases += 1
                    worst = max(worst, err)
                    if not err <= 1e-9:
                        bad += 1
                        print(f"MISMATCH {m}x{n} dim {d} lam {lam} rung {mode} tile {tile}: relative error {err:.3e}", flush=True)
    lib.proxtv_set_option(b"chunk_mode", -1)
    lib.proxtv_set_option(b"tile", 1)
    print(f"repair_scan_check seed {seed}: {cases} cases, worst {worst:.2e}, mismatches {bad}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
