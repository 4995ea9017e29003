#!/usr/bin/env python3
"""HOST-ONLY design study: levels of the pinning solver with and without the knots known a priori pinned before the first level,
on the inputs of the sweeps of DR solves (unit noise, fibres of 4096 samples).

    python tools/study/pin_study.py [lambda ...]
"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import links_study as L

so = os.path.join(tempfile.gettempdir(), "pin_study.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "pin_study.cpp")], check=True)
lib = C.CDLL(so)
lib.pin_levels.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_int, C.c_void_p]


def levels(fibres, lam):
    out = []
    for f in np.ascontiguousarray(fibres):
        x0, x1 = np.empty_like(f), np.empty_like(f)
        n0, n1 = C.c_int(0), C.c_int(0)
        l0 = lib.pin_levels(f.ctypes.data, f.size, lam, x0.ctypes.data, 0, C.byref(n0))
        l1 = lib.pin_levels(f.ctypes.data, f.size, lam, x1.ctypes.data, 1, C.byref(n1))
        ref = np.empty_like(f)
        L.lib.study_prox(f.ctypes.data, f.size, lam, ref.ctypes.data)
        err = max(np.abs(x0 - ref).max(), np.abs(x1 - ref).max()) / max(np.abs(ref).max(), 1e-300)
        out.append((l0, l1, n1.value, err, len(np.flatnonzero(np.diff(ref))) + 1))
    return np.array(out)


if __name__ == "__main__":
    lams = [float(v) for v in sys.argv[1:]] or [0.7, 0.8, 1.0, 2.0, 3.0]
    rng = np.random.default_rng(5)
    Ucol, Urow = rng.standard_normal((4096, 48)), rng.standard_normal((48, 4096))
    its = (1, 3, 8, 16, 25, 34)
    for lam in lams:
        cols, _ = L.dr_inputs(Ucol, lam, at=its)
        _, rows = L.dr_inputs(Urow, lam, at=its)
        print(f"lambda = {lam}")
        for k, it in enumerate(its):
            for name, arr in (("col", cols[k].T), ("row", rows[k])):
                r = levels(arr, lam)
                print(f"  it {it:2d} {name}: levels {r[:, 0].mean():5.1f} (max {r[:, 0].max():.0f}) -> {r[:, 1].mean():5.1f} (max {r[:, 1].max():.0f}) with "
                      f"{r[:, 2].mean():6.1f} knots pinned a priori of {r[:, 4].mean():6.1f} pieces; worst error vs the walk {r[:, 3].max():.1e}")
