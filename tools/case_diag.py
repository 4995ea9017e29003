#!/usr/bin/env python3
"""Localise a mismatch tools/fuzz.py reported in a PD2 case: which iteration, which fibre, under which options.

    python tools/case_diag.py <X.npy> <lambda> <outdir> [mode]

Runs the case under repair_jobs 0 / 1 / 2, then proximal Dykstra iteration by iteration (max_iters = k) on the GPU next to
a numpy emulation whose 1-D proxes are the oracle's, and writes the operands of the first iteration that differs to <outdir>.
"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proxtv_amd as ptv
from proxtv_amd import _lib
from oracle import cpu


def main():
    X = np.asfortranarray(np.load(sys.argv[1]))
    lam = float(sys.argv[2])
    out = sys.argv[3]
    mode = int(sys.argv[4]) if len(sys.argv) > 4 else 0
    os.makedirs(out, exist_ok=True)
    lib = _lib.require_device()
    orc = cpu.oracle()
    M, N = X.shape
    scale = float(np.max(np.abs(X)))

    def opts(jobs, mode=mode, tile=1):
        for k, v in ((b"chunk_mode", mode), (b"deterministic", 0), (b"dr_form", 0), (b"tile", tile), (b"pin_seed", 0),
                     (b"repair_jobs", jobs), (b"replay", 0)):
            lib.proxtv_set_option(k, v)

    def pd(k=0):
        y = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        l2, nrm, dm, ns = np.array([lam, lam]), np.ones(2), np.array([1.0, 2.0]), np.array(X.shape, dtype=np.int32)
        lib.PD2_TV(X.ctypes.data, l2.ctypes.data, nrm.ctypes.data, dm.ctypes.data, y.ctypes.data, info.ctypes.data, ns.ctypes.data, 2, 2, 1, k)
        _lib.check("PD2_TV")
        return y, info

    def cnt(name):
        return int(lib.proxtv_debug_counter(name))

    want, winfo = orc.pd2(X, [lam, lam], [1, 2])[:2]
    print(f"case {M}x{N} lambda {lam} mode {mode}: oracle info {winfo}", flush=True)
    for jobs in (2, 0, 1, 2, 2):
        opts(jobs)
        j0, r0 = cnt(b"repair_jobs_launches"), cnt(b"repair_launches")
        y, info = pd()
        print(f"  repair_jobs={jobs}: error {np.max(np.abs(y - want)) / scale:.3e} info {info}  jobs launches {cnt(b'repair_jobs_launches') - j0} "
              f"repair launches {cnt(b'repair_launches') - r0}", flush=True)
    for tile in (0, 1):
        for m in (-1, 1, 3):
            opts(2, m, tile)
            y, info = pd()
            print(f"  repair_jobs=2 mode {m} tile {tile}: error {np.max(np.abs(y - want)) / scale:.3e}", flush=True)

    # iteration by iteration
    prox = lambda A, axis: np.asfortranarray(np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), lam), axis, A))
    x, p, q = X.copy(), np.zeros_like(X), np.zeros_like(X)
    first = None
    for k in range(1, int(winfo[0]) + 1):
        a_in = x + p
        z = prox(a_in, 0)
        p = p + (x - z)
        b_in = z + q
        xn = prox(b_in, 1)
        q = q + (z - xn)
        x = xn
        opts(0); y0 = pd(k)[0]
        opts(2); y2 = pd(k)[0]
        e0, e2 = np.max(np.abs(y0 - x)) / scale, np.max(np.abs(y2 - x)) / scale
        d = np.abs(y2 - x)
        r, c = np.unravel_index(np.argmax(d), d.shape)
        rows, cols = np.nonzero(d > 1e-12 * scale)
        print(f"  k={k:2d}: jobs 0 vs emulation {e0:.2e}   jobs 2 vs emulation {e2:.2e}   worst at ({r},{c})  "
              f"rows {sorted(set(rows.tolist()))[:12]} cols {sorted(set(cols.tolist()))[:12]} ({rows.size} entries)", flush=True)
        if first is None and e2 > 1e-11:
            first = k
            np.savez_compressed(os.path.join(out, "first_bad.npz"), k=k, a_in=a_in, b_in=b_in, z=z, x=x, y0=y0, y2=y2, lam=lam)
            # the same operands through the plain prox (another epilogue, same walk)
            for dim, A, W in ((1, a_in, z), (2, b_in, x)):
                for jobs in (0, 2):
                    opts(jobs)
                    g = ptv.tvgen(A, [lam], [dim], [1])
                    dd = np.abs(g - W)
                    rr, cc = np.nonzero(dd > 1e-12 * scale)
                    print(f"      plain prox dim {dim} repair_jobs={jobs}: error {dd.max() / scale:.3e}  rows {sorted(set(rr.tolist()))[:8]} "
                          f"cols {sorted(set(cc.tolist()))[:8]}", flush=True)
            if k + 2 < int(winfo[0]):
                pass
        if first is not None and k >= first + 2:
            break
    opts(1, -1)
    lib.proxtv_set_option(b"deterministic", 1)
    lib.proxtv_set_option(b"dr_form", 1)
    lib.proxtv_set_option(b"pin_seed", 1)


if __name__ == "__main__":
    main()
