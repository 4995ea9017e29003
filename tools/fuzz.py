#!/usr/bin/env python3
"""Randomised differential test of the HIP path against the CPU oracle: shapes, data families, penalties over five
decades, every geometry mode pinned or adaptive, weighted / unweighted, every 2-D solver and both sweep directions.

    python tools/fuzz.py [seconds] [seed] [nd | long]
    python tools/fuzz.py <seconds> <seed> from <case>     # the same sequence, run from case number <case> on (tools/case_diag.py)
"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proxtv_amd as ptv
from proxtv_amd import _lib
from oracle import cpu


def data(rng, kind, shape):
    M, N = shape
    if kind == 0: return rng.standard_normal(shape)
    if kind == 1: return np.kron(rng.standard_normal((M // 16 + 1, N // 16 + 1)), np.ones((16, 16)))[:M, :N] + 0.2 * rng.standard_normal(shape)
    if kind == 2: return np.add.outer(np.linspace(-3, 3, M), np.linspace(2, -2, N)) + 0.05 * rng.standard_normal(shape)
    if kind == 3: return np.full(shape, 1.5) + (rng.random(shape) < 0.01) * 8.0
    if kind == 4: return np.cumsum(rng.standard_normal(shape), axis=int(rng.integers(0, 2))) * 0.3
    return np.round(rng.standard_normal(shape) * 3)          # many exact ties


def rel(a, b, scale=0.0):
    """Largest difference relative to the size of the expected result -- or of the INPUT when that is larger: with a huge
    penalty on integer data whose mean is exactly zero the expected result is rounding noise around zero (4e-15), and a
    difference of that size is not an error of 100 %."""
    return float(np.max(np.abs(a - b))) / max(float(np.max(np.abs(b))), float(scale), 1e-300)


def run(budget=60.0, seed=0, tol=1e-9, sizes=(2, 3, 17, 95, 96, 97, 130, 257, 400, 700, 1100), skip=0):
    """Returns (cases, worst relative error, description of the worst case); raises AssertionError on a mismatch.
    `skip`: the first cases of the seed are drawn but not run (to replay a run from shortly before a case of interest)."""
    lib = _lib.require_device()
    orc = cpu.oracle()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases, worst, worst_case = 0, 0.0, ""
    before = lib.proxtv_set_option(b"chunk_mode", -1)
    failures0 = lib.proxtv_debug_counter(b"certify_failures")
    try:
        while time.time() < t_end:
            M, N = (int(v) for v in rng.choice(list(sizes), 2))
            X = data(rng, int(rng.integers(0, 6)), (M, N))
            lam = float(10 ** rng.uniform(-3, 2))
            mode = int(rng.integers(-1, 6))
            lib.proxtv_set_option(b"chunk_mode", mode)
            lib.proxtv_set_option(b"deterministic", int(rng.integers(0, 2)))   # (mode -1: seeded-deterministic or hill-climbing policy)
            form = int(rng.integers(0, 3))
            lib.proxtv_set_option(b"dr_form", form)                            # (which of the two forms of the DR iteration: never / rung 1 / rungs 0, 1)
            tile, seeds = int(rng.integers(0, 2)), int(rng.choice([0, 1, 2, 2]))
            lib.proxtv_set_option(b"tile", tile)                               # (32-fibre x 4-wave or 64-fibre x 8-wave tiles)
            lib.proxtv_set_option(b"pin_seed", seeds)                          # (the pinning solver without / with the knots known a priori: jumps, windows as well)
            jobs, rep = int(rng.integers(0, 3)), int(rng.integers(0, 2))
            lib.proxtv_set_option(b"repair_jobs", jobs)                        # (failed links across workgroups one lane each: never / seeded / always)
            lib.proxtv_set_option(b"certify", rep)                             # (every sweep followed by the check of the optimality conditions: no fibre may fail)
            what = int(rng.integers(0, 6))
            W1 = W2 = None
            its = d = 0
            if what == 1: W1, W2 = rng.uniform(0, 2 * lam, (M - 1, N)), rng.uniform(0, 2 * lam, (M, N - 1))
            elif what == 4: its = int(rng.integers(1, 40))
            elif what == 5: d = int(rng.integers(1, 3))
            if cases < skip:          # (replaying a run up to a case of interest: the draws only)
                cases += 1
                continue
            if what == 0:
                got, want, name = ptv.tv1_2d(X, lam), orc.dr2(X, lam)[0], "dr2"
            elif what == 1:
                got, want, name = ptv.tv1w_2d(X, W1, W2), orc.dr2w(X, W1, W2)[0], "dr2w"
            elif what == 2:
                got, want, name = ptv.tv1_2d(X, lam, method="pd"), orc.pd2(X, [lam, lam], [1, 2])[0], "pd2"
            elif what == 3:
                got, want, name = ptv.tv1_2d(X, lam, method="yang"), orc.yang2(X, lam)[0], "yang2"
            elif what == 4:
                got, want, name = ptv.tv1_2d(X, lam, method="kolmogorov", max_iters=its), orc.kolmogorov2(X, lam, its)[0], "kolmogorov"
            else:
                got = ptv.tvgen(X, [lam], [d], [1])
                want = np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), lam), d - 1, X)
                name = f"prox dim {d}"
            e = rel(got, want, np.max(np.abs(X)))
            desc = f"{name} {M}x{N} lam={lam:.4g} mode={mode} dr_form={form} tile={tile} pin_seed={seeds} repair_jobs={jobs} certify={rep}"
            if e > worst:
                worst, worst_case = e, desc
            cases += 1
            assert e <= tol, f"MISMATCH {desc}: relative error {e:.3e}"
    finally:
        lib.proxtv_set_option(b"chunk_mode", before)
        lib.proxtv_set_option(b"deterministic", 1)
        lib.proxtv_set_option(b"dr_form", 1)
        lib.proxtv_set_option(b"tile", 1)
        lib.proxtv_set_option(b"pin_seed", 2)
        lib.proxtv_set_option(b"repair_jobs", 1)
        lib.proxtv_set_option(b"certify", 0)
    caught = lib.proxtv_debug_counter(b"certify_failures") - failures0
    assert caught == 0, f"the certifier re-solved {caught} fibres during the run: a sweep was wrong before it was repaired"
    return cases, worst, worst_case


def run_nd(budget=30.0, seed=0, tol=1e-9, sizes=(2, 5, 33, 96, 130, 210)):
    """The same for 3-D volumes: tvgen (PD_TV with 1-4 penalty terms on random dimensions), PDR_TV, Yang3_TV through the C-ABI."""
    import ctypes as C
    lib = _lib.require_device()
    orc = cpu.oracle()
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    cases, worst, worst_case = 0, 0.0, ""
    before = lib.proxtv_set_option(b"chunk_mode", -1)
    try:
        while time.time() < t_end:
            shape = tuple(int(v) for v in rng.choice(list(sizes), 3))
            if np.prod(shape) > 3_000_000:
                continue
            V = np.asfortranarray(rng.standard_normal(shape) if rng.random() < 0.6 else
                                  np.cumsum(rng.standard_normal(shape), axis=int(rng.integers(0, 3))) * 0.2)
            mode = int(rng.integers(-1, 6))
            lib.proxtv_set_option(b"chunk_mode", mode)
            lib.proxtv_set_option(b"deterministic", int(rng.integers(0, 2)))
            what = int(rng.integers(0, 3))
            if what == 0:
                npen = int(rng.integers(1, 5))
                lams = [float(10 ** rng.uniform(-2, 1)) for _ in range(npen)]
                dims = [int(rng.integers(1, 4)) for _ in range(npen)]
                got = ptv.tvgen(V, list(lams), dims, [1] * npen)
                want = (orc.pd2(V, lams, dims)[0] if npen == 2 else orc.pd(V, lams, dims)[0])   # tvgen's dispatch
                name = f"tvgen {npen} terms dims {dims}"
            elif what == 1:
                lams = np.array([float(10 ** rng.uniform(-2, 1)) for _ in range(3)])
                want = orc.pdr(V, lams, [1, 2, 3])[0]
                out, info = np.zeros(shape, order="F"), np.zeros(3)
                l2, nrm, dm, ns = lams.copy(), np.ones(3), np.array([1.0, 2.0, 3.0]), np.array(shape, dtype=np.int32)
                lib.PDR_TV(V.ctypes.data, l2.ctypes.data, nrm.ctypes.data, dm.ctypes.data, out.ctypes.data, info.ctypes.data,
                           ns.ctypes.data, 3, 3, 1, 0)
                got, name = out, "PDR_TV"
            else:
                lam = float(10 ** rng.uniform(-2, 1))
                its = int(rng.integers(1, 36))
                want = orc.yang3(V, lam, its)[0]
                out, info = np.zeros(shape, order="F"), np.zeros(3)
                lib.Yang3_TV(shape[0], shape[1], shape[2], V.ctypes.data, lam, out.ctypes.data, its, info.ctypes.data)
                got, name = out, f"Yang3_TV {its} its"
            e = rel(got, want, np.max(np.abs(V)))
            desc = f"{name} {shape} mode={mode}"
            if e > worst:
                worst, worst_case = e, desc
            cases += 1
            assert e <= tol, f"MISMATCH {desc}: relative error {e:.3e}"
    finally:
        lib.proxtv_set_option(b"chunk_mode", before)
        lib.proxtv_set_option(b"deterministic", 1)
    return cases, worst, worst_case


if __name__ == "__main__":
    if len(sys.argv) > 3 and sys.argv[3] == "nd":
        n, w, where = run_nd(float(sys.argv[1]), int(sys.argv[2]))
        print(f"fuzz nd: {n} cases, worst relative error {w:.2e} ({where})")
        sys.exit(0)
    if len(sys.argv) > 3 and sys.argv[3] == "long":   # fibres of 2-4 segments / 9-34 blocks: the machinery across workgroups
        n, w, where = run(float(sys.argv[1]), int(sys.argv[2]), sizes=(96, 130, 1089, 2177, 3300, 4353))
        print(f"fuzz long fibres: {n} cases, worst relative error {w:.2e} ({where})")
        sys.exit(0)
    if len(sys.argv) > 4 and sys.argv[3] == "from":
        n, w, where = run(float(sys.argv[1]), int(sys.argv[2]), skip=int(sys.argv[4]))
        print(f"fuzz from case {sys.argv[4]}: {n} cases, worst relative error {w:.2e} ({where})")
        sys.exit(0)
    n, w, where = run(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0, int(sys.argv[2]) if len(sys.argv) > 2 else 0)
    print(f"fuzz: {n} cases, worst relative error {w:.2e} ({where})")
