"""The validation matrix inside the suite the driver runs: every pinned rung of the geometry ladder that a default policy can
land on (0 plain chunk kernels, 1 robust chunk kernels, 3 pinning solver, 5 sequential) x the jobs repair gated / always on.

Round 5's wrong result (DESIGN.md 6 (ii')) lived on a pinned rung that only builder-side environment-variable runs exercised
(profiles/r05_suite_runs.txt); here the same matrix is part of `pytest -m gpu`.  Bounded: the reference's goldens (small), and
medium images / long fibres against the CPU oracle -- sizes at which every fibre spans several chunks, segments and tile
blocks, so links, second chances, hand-overs and both repair kernels all run; the 4096^2 digests stay at the default policy
(tests/test_gpu_large.py).  Every case is also the input of a DR / Dykstra loop: late iterates (the class that exposed the
round-5 hole) come from running the loops to their end."""
import os

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

RUNGS = (0, 1, 3, 5)
JOBS = (1, 2)


@pytest.fixture(params=[(r, j) for r in RUNGS for j in JOBS], ids=lambda p: f"rung{p[0]}-jobs{p[1]}")
def pinned(clib, request):
    rung, jobs = request.param
    before = (clib.proxtv_set_option(b"chunk_mode", rung), clib.proxtv_set_option(b"repair_jobs", jobs))
    yield rung, jobs
    clib.proxtv_set_option(b"chunk_mode", before[0])
    clib.proxtv_set_option(b"repair_jobs", before[1])


def test_goldens_on_every_rung(ptv, g1d, g2d, gnd, pinned):
    """The compiled reference's own outputs (tests/golden): 1-D weighted and unweighted, DR / weighted DR / PD2 / Yang2, PD_TV / Yang3."""
    for name in g1d["names"]:
        x, lam = g1d[f"{name}/x"], float(g1d[f"{name}/lam"])
        if lam >= 0:
            assert_close(ptv.tv1_1d(x, lam), g1d[f"{name}/hybrid"], tol=1e-11, what=f"{name}:hybrid {pinned}")
        if f"{name}/weighted" in g1d:
            assert_close(ptv.tv1w_1d(x, g1d[f"{name}/w"]), g1d[f"{name}/weighted"], what=f"{name}:weighted {pinned}")
    for name in g2d["names"]:
        X, lam = g2d[f"{name}/X"], float(g2d[f"{name}/lam"])
        assert_close(ptv.tv1_2d(X, lam), g2d[f"{name}/dr2"], what=f"{name}:dr2 {pinned}")
        assert_close(ptv.tv1w_2d(X, g2d[f"{name}/W1"], g2d[f"{name}/W2"]), g2d[f"{name}/dr2w"], what=f"{name}:dr2w {pinned}")
        assert_close(ptv.tv1_2d(X, lam, method="pd"), g2d[f"{name}/pd2"], what=f"{name}:pd2 {pinned}")
        assert_close(ptv.tv1_2d(X, lam, method="yang"), g2d[f"{name}/yang2"], what=f"{name}:yang2 {pinned}")
    for name in gnd["names"]:
        X, lams = gnd[f"{name}/X"], gnd[f"{name}/lams"]
        assert_close(ptv.tvgen(X, list(lams), list(range(1, X.ndim + 1)), [1] * X.ndim), gnd[f"{name}/pd"], what=f"{name}:pd {pinned}")


def test_medium_images_on_every_rung(ptv, oracle, pinned):
    """Images whose fibres span several chunks, segments (columns of 1100+ samples) and tile blocks, at penalties from "every sample
    bends" to "pieces of ~100 samples": the loops run to their end, so the late sweeps see operands full of near-ties."""
    rung, _ = pinned
    rng = np.random.default_rng(600 + rung)
    cases = [((1150, 200), 0.1), ((300, 1200), 0.5), ((400, 1100), 1.0), ((1300, 260), 6.0)]
    for (M, N), lam in cases:
        if rung in (0, 1) and lam >= 6.0 and (M, N) == (1300, 260):
            blocks = np.kron(rng.standard_normal((M // 16 + 1, N // 16 + 1)), np.ones((16, 16)))[:M, :N]
            X = blocks + 0.2 * rng.standard_normal((M, N))       # (the data family of round 5's failing case: fuzz seed 111, case 1260)
        else:
            X = rng.standard_normal((M, N))
        assert_close(ptv.tv1_2d(X, lam), oracle.dr2(X, lam)[0], tol=1e-9, what=f"dr2 {M}x{N} lam {lam} {pinned}")
        assert_close(ptv.tv1_2d(X, lam, method="pd"), oracle.pd2(X, [lam, lam], [1, 2])[0], tol=1e-9, what=f"pd2 {M}x{N} lam {lam} {pinned}")
    (M, N), lam = (700, 420), 0.4
    X = rng.standard_normal((M, N))
    W1, W2 = rng.uniform(0.3 * lam, 1.7 * lam, (M - 1, N)), rng.uniform(0.3 * lam, 1.7 * lam, (M, N - 1))
    assert_close(ptv.tv1w_2d(X, W1, W2), oracle.dr2w(X, W1, W2)[0], tol=1e-9, what=f"dr2w {pinned}")
    assert_close(ptv.tv1_2d(X, lam, method="yang"), oracle.yang2(X, lam)[0], tol=1e-9, what=f"yang2 {pinned}")
    V = rng.standard_normal((210, 96, 33))
    assert_close(ptv.tvgen(V, [0.3, 0.2, 0.6], [1, 2, 3], [1, 1, 1]), oracle.pd(V, [0.3, 0.2, 0.6], [1, 2, 3])[0], tol=1e-9, what=f"pd {pinned}")


def test_long_fibres_and_the_degenerate_knot_on_every_rung(ptv, oracle, pinned):
    """Single fibres of several segments (the links across waves and workgroups), and the fixture of round 5's failure as a fibre, as
    columns and as rows."""
    rung, _ = pinned
    rng = np.random.default_rng(650 + rung)
    for n in (2300, 4500):
        for name, x in (("randn", rng.standard_normal(n)), ("blocks", np.repeat(rng.standard_normal(n // 50 + 1), 50)[:n] + 0.2 * rng.standard_normal(n)),
                        ("walk", np.cumsum(rng.standard_normal(n)) * 0.3)):
            for lam in (0.05, 0.7, 5.0):
                assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"{name} n={n} lam={lam} {pinned}")
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate_knot_fibre.npz"))
    y, lam, want = g["y"], float(g["lam"]), g["expected"]
    cols = np.asfortranarray(np.repeat(y[:, None], 70, axis=1))
    cols[:, 1::3] += 1e-3 * rng.standard_normal(cols[:, 1::3].shape)
    want_cols = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, cols)
    assert_close(ptv.tv1_1d(y, lam), want, tol=1e-12, what=f"the fibre alone {pinned}")
    assert_close(ptv.tvgen(cols, [lam], [1], [1]), want_cols, tol=1e-12, what=f"as columns {pinned}")
    assert_close(ptv.tvgen(np.asfortranarray(cols.T), [lam], [2], [1]), want_cols.T, tol=1e-12, what=f"as rows {pinned}")
