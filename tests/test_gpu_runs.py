"""Option "runs" (default on; proxtv_amd/csrc/chunkcore.hpp "known runs", sweep_along_kernel RUNS): on rung 0, dimension-0 sweeps over
data most of whose edges are bends known a priori cut the interior segments of their fibres at those bends and solve them run by run.
On the GPU: it engages where the statistics say so (the "why" counter of waves solved that way moves) and nowhere else, single sweeps
and whole solves agree with the CPU oracle and with the speculative path, segments that do not fit fall back, and the certifier
(option certify) finds nothing to object to."""
import ctypes as C

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def why(clib):
    before = (clib.proxtv_set_option(b"why", 1), clib.proxtv_set_option(b"chunk_mode", -1))
    if before[1] != -1:
        clib.proxtv_set_option(b"why", before[0])
        clib.proxtv_set_option(b"chunk_mode", before[1])
        pytest.skip("a pinned rung (PROXTV_CHUNK_MODE): the seeded policy decides where runs are used")
    buf = (C.c_uint * 8)()

    def read():
        clib.proxtv_debug_why(buf)
        return np.array(list(buf), dtype=np.int64)
    read()
    yield read
    clib.proxtv_set_option(b"why", before[0])


def _columns(ptv, X, lam):
    return ptv.tvgen(X, [lam], [1], [1])


def test_it_engages_on_noisy_columns_and_is_the_prox(ptv, clib, oracle, why):
    rng = np.random.default_rng(91)
    for (M, N), lam, scale in (((4096, 96), 0.1, 1.0), ((2300, 130), 0.05, 1.0), ((3300, 70), 40.0, 1e3), ((5000, 64), 1e-4, 1e-3)):
        X = rng.standard_normal((M, N)) * scale
        why()
        got = _columns(ptv, X, lam)
        w = why()
        want = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, X)
        assert_close(got, want, tol=1e-12, what=f"columns {M}x{N} lam {lam}")
        interior = (M - 9) // 1088          # segments whose window and look-ahead lie inside the fibre
        assert w[5] >= 0.9 * interior * N, (w, interior * N)      # (1 % of the segments have no known bend at their start)
        before = clib.proxtv_set_option(b"runs", 0)
        try:
            plain = _columns(ptv, X, lam)
            w0 = why()
        finally:
            clib.proxtv_set_option(b"runs", before)
        assert w0[5] == 0
        assert_close(got, plain, tol=1e-13, what="against the speculative path")


def test_it_stays_out_of_the_way_elsewhere(ptv, clib, oracle, why):
    """Longer pieces (lambda = 0.3 on unit noise: rung 1), weighted sweeps, fibres of one segment: the speculative kernels as before."""
    rng = np.random.default_rng(92)
    X = rng.standard_normal((2300, 100))
    why()
    _columns(ptv, X, 0.3)
    _columns(ptv, X[:1000], 0.1)
    ptv.tv1w_2d(X, rng.uniform(0.05, 0.15, (2299, 100)), rng.uniform(0.05, 0.15, (2300, 99)), max_iters=2)
    assert why()[5] == 0


def test_ties_blocks_and_flat_stretches_across_segment_boundaries(ptv, clib, oracle, why):
    """What the rule for two-sample runs and the fall-back have to survive: integer data (jumps of exactly 0, 2 lambda and 4 lambda),
    blocks, and stretches without a known bend laid across the segment boundaries -- short ones (the run comes in from before the
    segment) and long ones (no bend within reach: those segments take the speculative walk)."""
    rng = np.random.default_rng(93)
    cases = []
    cases.append((rng.integers(-3, 4, (3400, 80)).astype(float), 0.25))
    cases.append((rng.integers(-3, 4, (3400, 80)).astype(float), 0.5))
    cases.append((np.round(rng.standard_normal((2400, 90)) * 3) * 0.5, 0.125))
    cases.append((np.repeat(rng.standard_normal((700, 70)), 5, axis=0)[:3400] + 0.05 * rng.standard_normal((3400, 70)), 0.004))
    for width in (1, 2, 3, 16):
        X = rng.standard_normal((4500, 72)) * 2.0
        for s in range(1088, 4400, 1088):
            X[s - width // 2 - 1: s - width // 2 + width, :] = X[s] + 0.01 * rng.standard_normal((width + 1, 72))
        cases.append((X, 0.1))
    solved = 0
    for X, lam in cases:
        why()
        got = _columns(ptv, X, lam)
        solved += why()[5]
        want = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, X)
        assert_close(got, want, tol=1e-12, what=f"{X.shape} lam {lam}")
    assert solved > 500


def test_whole_solves_and_the_certifier(ptv, clib, oracle, why):
    """DR / PD2 / Yang / PD_TV with their column sweeps solved run by run: against the oracle, and under the certifier -- which looks at
    every sweep of every iteration, late iterates included -- no fibre fails."""
    rng = np.random.default_rng(94)
    X = rng.standard_normal((2400, 300))
    V = rng.standard_normal((1200, 40, 30))
    c0 = clib.proxtv_debug_counter(b"certify_failures")
    clib.proxtv_set_option(b"certify", 1)
    try:
        why()
        assert_close(ptv.tv1_2d(X, 0.1), oracle.dr2(X, 0.1)[0], tol=1e-9, what="dr2")
        assert why()[5] > 20 * 300
        assert_close(ptv.tv1_2d(X, 0.08, method="pd"), oracle.pd2(X, [0.08, 0.08], [1, 2])[0], tol=1e-9, what="pd2")
        assert_close(ptv.tv1_2d(X, 0.5, method="yang"), oracle.yang2(X, 0.5)[0], tol=1e-9, what="yang2")     # (prox at lambda / rho = 0.05)
        assert_close(ptv.tvgen(V, [0.03, 0.03, 0.03], [1, 2, 3], [1, 1, 1]), oracle.pd(V, [0.03, 0.03, 0.03], [1, 2, 3])[0], tol=1e-9, what="pd")
        assert why()[5] > 0
    finally:
        clib.proxtv_set_option(b"certify", 0)
    assert clib.proxtv_debug_counter(b"certify_failures") == c0


def test_edges_within_a_hair_of_four_lambda(ptv, clib, oracle, why):
    """tests/golden/sliver_edge_fibre.npz (found by the certifier in a full-size PD2 solve, round 6): an inner edge of -4.00000006 lambda
    between two FLOOR bends.  As columns among noise, and planted in numbers; the certifier agrees."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sliver_edge_fibre.npz"))
    y, lam, want = g["y"], float(g["lam"]), g["expected"]
    rng = np.random.default_rng(95)
    cols = np.asfortranarray(rng.standard_normal((y.size, 64)))
    cols[:, ::4] = y[:, None]
    want_cols = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, cols)
    assert np.max(np.abs(want_cols[:, 0] - want)) <= 1e-14
    c0 = clib.proxtv_debug_counter(b"certify_failures")
    clib.proxtv_set_option(b"certify", 1)
    try:
        why()
        assert_close(_columns(ptv, cols, lam), want_cols, tol=1e-13, what="the captured fibre as columns")
        assert why()[5] > 100
        Y = rng.standard_normal((3400, 64)) * 2.0
        for k in range(40, 3300, 23):
            Y[k] = Y[k - 1] + rng.choice([-1.0, 1.0], 64) * (5.0 + rng.random(64)) * lam
            Y[k + 1] = Y[k] + rng.choice([-1.0, 1.0], 64) * 4.0 * lam * (1.0 + rng.choice([0.0, 6e-9, -6e-9, 5e-8, -5e-8, 2e-7], 64))
            Y[k + 2] = Y[k + 1] + rng.choice([-1.0, 1.0], 64) * (5.0 + rng.random(64)) * lam
        wantY = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, Y)
        assert_close(_columns(ptv, Y, lam), wantY, tol=1e-13, what="planted edges")
    finally:
        clib.proxtv_set_option(b"certify", 0)
    assert clib.proxtv_debug_counter(b"certify_failures") == c0
