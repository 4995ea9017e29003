"""Option "optimistic" (default on): a DR solve whose every sweep will run on rung 0 launches no repair kernels behind its sweeps; a
sweep that leaves anything marks one sticky word, read once at the end of the solve, and such a solve is run again with the repairs.
On the GPU: the bracket engages where the statistics say rung 0 and nowhere else, it launches no repair kernel, the result has the
bits of the plain solve, and a solve in which a sweep DOES leave something is run again and is exact."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

KEYS = ("optimistic_solves", "optimistic_redone", "repair_launches", "sweep_launches")


def _counters(clib):
    return {k: clib.proxtv_debug_counter(k.encode()) for k in KEYS}


def _delta(clib, fn):
    c0 = _counters(clib)
    out = fn()
    c1 = _counters(clib)
    return out, {k: c1[k] - c0[k] for k in KEYS}


@pytest.fixture()
def default_policy(clib):
    before = clib.proxtv_set_option(b"chunk_mode", -1)
    if before != -1:
        clib.proxtv_set_option(b"chunk_mode", before)
        pytest.skip("a pinned rung (PROXTV_CHUNK_MODE): the bracket is for the seeded policy")
    clib.proxtv_set_option(b"optimistic", 1)   # (setting the option also forgets the calling thread's back-off history)
    yield
    clib.proxtv_set_option(b"optimistic", 1)


def test_clean_solves_launch_no_repairs_and_keep_their_bits(ptv, clib, oracle, default_policy):
    rng = np.random.default_rng(81)
    for shape, lam in (((1200, 700), 0.1), ((300, 2300), 0.12), ((512, 512), 0.05)):
        X = rng.standard_normal(shape)
        y, d = _delta(clib, lambda: ptv.tv1_2d(X, lam))
        assert d["optimistic_solves"] == 1 and d["optimistic_redone"] == 0, d
        assert d["repair_launches"] == 0 and d["sweep_launches"] >= 70, d
        before = clib.proxtv_set_option(b"optimistic", 0)
        try:
            y0, d0 = _delta(clib, lambda: ptv.tv1_2d(X, lam))
        finally:
            clib.proxtv_set_option(b"optimistic", before)
        assert d0["optimistic_solves"] == 0 and d0["repair_launches"] >= 70, d0
        np.testing.assert_array_equal(y, y0)
        assert_close(y, oracle.dr2(X, lam)[0], tol=1e-9, what=f"dr2 {shape}")
    # weighted and batched solves take the same bracket
    X = rng.standard_normal((700, 600))
    W1, W2 = rng.uniform(0.05, 0.15, (699, 600)), rng.uniform(0.05, 0.15, (700, 599))
    y, d = _delta(clib, lambda: ptv.tv1w_2d(X, W1, W2))
    assert d["optimistic_solves"] == 1 and d["repair_launches"] == 0, d
    assert_close(y, oracle.dr2w(X, W1, W2)[0], tol=1e-9, what="dr2w")


def test_longer_pieces_keep_their_repairs(ptv, clib, default_policy):
    """lambda = 0.5 on unit noise: rung 1, where links do fail -- the bracket stays off."""
    X = np.random.default_rng(82).standard_normal((900, 800))
    _, d = _delta(clib, lambda: ptv.tv1_2d(X, 0.5))
    assert d["optimistic_solves"] == 0 and d["repair_launches"] > 0, d


def test_a_sweep_that_leaves_something_sends_the_solve_round_again(ptv, clib, oracle, default_policy):
    """Unit noise says rung 0; three constant columns and two constant rows (0.3 % of the image: below what the policy's flat-stretch
    statistic reacts to) are one piece each, whatever lambda: their chunks have nothing to meet at, every sweep over them is left to
    the repair kernel.  The optimistic run notices, the solve is run again with the repairs, and the result is exact."""
    rng = np.random.default_rng(83)
    X = rng.standard_normal((1400, 1250))
    X[:, [17, 600, 1249]] = np.array([0.5, -1.0, 2.0])
    X[[3, 777], :] = np.array([[1.5], [-0.25]])
    y, d = _delta(clib, lambda: ptv.tv1_2d(X, 0.1))
    assert d["optimistic_solves"] == 1 and d["optimistic_redone"] == 1, d
    assert d["repair_launches"] >= 70, d
    assert_close(y, oracle.dr2(X, 0.1)[0], tol=1e-9, what="dr2, redone")
    # ... and the bracket backs off: the next solves of this thread run with their repairs from the start, same bits
    y2, d2 = _delta(clib, lambda: ptv.tv1_2d(X, 0.1))
    assert d2["optimistic_solves"] == 0 and d2["optimistic_redone"] == 0 and d2["repair_launches"] >= 70, d2
    np.testing.assert_array_equal(y, y2)
    before = clib.proxtv_set_option(b"optimistic", 0)
    try:
        np.testing.assert_array_equal(y, ptv.tv1_2d(X, 0.1))
    finally:
        clib.proxtv_set_option(b"optimistic", before)
