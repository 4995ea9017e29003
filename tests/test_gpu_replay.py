"""Replay (option "replay", proxtv_amd/csrc/chunkcore.hpp: replay_lane): from the fourth sweep of a solve on, the along-fibre kernel
verifies the structure its previous sweep recorded against the optimality conditions of the prox instead of walking.  On the GPU:
it happens (the counter of replayed wavefronts moves), the result is exact, and it is the result of the walk to rounding."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _why(clib):
    buf = np.zeros(8, dtype=np.uint32)
    assert clib.proxtv_debug_why(buf.ctypes.data) == 8
    return buf


@pytest.fixture()
def knobs(clib):
    before = {k: clib.proxtv_set_option(k, v) for k, v in ((b"replay", 1), (b"why", 1))}
    yield
    for k, v in before.items():
        clib.proxtv_set_option(k, v)


def test_replayed_sweeps_are_exact_and_agree_with_the_walk(ptv, clib, oracle, knobs):
    if clib.proxtv_set_option(b"chunk_mode", -1) not in (-1, 0):
        clib.proxtv_set_option(b"chunk_mode", -1)
        pytest.skip("a pinned rung other than 0 (PROXTV_CHUNK_MODE) keeps the plain along-fibre kernel out")
    rng = np.random.default_rng(21)
    replayed = 0
    for shape, lam in (((2400, 300), 0.1), ((4500, 130), 0.05), ((3300, 200), 0.2), ((2300, 64), 0.1)):
        X = rng.standard_normal(shape)
        want = oracle.dr2(X, lam)[0]
        clib.proxtv_set_option(b"replay", 0)
        walked = ptv.tv1_2d(X, lam)
        _why(clib)
        clib.proxtv_set_option(b"replay", 1)
        got = ptv.tv1_2d(X, lam)
        n = int(_why(clib)[5])
        replayed += n
        assert_close(got, want, tol=1e-11, what=f"replay {shape} lam {lam}")
        assert np.max(np.abs(got - walked)) <= 1e-13 * max(1.0, np.max(np.abs(want))), (shape, lam)
        # columns of two or more interior segments, 35 iterations, noisy data: most waves of the later sweeps replay
        assert n > 0, (shape, lam)
    assert replayed > 1000


def test_replay_survives_a_change_of_data_between_solves(ptv, clib, oracle, knobs):
    """The record is per geometry, not per image: a second image of the same shape finds the first one's structure in the buffer.
    Nothing is trusted -- the first sweeps of a solve only record, and what is replayed later is verified."""
    rng = np.random.default_rng(22)
    for k in range(3):
        X = rng.standard_normal((2400, 96)) * (1.0 + k)
        assert_close(ptv.tv1_2d(X, 0.1 * (1 + k)), oracle.dr2(X, 0.1 * (1 + k))[0], tol=1e-11, what=f"image {k}")
    # ... and a plain batched 1-D call (one sweep per solve) never replays
    _why(clib)
    x = rng.standard_normal((2400, 96))
    got = ptv.tv1_2d(x, 0.1, max_iters=2)
    assert int(_why(clib)[5]) == 0
    assert_close(got, oracle.dr2(x, 0.1, max_iters=2)[0], tol=1e-11, what="two iterations: recording only")
