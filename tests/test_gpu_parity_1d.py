"""HIP path vs golden vectors and vs the CPU oracle: 1-D entry points (tv1_1d / tv1w_1d and their C symbols).

All calls go through the C-ABI of libproxtv_amd.so.  Tolerance: BASELINE.json's 1e-6 relative (conftest.REL_TOL);
the tight second assertion (1e-11) documents what the exact solver really achieves.
"""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu

UNWEIGHTED = [
    # (python method name, maxbacktracks, golden key)
    ("hybridtautstring", None, "hybrid"),
    ("hybridtautstring", 1.2, "hybrid_1p2"),
    ("hybridtautstring", 0.5, "hybrid_0p5"),
    ("linearizedtautstring", None, "linearized"),
    ("classictautstring", None, "classic"),
    ("condat", None, "condat"),
]


def test_golden_unweighted(ptv, g1d):
    for name in g1d["names"]:
        x, lam = g1d[f"{name}/x"], float(g1d[f"{name}/lam"])
        if lam < 0:
            continue   # the Python surface asserts w >= 0 like the reference; covered through the C symbol below
        for method, mb, key in UNWEIGHTED:
            got = ptv.tv1_1d(x, lam, method=method, maxbacktracks=mb)
            assert got.shape == (x.size,)
            assert_close(got, g1d[f"{name}/{key}"], what=f"{name}:{key}")
            assert_close(got, g1d[f"{name}/{key}"], tol=1e-11, what=f"{name}:{key} (tight)")


def test_golden_alias_methods(ptv, g1d):
    """'pn', 'dp', 'condattautstring', 'kolmogorov' are aliases of the exact solver (unique minimiser)."""
    for name in ("randn_n100", "blocks_noise", "ka_zigzag"):
        x, lam = g1d[f"{name}/x"], float(g1d[f"{name}/lam"])
        for method in ("pn", "dp", "condattautstring", "kolmogorov"):
            assert_close(ptv.tv1_1d(x, lam, method=method), g1d[f"{name}/hybrid"], what=f"{name}:{method}")


def test_other_methods_against_the_references_own_outputs(ptv, g1dm):
    """Same method names, against what the compiled reference returns for each of them (not for its hybrid solver)."""
    for name in g1dm["names"]:
        x, lam = g1dm[f"{name}/x"], float(g1dm[f"{name}/lam"])
        for method in ("pn", "kolmogorov", "condattautstring", "dp"):
            assert_close(ptv.tv1_1d(x, lam, method=method), g1dm[f"{name}/{method}"], what=f"{name}:{method}")


def test_golden_weighted(ptv, g1d):
    for name in g1d["names"]:
        if f"{name}/weighted" not in g1d:
            continue
        x = g1d[f"{name}/x"]
        assert_close(ptv.tv1w_1d(x, g1d[f"{name}/w"]), g1d[f"{name}/weighted"], what=f"{name}:weighted")
        lam = float(g1d[f"{name}/lam"])
        assert_close(ptv.tv1w_1d(x, np.full(x.size - 1, lam)), g1d[f"{name}/weighted_uniform"],
                     what=f"{name}:weighted_uniform")


def test_known_answers(ptv):
    """All five reference solvers agree on these."""
    cases = [([3.0], 1.0, [3.0]), ([1, 5], 1.0, [2, 4]), ([1, 5], 10.0, [3, 3]), ([2, 2, 2, 2], 0.5, [2, 2, 2, 2]),
             ([1, 4, 2, 8, 3], 0.0, [1, 4, 2, 8, 3]), ([0, 10, 0, 10, 0], 2.0, [2, 6, 4, 6, 2])]
    for x, lam, want in cases:
        for method in ("hybridtautstring", "linearizedtautstring", "classictautstring", "condat"):
            got = ptv.tv1_1d(np.array(x, dtype=float), lam, method=method)
            np.testing.assert_allclose(got, want, atol=1e-12, err_msg=f"{x} {lam} {method}")


def test_negative_lambda_through_c_symbol(clib, g1d):
    """The reference's tvgen smoke test feeds negative weights (prox_tv_test.py:202-209); the walker must behave
    like the reference's, not hang."""
    x, lam = g1d["neg_lambda/x"], float(g1d["neg_lambda/lam"])
    out = np.zeros_like(x)
    clib.hybridTautString_TV1(x.ctypes.data, x.size, lam, out.ctypes.data)
    assert_close(out, g1d["neg_lambda/hybrid"], what="neg_lambda hybrid")
    out2 = np.zeros_like(x)
    clib.linearizedTautString_TV1(x.ctypes.data, lam, out2.ctypes.data, x.size)
    assert_close(out2, g1d["neg_lambda/linearized"], what="neg_lambda linearized")


def test_int_and_list_inputs(ptv, oracle):
    """prox_tv_test.py:47-53 feeds integer arrays; lists are accepted through force_float_matrix."""
    x = (100 * np.random.default_rng(3).standard_normal(25)).astype("int")
    assert_close(ptv.tv1_1d(x, 7.5), oracle.tv1_hybrid(x.astype(float), 7.5))
    assert_close(ptv.tv1_1d([1, 5, 2, 8], 1), oracle.tv1_hybrid(np.array([1.0, 5, 2, 8]), 1.0))
    # 2-D input is flattened by np.size like the reference and returns 1-D
    x2 = np.random.default_rng(4).standard_normal((4, 6))
    assert ptv.tv1_1d(x2, 0.3).shape == (24,)


def test_random_vs_oracle(ptv, oracle):
    """Mirrors prox_tv_test.py:18-62 with seeds: all methods agree with the oracle over many random sizes."""
    rng = np.random.default_rng(11)
    for trial in range(60):
        n = int(rng.integers(1, 600))
        x = 100 * rng.standard_normal(n)
        w = 20 * rng.random()
        want = oracle.tv1_hybrid(x, w)
        for method in ("hybridtautstring", "linearizedtautstring", "classictautstring", "condat"):
            assert_close(ptv.tv1_1d(x, w, method=method), want, what=f"n={n} {method}")
        if n >= 2:
            w1 = rng.random()
            assert_close(ptv.tv1w_1d(x, np.ones(n - 1) * w1), oracle.tv1_hybrid(x, w1), what=f"n={n} uniform weights")
            wv = 20 * rng.random(n - 1)
            assert_close(ptv.tv1w_1d(x, wv), oracle.tv1_weighted(x, wv), what=f"n={n} weighted")


def test_heavy_backtracking_long_fibre(ptv, oracle):
    """Piecewise-constant + noise with a large lambda: long segments, long rewinds."""
    rng = np.random.default_rng(5)
    x = np.repeat(rng.standard_normal(40), 250) + 0.2 * rng.standard_normal(10000)
    for lam in (0.5, 5.0, 200.0):
        assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), what=f"blocks lam={lam}")


def test_long_fibres_at_penalties_around_the_noise_level(ptv, clib, oracle):
    """Fibres beyond one workgroup's LDS (16384 samples) under the default policy: where the certain fraction asks for the pinning rung --
    there a cooperative grid, milliseconds whatever the data -- but a tenth of the edges still exceeds one penalty, 64-sample zones take
    the sweep (rung 2: a fraction of a millisecond); with longer pieces the pinning solver.  Exact either way."""
    rng = np.random.default_rng(8)
    for n in (20000, 300000):
        x = rng.standard_normal(n) + 0.05 * np.cumsum(rng.standard_normal(n))
        for lam, rung in ((0.5, 1), (1.0, 2), (2.0, 2), (6.0, 3)):
            assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), what=f"n={n} lam={lam}")
            assert clib.proxtv_chunk_mode() == rung, (n, lam, clib.proxtv_chunk_mode())
    X = rng.standard_normal((24, 40000))                                                 # long STRIDED fibres (rows), and long columns
    for lam in (1.0, 4.0):
        assert_close(ptv.tv1_2d(X, lam, max_iters=3), oracle.dr2(X, lam, max_iters=3)[0], tol=1e-9, what=f"24 x 40000 lam={lam}")
        assert_close(ptv.tv1_2d(X.T.copy(), lam, max_iters=3), oracle.dr2(X.T.copy(), lam, max_iters=3)[0], tol=1e-9, what=f"40000 x 24 lam={lam}")
    x = np.repeat(rng.standard_normal(60), 5000) + 0.3 * rng.standard_normal(300000)     # blocks: lively edges, pieces of thousands
    for lam in (0.4, 1.0, 3.0):
        assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), what=f"blocks lam={lam}")


def test_classic_offset_symbol(clib, oracle):
    rng = np.random.default_rng(6)
    x = rng.standard_normal(300)
    for off in (0.4, -0.4):
        out = np.zeros_like(x)
        clib.classicTautString_TV1_offset(x.ctypes.data, x.size, 0.4, out.ctypes.data, off)
        assert_close(out, oracle.tv1_classic(x, 0.4, off), what=f"offset {off}")


def test_tv_dispatcher_symbol(clib, oracle):
    x = np.random.default_rng(7).standard_normal(200)
    out, info = np.zeros_like(x), np.full(3, -1.0)
    rc = clib.TV(x.ctypes.data, 0.3, out.ctypes.data, info.ctypes.data, x.size, 1.0, None)
    assert rc == 1 and list(info) == [0.0, 0.0, 0.0]
    assert_close(out, oracle.tv1_hybrid(x, 0.3))
    info[:] = -1
    assert clib.TV(x.ctypes.data, 0.3, out.ctypes.data, info.ctypes.data, x.size, 0.5, None) == 0 and info[2] == 3
    info[:] = -1
    assert clib.TV(x.ctypes.data, 0.3, out.ctypes.data, info.ctypes.data, x.size, 3.0, None) == 0 and info[2] == 3   # general p
    info[:] = -1
    assert clib.TV(x.ctypes.data, 0.3, out.ctypes.data, info.ctypes.data, x.size, 2.0, None) == 1 and info[2] == 0   # TV-L2
    assert_close(out, oracle.tv(x, 0.3, 2)[0], tol=1e-10)


def test_condat_inplace_and_noop(clib, oracle):
    x = np.random.default_rng(8).standard_normal(128)
    want = oracle.tv1_condat(x, 0.7)
    buf = x.copy()
    clib.TV1D_denoise(buf.ctypes.data, buf.ctypes.data, buf.size, 0.7)   # in place (src/condat_fast_tv.cpp:72-76)
    assert_close(buf, want)
    untouched = np.full(5, 9.0)
    clib.TV1D_denoise(x.ctypes.data, untouched.ctypes.data, 0, 0.7)        # width <= 0: nothing is done
    clib.TV1D_denoise(x.ctypes.data, untouched.ctypes.data, 5, -1.0)       # lambda < 0: nothing is done
    assert (untouched == 9.0).all()


def test_c1_full_size(ptv, glarge):
    """BASELINE config #1: tv1_1d on a 1e6 float64 N(0,1) signal, lambda 0.5, Condat entry point."""
    x = np.random.default_rng(0).standard_normal(1_000_000)
    step = int(glarge["step"])
    np.testing.assert_array_equal(x[::step], glarge["c1/x/sub"])            # same synthetic input as the fixture
    y = ptv.tv1_1d(x, 0.5, method="condat")
    assert_close(y[::step], glarge["c1/condat/sub"], what="c1 subsample")
    assert abs(y.sum() - float(glarge["c1/condat/sum"])) <= 1e-6 * float(glarge["c1/condat/abs"])
    assert abs((y * y).sum() - float(glarge["c1/condat/sq"])) <= 1e-6 * float(glarge["c1/condat/sq"])
    # size-independent properties of the prox: mean is preserved, TV does not increase, idempotent at lambda = 0
    assert abs(y.mean() - x.mean()) < 1e-9
    assert np.abs(np.diff(y)).sum() <= np.abs(np.diff(x)).sum()


def test_alternative_algorithm_symbols(clib, oracle):
    """The remaining 1-D symbols of the reference's cdef (PN_TV1, PN_TV1_Weighted, SolveTVConvexQuadratic_a1[_nw],
    TV1D_denoise_tautstring, dp, the five *_TVp): reference argument order, trivial cases and info conventions."""
    rng = np.random.default_rng(21)
    x = rng.standard_normal(777)
    w = rng.uniform(0.1, 0.9, x.size - 1)
    want, wantw = oracle.tv1_hybrid(x, 0.4), oracle.tv1_weighted(x, w)
    out, info = np.zeros_like(x), np.full(3, -1.0)
    assert clib.PN_TV1(x.ctypes.data, 0.4, out.ctypes.data, info.ctypes.data, x.size, 0.05, None) == 1
    assert list(info) == [0.0, 0.0, 0.0]
    assert_close(out, want)
    out[:] = 0; info[:] = -1
    assert clib.PN_TV1_Weighted(x.ctypes.data, w.ctypes.data, out.ctypes.data, info.ctypes.data, x.size, 0.05, None) == 1
    assert info[2] == 0
    assert_close(out, wantw)
    out[:] = 0
    clib.SolveTVConvexQuadratic_a1_nw(x.size, x.ctypes.data, 0.4, out.ctypes.data)
    assert_close(out, want)
    out[:] = 0
    clib.SolveTVConvexQuadratic_a1(x.size, x.ctypes.data, w.ctypes.data, out.ctypes.data)
    assert_close(out, wantw)
    out[:] = 0
    clib.TV1D_denoise_tautstring(x.ctypes.data, out.ctypes.data, x.size, 0.4)
    assert_close(out, want)
    out[:] = 0
    clib.dp(x.size, x.ctypes.data, 0.4, out.ctypes.data)
    assert_close(out, want)
    # trivial cases, as the reference handles them before any work
    one, o1 = np.array([5.0]), np.array([-1.0])
    clib.SolveTVConvexQuadratic_a1_nw(1, one.ctypes.data, 0.4, o1.ctypes.data)
    assert o1[0] == 5.0
    o1[:] = -1
    clib.dp(1, one.ctypes.data, 0.4, o1.ctypes.data)
    assert o1[0] == 5.0
    out[:] = 0
    clib.dp(x.size, x.ctypes.data, 0.0, out.ctypes.data)
    assert (out == x).all()
    untouched = np.full(4, 9.0)
    clib.dp(0, x.ctypes.data, 0.4, untouched.ctypes.data)
    clib.TV1D_denoise_tautstring(x.ctypes.data, untouched.ctypes.data, 0, 0.4)
    clib.SolveTVConvexQuadratic_a1(0, x.ctypes.data, w.ctypes.data, untouched.ctypes.data)
    assert (untouched == 9.0).all()
    # TV-Lp schemes: p in {1, 2} through the exact solvers, general p -> RC_ERROR
    for fn in (clib.GP_TVp, clib.OGP_TVp, clib.FISTA_TVp, clib.FW_TVp, clib.GPFW_TVp):
        out[:] = 0; info[:] = -1
        assert fn(x.ctypes.data, 0.4, out.ctypes.data, info.ctypes.data, x.size, 1.0, None) == 1 and info[2] == 0
        assert_close(out, want)
        info[:] = -1
        assert fn(x.ctypes.data, 0.4, out.ctypes.data, info.ctypes.data, x.size, 1.5, None) == 0 and info[2] == 3
    out[:] = 0; info[:] = -1
    assert clib.GPFW_TVp(x.ctypes.data, 0.4, out.ctypes.data, info.ctypes.data, x.size, 2.0, None) == 1 and info[2] == 0
    assert_close(out, oracle.tv(x, 0.4, 2)[0], tol=1e-10)
