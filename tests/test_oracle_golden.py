"""Pins the CPU oracle (oracle/*.c, this repo's restatement) -- runs without a GPU.

1. against every golden vector in tests/golden/ (outputs of the compiled reference, made by oracle/gen_golden.py);
2. where oracle/_ref/libproxtv_ref.so exists (the build container, or the GPU box when the .so travelled),
   against the compiled reference itself on fresh seeded inputs, bit for bit.
The oracle follows the reference's arithmetic order, so the bar here is 1e-12 relative, far below the product's 1e-6.
"""
import numpy as np
import pytest

from conftest import assert_close

TIGHT = 1e-12


def test_golden_1d(oracle, g1d):
    for name in g1d["names"]:
        x, lam = g1d[f"{name}/x"], float(g1d[f"{name}/lam"])
        assert_close(oracle.tv1_hybrid(x, lam), g1d[f"{name}/hybrid"], TIGHT, f"{name}:hybrid")
        assert_close(oracle.tv1_hybrid(x, lam, 1.2), g1d[f"{name}/hybrid_1p2"], TIGHT, f"{name}:hybrid_1p2")
        assert_close(oracle.tv1_hybrid(x, lam, 0.5), g1d[f"{name}/hybrid_0p5"], TIGHT, f"{name}:hybrid_0p5")
        assert_close(oracle.tv1_linearized(x, lam), g1d[f"{name}/linearized"], TIGHT, f"{name}:linearized")
        if f"{name}/classic" in g1d:
            assert_close(oracle.tv1_classic(x, lam), g1d[f"{name}/classic"], TIGHT, f"{name}:classic")
            assert_close(oracle.tv1_condat(x, lam), g1d[f"{name}/condat"], TIGHT, f"{name}:condat")
        if f"{name}/weighted" in g1d:
            assert_close(oracle.tv1_weighted(x, g1d[f"{name}/w"]), g1d[f"{name}/weighted"], TIGHT, f"{name}:weighted")
            assert_close(oracle.tv1_weighted(x, np.full(x.size - 1, lam)), g1d[f"{name}/weighted_uniform"], TIGHT,
                         f"{name}:weighted_uniform")


def test_known_answers(oracle):
    cases = [([3.0], 1.0, [3.0]), ([1, 5], 1.0, [2, 4]), ([1, 5], 10.0, [3, 3]), ([2, 2, 2, 2], 0.5, [2, 2, 2, 2]),
             ([1, 4, 2, 8, 3], 0.0, [1, 4, 2, 8, 3]), ([0, 10, 0, 10, 0], 2.0, [2, 6, 4, 6, 2])]
    for x, lam, want in cases:
        x = np.array(x, dtype=float)
        for got in (oracle.tv1_hybrid(x, lam), oracle.tv1_linearized(x, lam), oracle.tv1_classic(x, lam),
                    oracle.tv1_condat(x, lam)):
            np.testing.assert_allclose(got, want, atol=1e-12)


def test_golden_2d(oracle, g2d):
    for name in g2d["names"]:
        X, lam = g2d[f"{name}/X"], float(g2d[f"{name}/lam"])
        y, info, rc = oracle.dr2(X, lam)
        assert_close(y, g2d[f"{name}/dr2"], TIGHT, f"{name}:dr2")
        assert rc == int(g2d[f"{name}/dr2_rc"]) and info[0] == g2d[f"{name}/dr2_info"][0]
        assert_close(oracle.dr2(X, lam, max_iters=7)[0], g2d[f"{name}/dr2_it7"], TIGHT, f"{name}:dr2_it7")
        assert_close(oracle.dr2(X, lam, 0.5 * lam)[0], g2d[f"{name}/dr2_aniso"], TIGHT, f"{name}:dr2_aniso")
        y, info, rc = oracle.dr2w(X, g2d[f"{name}/W1"], g2d[f"{name}/W2"])
        assert_close(y, g2d[f"{name}/dr2w"], TIGHT, f"{name}:dr2w")
        assert rc == int(g2d[f"{name}/dr2w_rc"])
        y, info, rc, _ = oracle.pd2(X, [lam, lam], [1, 2])
        assert_close(y, g2d[f"{name}/pd2"], TIGHT, f"{name}:pd2")
        np.testing.assert_allclose(info, g2d[f"{name}/pd2_info"], rtol=1e-12, atol=0)
        y, info, rc, _ = oracle.pd2(X, [lam, lam], [1, 2], max_iters=3)
        assert_close(y, g2d[f"{name}/pd2_it3"], TIGHT, f"{name}:pd2_it3")
        assert_close(oracle.pd2(X, [lam], [2])[0], g2d[f"{name}/pd2_single"], TIGHT, f"{name}:pd2_single")
        assert_close(oracle.pd(X, [lam], [1])[0], g2d[f"{name}/pd_single"], TIGHT, f"{name}:pd_single")
        y, info, rc = oracle.yang2(X, lam)
        assert_close(y, g2d[f"{name}/yang2"], TIGHT, f"{name}:yang2")
        assert info[0] == g2d[f"{name}/yang2_info"][0] == 36
    a = g2d["emengd/X"]
    assert_close(oracle.dr2w(a, np.ones((2, 3)), np.ones((3, 2)), max_iters=100)[0], g2d["emengd/dr2w_it100"], TIGHT)
    assert_close(oracle.dr2(a, 1.0)[0], g2d["emengd/dr2"], TIGHT)
    assert_close(oracle.dr2(g2d["blocks/X"], 0.5)[0], g2d["blocks/dr2_l0p5"], TIGHT)
    assert_close(oracle.pd2(g2d["blocks/X"], [0.5, 0.5], [1, 2])[0], g2d["blocks/pd2_l0p5"], TIGHT)
    y, info, rc, lam_after = oracle.pd(g2d["multireg/X"], g2d["multireg/lams"], g2d["multireg/dims"], max_iters=1000)
    assert_close(y, g2d["multireg/pd_it1000"], TIGHT)
    np.testing.assert_array_equal(lam_after, g2d["multireg/lams_after"])


def test_golden_nd(oracle, gnd):
    for name in gnd["names"]:
        X, lams = gnd[f"{name}/X"], gnd[f"{name}/lams"]
        nd = X.ndim
        dims = list(range(1, nd + 1))
        y, info, rc, lam_after = oracle.pd(X, lams, dims)
        assert_close(y, gnd[f"{name}/pd"], TIGHT, f"{name}:pd")
        np.testing.assert_allclose(info, gnd[f"{name}/pd_info"], rtol=1e-12, atol=0)
        np.testing.assert_array_equal(lam_after, gnd[f"{name}/pd_lams_after"])
        assert_close(oracle.pd(X, lams, dims, max_iters=4)[0], gnd[f"{name}/pd_it4"], TIGHT, f"{name}:pd_it4")
        y, info, rc, _ = oracle.pdr(X, lams, dims)
        assert_close(y, gnd[f"{name}/pdr"], TIGHT, f"{name}:pdr")
        np.testing.assert_allclose(info, gnd[f"{name}/pdr_info"], rtol=1e-12, atol=0)
        assert_close(oracle.pd2(X, [lams[0], lams[nd - 1]], [1, nd])[0], gnd[f"{name}/pd2_first_last"], TIGHT)
        assert_close(oracle.pd2(X, [lams[1], lams[1]], [2, 2])[0], gnd[f"{name}/pd2_22"], TIGHT)
        if f"{name}/yang3" in gnd:
            y, info, rc = oracle.yang3(X, 0.2)
            assert_close(y, gnd[f"{name}/yang3"], TIGHT, f"{name}:yang3")
            assert info[0] == 36 and rc == int(gnd[f"{name}/yang3_rc"])
            assert_close(oracle.yang3(X, 0.2, max_iters=5)[0], gnd[f"{name}/yang3_it5"], TIGHT)
    assert_close(oracle.pd2(gnd["color/X"], [0.15, 0.15], [1, 2])[0], gnd["color/pd2_12"], TIGHT)


def test_yang_perdim_extension_consistent(oracle):
    """The oracle's per-dimension-lambda Yang (used to check the product's extension) reduces to the pinned scalar one."""
    X = np.random.default_rng(2).standard_normal((9, 8, 7))
    np.testing.assert_array_equal(oracle.yang3(X, [0.3, 0.3, 0.3])[0], oracle.yang3(X, 0.3)[0])


def test_other_1d_methods_of_the_reference_agree_with_the_exact_solver(oracle, g1dm):
    """tv1_1d's other method names ('pn', 'kolmogorov', 'condattautstring', 'dp'; prox_tv/__init__.py:163-172) have no
    restatement here: the product serves them with the exact solver.  These vectors are what the compiled reference
    returns for them; the restated hybrid solver (bit-identical to the reference's) matches every one of them within
    the parity bar (Condat's taut-string variant is the loosest, 4e-8)."""
    for name in g1dm["names"]:
        x, lam = g1dm[f"{name}/x"], float(g1dm[f"{name}/lam"])
        mine = oracle.tv1_hybrid(x, lam)
        np.testing.assert_array_equal(mine, g1dm[f"{name}/hybrid"])
        for m in ("pn", "kolmogorov", "condattautstring", "dp"):
            assert_close(mine, g1dm[f"{name}/{m}"], 1e-6, f"{name}:{m}")
            assert_close(mine, g1dm[f"{name}/{m}"], 1e-7 if m == "condattautstring" else 1e-12, f"{name}:{m} (tight)")


def test_johnson_dp_restatement(oracle, g1dm):
    """orc_dp -- an exact algorithm of a different kind (dynamic programming on piecewise-linear derivatives) -- against
    what the compiled reference's `dp` returned, bit for bit, and against the taut-string solvers."""
    for name in g1dm["names"]:
        x, lam = g1dm[f"{name}/x"], float(g1dm[f"{name}/lam"])
        got = oracle.tv1_dp(x, lam)
        np.testing.assert_array_equal(got, g1dm[f"{name}/dp"])
        assert_close(got, oracle.tv1_hybrid(x, lam), 1e-12, f"{name}: dp vs taut string")
    rng = np.random.default_rng(5)
    for _ in range(300):
        n = int(rng.integers(1, 120))
        x = rng.standard_normal(n) * float(rng.choice([1, 100]))
        lam = float(rng.choice([0, 0.05, 0.7, 5, 300]))
        assert_close(oracle.tv1_dp(x, lam), oracle.tv1_linearized(x, lam), 1e-11, f"n={n} lam={lam}")


def test_golden_2d_primal_dual(oracle, gpd):
    """Kolmogorov2_TV / CondatChambollePock2_TV restatements against vectors of the compiled reference."""
    for name in gpd["names"]:
        X, lam = gpd[f"{name}/X"], float(gpd[f"{name}/lam"])
        y, info, rc = oracle.kolmogorov2(X, lam)
        assert_close(y, gpd[f"{name}/kol"], TIGHT, f"{name}:kol")
        assert rc == int(gpd[f"{name}/kol_rc"]) == 1
        np.testing.assert_array_equal(info, gpd[f"{name}/kol_info"])
        y, info, rc = oracle.kolmogorov2(X, lam, max_iters=40)
        assert_close(y, gpd[f"{name}/kol_it40"], TIGHT, f"{name}:kol_it40")
        assert info[0] == gpd[f"{name}/kol_it40_info"][0] == 41
        for alg in (0, 1, 2):
            y, info, rc = oracle.ccp2(X, lam, alg)
            assert_close(y, gpd[f"{name}/ccp{alg}"], TIGHT, f"{name}:ccp{alg}")
            assert rc == int(gpd[f"{name}/ccp{alg}_rc"]) == 1
            np.testing.assert_array_equal(info, gpd[f"{name}/ccp{alg}_info"])
            y, info, rc = oracle.ccp2(X, lam, alg, max_iters=60)
            assert_close(y, gpd[f"{name}/ccp{alg}_it60"], TIGHT, f"{name}:ccp{alg}_it60")
            assert info[0] == gpd[f"{name}/ccp{alg}_it60_info"][0] == 61
    C = gpd["const/X"]
    for alg in (0, 1, 2):   # a constant image is a fixed point: the loop exits through `stop > 0` after one iteration
        y, info, rc = oracle.ccp2(C, 0.7, alg)
        np.testing.assert_array_equal(y, gpd[f"const/ccp{alg}"])
        np.testing.assert_array_equal(info, gpd[f"const/ccp{alg}_info"])
        assert info[0] == 2
    y, info, rc = oracle.kolmogorov2(C, 0.7)
    assert_close(y, gpd["const/kol"], TIGHT)
    np.testing.assert_array_equal(info, gpd["const/kol_info"])
    y, info, rc = oracle.ccp2(C, 0.7, 5)
    assert rc == int(gpd["const/ccp_bad_rc"]) == 0 and info[2] == gpd["const/ccp_bad_info"][2] == 3


def test_bitwise_against_compiled_reference(oracle, reference):
    """Fresh seeded inputs, restatement vs the unmodified reference build: identical to the last bit."""
    rng = np.random.default_rng(99)
    for trial in range(120):
        n = int(rng.integers(1, 500))
        kind = trial % 3
        if kind == 0:
            x = rng.standard_normal(n)
        elif kind == 1:
            x = np.repeat(rng.standard_normal(n // 20 + 1), 20)[:n] + 0.2 * rng.standard_normal(n)
        else:
            x = np.cumsum(rng.standard_normal(n))
        lam = float(rng.choice([0, 0.01, 0.1, 0.5, 1, 5, 50]))
        for fo, fr in ((oracle.tv1_linearized, reference.tv1_linearized), (oracle.tv1_hybrid, reference.tv1_hybrid),
                       (oracle.tv1_classic, reference.tv1_classic), (oracle.tv1_condat, reference.tv1_condat)):
            np.testing.assert_array_equal(fo(x, lam), fr(x, lam))
        np.testing.assert_array_equal(oracle.tv1_hybrid(x, lam, 0.5), reference.tv1_hybrid(x, lam, 0.5))
        np.testing.assert_array_equal(oracle.tv1_dp(x, lam), reference.tv1_dp(x, lam))
        if n >= 2:
            w = rng.uniform(0, 2 * lam + 0.01, n - 1)
            np.testing.assert_array_equal(oracle.tv1_weighted(x, w), reference.tv1_weighted(x, w))
    for trial in range(6):
        M, N, O = (int(v) for v in rng.integers(2, 30, 3))
        X, V = rng.standard_normal((M, N)), rng.standard_normal((M, N, O))
        lam = float(rng.choice([0.05, 0.3, 2.0]))
        np.testing.assert_array_equal(oracle.dr2(X, lam)[0], reference.dr2(X, lam)[0])
        W1, W2 = rng.uniform(0, 1, (M - 1, N)), rng.uniform(0, 1, (M, N - 1))
        np.testing.assert_array_equal(oracle.dr2w(X, W1, W2)[0], reference.dr2w(X, W1, W2)[0])
        np.testing.assert_array_equal(oracle.pd2(X, [lam, lam], [1, 2])[0], reference.pd2(X, [lam, lam], [1, 2])[0])
        np.testing.assert_array_equal(oracle.yang2(X, lam)[0], reference.yang2(X, lam)[0])
        np.testing.assert_array_equal(oracle.kolmogorov2(X, lam, 150)[0], reference.kolmogorov2(X, lam, 150)[0])
        for alg in (0, 1, 2):
            np.testing.assert_array_equal(oracle.ccp2(X, lam, alg, 300)[0], reference.ccp2(X, lam, alg, 300)[0])
        np.testing.assert_array_equal(oracle.pd(V, [lam, lam, lam / 2], [1, 2, 3])[0], reference.pd(V, [lam, lam, lam / 2], [1, 2, 3])[0])
        np.testing.assert_array_equal(oracle.pdr(V, [lam, lam, lam / 2], [1, 2, 3])[0], reference.pdr(V, [lam, lam, lam / 2], [1, 2, 3])[0])
        np.testing.assert_array_equal(oracle.yang3(V, lam)[0], reference.yang3(V, lam)[0])
