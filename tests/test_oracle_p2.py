"""TV-L2 (p = 2) fibres on the CPU: the oracle's exact solver (orc_TV2_exact and the splitting loops that call it)
against optimality conditions, and against what the compiled reference returns (tests/golden/golden_p2.npz, generated
single-threaded by oracle/gen_golden.py) -- within the reference's OWN accuracy: morePG_TV2 stops at a duality gap of
STOP_MS = 1e-5 (src/TVopt.h:36), the objective is 1-strongly convex, so its x is guaranteed only to
||x_ref - x*||_2 <= sqrt(2 * 1e-5) = 4.5e-3 per fibre.  That, not 1e-6, is the bar here; it is the one place the 1e-6
parity bar cannot apply because the reference itself is not converged."""
import numpy as np
import pytest

from conftest import load_golden

REF_FIBRE_BOUND = np.sqrt(2 * 1e-5)


@pytest.fixture(scope="module")
def gp2():
    return load_golden("golden_p2.npz")


def objective(x, y, lam):
    return 0.5 * np.sum((x - y) ** 2) + lam * np.sqrt(np.sum(np.diff(x) ** 2))


def kkt_gap(x, y, lam):
    """Duality gap of x for the TV-L2 prox: lambda ||Dx|| = max over ||u|| <= lambda of u'Dx gives the dual
    D(u) = u'Dy - 1/2 ||D'u||^2 with x = y - D'u, so the dual point is read off x itself (u = cumsum(x - y)), scaled
    back into the ball if need be; gap = P(x) - D(u) >= 0, zero at the minimiser."""
    if x.size < 2:
        return float(np.max(np.abs(x - y), initial=0.0))
    u = np.cumsum(x - y)[:-1]
    nu = np.sqrt(np.sum(u * u))
    if nu > lam:
        u = u * (lam / nu)
    dtu = np.concatenate(([u[0]], np.diff(u), [-u[-1]]))          # D'u
    dual = -0.5 * np.sum(dtu ** 2) + np.sum(u * np.diff(y))
    return float(objective(x, y, lam) - dual), float(max(nu - lam, 0.0))


def test_exact_tv2_satisfies_optimality(oracle):
    rng = np.random.default_rng(5)
    for t in range(300):
        n = int(rng.integers(1, 600))
        y = rng.standard_normal(n) * float(rng.choice([1e-3, 1.0, 50.0]))
        lam = float(abs(rng.standard_normal()) * rng.choice([0.0, 0.01, 1.0, 30.0, 1e4]))
        x, info = oracle.tv(y, lam, 2)
        assert info[2] == 0
        scale = max(1.0, objective(y * 0 + y.mean(), y, lam))
        if n >= 2 and lam > 0:
            gap, infeas = kkt_gap(x, y, lam)
            assert gap <= 1e-11 * scale and infeas <= 1e-12 * max(lam, 1.0), (n, lam, gap, infeas)
        else:
            np.testing.assert_array_equal(x, y)
        assert abs(x.mean() - y.mean()) <= 1e-12 * max(1.0, abs(y).max())


def test_exact_tv2_vs_reference_within_its_accuracy(oracle, gp2):
    worst = 0.0
    for name in gp2["names1"]:
        x, lam = gp2[f"{name}/x"], float(gp2[f"{name}/lam"])
        got, _ = oracle.tv(x, lam, 2)
        if x.size == 1:   # the reference fails here (malloc(0) -> "out of memory", x untouched); the prox is the identity
            np.testing.assert_array_equal(got, x)
            continue
        ref = gp2[f"{name}/tv2"]
        err = np.sqrt(np.sum((got - ref) ** 2))
        worst = max(worst, err)
        assert err <= REF_FIBRE_BOUND, (name, err)
        assert objective(got, x, lam) <= objective(ref, x, lam) + 1e-12 * max(1.0, objective(ref, x, lam)), name
    assert worst > 1e-9          # (if this fails the reference became exact and the bar can be tightened)


def test_p2_loops_vs_reference_within_its_accuracy(oracle, gp2):
    """DR2 / PD2 / PD with TV-L2 fibres: same loops, exact fibre prox.  The per-fibre error of the reference passes
    through non-expansive steps, so whole-image differences stay of the order of the fibre bound (asserted loosely: a
    few times the bound in max-norm; typical values are printed by oracle/gen_golden.py's companion numbers in DESIGN.md)."""
    for name in gp2["names2"]:
        X, lam = gp2[f"{name}/X"], float(gp2[f"{name}/lam"])
        for n1, n2 in ((2, 2), (1, 2), (2, 1)):
            got = oracle.dr2(X, lam, 0.7 * lam, norm1=n1, norm2=n2)[0]
            assert np.max(np.abs(got - gp2[f"{name}/dr2_{n1}{n2}"])) <= 4 * REF_FIBRE_BOUND, (name, n1, n2)
            got, info, rc, _ = oracle.pd2(X, [lam, 0.7 * lam], [1, 2], norms=[n1, n2])
            assert np.max(np.abs(got - gp2[f"{name}/pd2_{n1}{n2}"])) <= 4 * REF_FIBRE_BOUND, (name, n1, n2)
            assert info[0] == gp2[f"{name}/pd2_{n1}{n2}_info"][0]
    got, info, rc, _ = oracle.pd(gp2["vol/X"], [0.3, 0.2, 0.4], [1, 2, 3], norms=[2, 1, 2])
    assert np.max(np.abs(got - gp2["vol/pd_212"])) <= 4 * REF_FIBRE_BOUND
    assert info[0] == gp2["vol/pd_212_info"][0]
