"""Option "certify": behind every fibre sweep a kernel checks the optimality conditions of the 1-D prox on what the sweep wrote
(u = cumsum(y - x): |u_k| <= lambda_k, u_k = -+lambda_k where x steps up / down, u_{n-1} = 0 -- the conditions the reference's
solvers implement, /root/reference/src/TVL1opt.cpp:359-564), re-solves a fibre that fails and counts it.

Two things are shown on the GPU: clean solves pass (no false alarm, same bits as without the option, on every rung and every op
of the library), and a wrong sweep is caught and repaired -- the WRONG sweep being the real one of round 5: the along-fibre kernel's
rebuild with the semantics of rounds 1-4 (option debug_legacy_rebuild) on the fixture that exposed them."""
import os

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _counters(clib):
    return {k: clib.proxtv_debug_counter(k.encode()) for k in ("certify_sweeps", "certify_failures", "certify_skipped")}


@pytest.fixture()
def certify(clib):
    before = clib.proxtv_set_option(b"certify", 1)
    yield
    clib.proxtv_set_option(b"certify", before)


@pytest.mark.parametrize("rung", [-1, 0, 1, 3, 5])
def test_clean_solves_pass_and_keep_their_bits(ptv, clib, oracle, rung):
    """Every op of the library under the certifier, default policy and pinned rungs: no fibre fails, nothing changes."""
    rng = np.random.default_rng(700 + rung)
    before_mode = clib.proxtv_set_option(b"chunk_mode", rung)
    try:
        X = rng.standard_normal((520, 330))
        W1, W2 = rng.uniform(0.05, 0.6, (519, 330)), rng.uniform(0.05, 0.6, (520, 329))
        V = rng.standard_normal((130, 96, 20))
        x1 = np.cumsum(rng.standard_normal(3000)) * 0.3

        def everything():
            out = []
            for lam in (0.1, 0.7, 4.0):
                out.append(ptv.tv1_2d(X, lam))                      # DR_COL / DR_ROW or DR_COL_V / DR_ROW_V, the two FINAL ops
                out.append(ptv.tv1_2d(X, lam, method="pd"))         # PD2_A / PD2_B
            out.append(ptv.tv1w_2d(X, W1, W2))                      # the weighted ops, DRW_ROW_FINAL
            out.append(ptv.tv1_2d(X, 0.3, method="yang"))           # YANG
            out.append(ptv.tv1_2d(X, 0.3, method="kolmogorov", max_iters=12))   # gated sweeps
            out.append(ptv.tvgen(V, [0.3, 0.2, 0.6], [1, 2, 3], [1, 1, 1]))    # PROX along every dimension
            out.append(ptv.tv1_1d(x1, 0.8))
            out.append(ptv.tv1w_1d(x1, rng.uniform(0.1, 1.0, x1.size - 1)))
            return out

        state = rng.bit_generator.state
        plain = everything()
        rng.bit_generator.state = state
        c0 = _counters(clib)
        clib.proxtv_set_option(b"certify", 1)
        try:
            checked = everything()
        finally:
            clib.proxtv_set_option(b"certify", 0)
        c1 = _counters(clib)
        assert c1["certify_failures"] == c0["certify_failures"], (c0, c1)
        assert c1["certify_sweeps"] - c0["certify_sweeps"] > 400
        for a, b in zip(plain, checked):
            np.testing.assert_array_equal(a, b)
        assert_close(checked[0], oracle.dr2(X, 0.1)[0], tol=1e-9, what="dr2")
    finally:
        clib.proxtv_set_option(b"chunk_mode", before_mode)


def test_sweeps_that_cannot_be_checked_are_counted(ptv, clib, certify):
    """lambda = 0 (the identity) and in-place sweeps (an output IS the operand) have nothing to check against: skipped, and said so."""
    c0 = _counters(clib)
    x = np.random.default_rng(1).standard_normal(500)
    np.testing.assert_array_equal(ptv.tv1_1d(x, 0.0), x)
    c1 = _counters(clib)
    assert c1["certify_skipped"] > c0["certify_skipped"] and c1["certify_failures"] == c0["certify_failures"]


def test_the_round_5_hole_planted_again_is_caught_and_repaired(ptv, clib, oracle):
    """tests/golden/degenerate_knot_fibre.npz on pinned rung 0 with the rebuild semantics of rounds 1-4 planted (an unproven chunk values
    its first piece from its own first row): the sweep is wrong by 0.028 in two rows, as in round 5.  With the certifier on, the
    fibre fails the optimality conditions, is re-solved, and the result is the reference's; the failure is counted."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate_knot_fibre.npz"))
    y, lam, want = g["y"], float(g["lam"]), g["expected"]
    rng = np.random.default_rng(47)
    cols = np.asfortranarray(np.repeat(y[:, None], 96, axis=1))
    cols[:, 1::3] += 1e-3 * rng.standard_normal((y.size, 32))
    want_cols = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, cols)
    before = (clib.proxtv_set_option(b"chunk_mode", 0), clib.proxtv_set_option(b"debug_legacy_rebuild", 1))
    try:
        wrong = ptv.tvgen(cols, [lam], [1], [1])
        err = np.abs(wrong - want_cols)
        assert err.max() > 1e-3, "the planted semantics no longer reproduce round 5's wrong rows: the test has lost its subject"
        bad_cols = np.flatnonzero(err.max(axis=0) > 1e-9)
        c0 = _counters(clib)
        clib.proxtv_set_option(b"certify", 1)
        try:
            fixed = ptv.tvgen(cols, [lam], [1], [1])
            alone = ptv.tv1_1d(y, lam)
        finally:
            clib.proxtv_set_option(b"certify", 0)
        c1 = _counters(clib)
        assert_close(fixed, want_cols, tol=1e-12, what="columns, certified")
        assert_close(alone, want, tol=1e-12, what="the fibre alone, certified")
        assert c1["certify_failures"] - c0["certify_failures"] >= bad_cols.size >= 1, (c0, c1, bad_cols)
    finally:
        clib.proxtv_set_option(b"chunk_mode", before[0])
        clib.proxtv_set_option(b"debug_legacy_rebuild", before[1])
    # ... and with the semantics as they are, the same sweep passes the certifier untouched
    before = clib.proxtv_set_option(b"chunk_mode", 0)
    c0 = _counters(clib)
    clib.proxtv_set_option(b"certify", 1)
    try:
        assert_close(ptv.tvgen(cols, [lam], [1], [1]), want_cols, tol=1e-12, what="columns, as built")
    finally:
        clib.proxtv_set_option(b"certify", 0)
        clib.proxtv_set_option(b"chunk_mode", before)
    assert _counters(clib)["certify_failures"] == c0["certify_failures"]


def test_the_certificate_on_its_own(ptv, clib, oracle):
    """proxtv_certify_fibres_dev: the same conditions as an entry point of their own, for results that came from anywhere.  The prox
    passes along both dimensions of an image, weighted and unweighted; one sample nudged by 1e-8 (a hundred thousand times below
    round 5's error, a hundred times below the parity bar) fails exactly its fibre; another penalty fails (all but) every fibre."""
    import torch
    from proxtv_amd import device
    rng = np.random.default_rng(3)
    X = device.to_colmajor(torch.from_numpy(rng.standard_normal((1500, 700))).cuda())
    for dim, lam in ((0, 0.4), (1, 0.4), (0, 3.0), (1, 0.02)):
        Y = device.tv1_fibres(X, lam, dim)
        assert device.certify_fibres(X, Y, lam, dim) == 0
        Z = Y.clone()
        Z[777, 333] += 1e-8
        assert device.certify_fibres(X, Z, lam, dim) == 1
        Z[5, 600] -= 1e-8
        assert device.certify_fibres(X, Z, lam, dim) == 2
        assert device.certify_fibres(X, Y, 1.5 * lam, dim) > 0.9 * X.shape[1 - dim]
    wshape = (1499, 700)
    W = device.to_colmajor(torch.from_numpy(rng.uniform(0.05, 0.8, wshape)).cuda())
    Y = device.tv1_fibres(X, 0.0, 0, weights=W)
    assert device.certify_fibres(X, Y, 0.0, 0, weights=W) == 0
    Y[1000, 0] += 1e-8
    assert device.certify_fibres(X, Y, 0.0, 0, weights=W) == 1
    assert device.certify_fibres(X, X, 0.4, 0) == -1                       # nothing to check against
    # ... and it agrees with the CPU oracle about what the prox IS (a column taken out and solved on the host)
    col = X[:, 17].cpu().numpy()
    got = device.tv1_fibres(X, 0.4, 0)[:, 17].cpu().numpy()
    assert_close(got, oracle.tv1_hybrid(np.ascontiguousarray(col), 0.4), tol=1e-12, what="column 17")
