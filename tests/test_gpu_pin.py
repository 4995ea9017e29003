"""The pinning solver (pin.hip; geometry rung 3) on the GPU: exact 1-D prox of every fibre found data-parallel inside the
fibre -- the rung the policy climbs to when pieces are long.  Pinned to that rung and checked against the oracle over
fibre lengths around every group geometry, both sweep directions, weighted sweeps, and through the splitting loops."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def rung3(clib):
    before = clib.proxtv_set_option(b"chunk_mode", 3)
    yield
    clib.proxtv_set_option(b"chunk_mode", before)


def _families(rng, n):
    yield "noise", rng.standard_normal(n)
    yield "blocks", np.repeat(rng.standard_normal(n // 37 + 1), 37)[:n] + 0.2 * rng.standard_normal(n)
    yield "walk", np.cumsum(rng.standard_normal(n)) * 0.3
    yield "constant", np.full(n, 1.25)
    yield "offset", 1000.0 + rng.standard_normal(n)


def test_single_fibres_every_group_geometry(ptv, oracle, rung3):
    """One fibre per call: lengths on both sides of 64 x 16, 256 x 16, 256 x 32, 256 x 64 (beyond: the previous rung 3)."""
    rng = np.random.default_rng(90)
    for n in (96, 257, 1023, 1024, 1025, 4096, 4097, 8192, 8193, 16384, 16385):
        for name, x in _families(rng, n):
            for lam in (0.05, 1.0, 30.0):
                assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"{name} n={n} lam={lam}")


def test_batched_fibres_both_directions(oracle, rung3):
    import torch
    from proxtv_amd import device
    rng = np.random.default_rng(91)
    for shape in ((1500, 333), (333, 1500), (130, 97, 5), (4100, 70)):
        X = rng.standard_normal(shape) + np.repeat(rng.standard_normal((shape[0] // 50 + 1,) + shape[1:]), 50, axis=0)[:shape[0]]
        xd = device.to_colmajor(torch.from_numpy(X).cuda())
        for dim in range(len(shape)):
            for lam in (0.2, 4.0):
                got = device.tv1_fibres(xd, lam, dim).cpu().numpy()
                want = np.apply_along_axis(lambda v: oracle.tv1_hybrid(np.ascontiguousarray(v), lam), dim, X)
                assert_close(got, want, tol=1e-11, what=f"{shape} dim {dim} lam {lam}")


def test_weighted_sweeps_both_directions(oracle, rung3):
    """Per-edge penalties along dimension 0 and along a strided dimension (the penalties are transposed with the data)."""
    import torch
    from proxtv_amd import device
    rng = np.random.default_rng(96)
    X = np.cumsum(rng.standard_normal((700, 300)), axis=0) * 0.2 + rng.standard_normal((700, 300))
    xd = device.to_colmajor(torch.from_numpy(X).cuda())
    for dim in (0, 1):
        shape = list(X.shape)
        shape[dim] -= 1
        W = 10 ** rng.uniform(-1.5, 1.0) * rng.uniform(0.2, 1.0, shape)
        wd = device.to_colmajor(torch.from_numpy(W).cuda())
        got = device.tv1_fibres(xd, 0.0, dim, weights=wd).cpu().numpy()
        want = np.empty_like(X)
        for j in range(X.shape[1 - dim]):
            if dim == 0:
                want[:, j] = oracle.tv1_weighted(np.ascontiguousarray(X[:, j]), np.ascontiguousarray(W[:, j]))
            else:
                want[j, :] = oracle.tv1_weighted(np.ascontiguousarray(X[j, :]), np.ascontiguousarray(W[j, :]))
        assert_close(got, want, tol=1e-11, what=f"weighted fibres dim {dim}")


def test_weighted_columns(ptv, oracle, rung3):
    rng = np.random.default_rng(92)
    for n in (300, 1024, 1025, 5000, 8192):
        x = np.cumsum(rng.standard_normal(n)) * 0.2 + rng.standard_normal(n)
        w = 10 ** rng.uniform(-1.5, 1.0) * rng.uniform(0.2, 1.0, n - 1)
        w[rng.integers(0, n - 1, 5)] = 0.0
        assert_close(ptv.tv1w_1d(x, w), oracle.tv1_weighted(x, w), tol=1e-11, what=f"weighted n={n}")


def test_splitting_loops_on_rung3(ptv, oracle, rung3):
    rng = np.random.default_rng(93)
    X = rng.standard_normal((500, 620))
    for lam in (0.3, 3.0):
        assert_close(ptv.tv1_2d(X, lam), oracle.dr2(X, lam)[0], tol=1e-9, what=f"DR lam={lam}")
        assert_close(ptv.tv1_2d(X, lam, method="pd"), oracle.pd2(X, [lam, lam], [1, 2])[0], tol=1e-9, what=f"PD2 lam={lam}")
        assert_close(ptv.tv1_2d(X, lam, method="yang"), oracle.yang2(X, lam)[0], tol=1e-9, what=f"Yang lam={lam}")
    W1, W2 = rng.uniform(0.5, 3.0, (499, 620)), rng.uniform(0.5, 3.0, (500, 619))
    assert_close(ptv.tv1w_2d(X, W1, W2), oracle.dr2w(X, W1, W2)[0], tol=1e-9, what="weighted DR")
    V = rng.standard_normal((120, 110, 100))
    assert_close(ptv.tvgen(V, [2.0, 1.0, 3.0], [1, 2, 3], [1, 1, 1]), oracle.pd(V, [2.0, 1.0, 3.0], [1, 2, 3])[0], tol=1e-9, what="PD 3-D")


def test_policy_climbs_to_the_pinning_rung_and_back(ptv, clib, oracle):
    """Adaptive policy (deterministic = 0): long pieces (lambda = 3 on unit noise) end on rung 3; white noise at small lambda
    returns to the chunk kernels.  The default policy takes the same rungs straight from the input's statistics."""
    before = clib.proxtv_set_option(b"chunk_mode", -1)
    det = clib.proxtv_set_option(b"deterministic", 1)
    try:
        X = np.random.default_rng(94).standard_normal((900, 1100))
        assert_close(ptv.tv1_2d(X, 3.0), oracle.dr2(X, 3.0)[0], tol=1e-9, what="DR lam=3, seeded")
        assert clib.proxtv_chunk_mode() == 3, clib.proxtv_chunk_mode()
        assert_close(ptv.tv1_2d(X, 0.05), oracle.dr2(X, 0.05)[0], tol=1e-9, what="DR lam=0.05, seeded")
        assert clib.proxtv_chunk_mode() == 0, clib.proxtv_chunk_mode()
        clib.proxtv_set_option(b"deterministic", 0)
        rng = np.random.default_rng(94)
        X = rng.standard_normal((900, 1100))
        want = oracle.dr2(X, 3.0)[0]
        for _ in range(3):
            got = ptv.tv1_2d(X, 3.0)
        assert_close(got, want, tol=1e-9, what="DR lam=3, adaptive")
        assert clib.proxtv_chunk_mode() == 3, clib.proxtv_chunk_mode()
        for _ in range(8):   # (a rejected direction is left alone for a couple of solves)
            got = ptv.tv1_2d(X, 0.05)
            if clib.proxtv_chunk_mode() <= 1:
                break
        assert_close(got, oracle.dr2(X, 0.05)[0], tol=1e-9, what="DR lam=0.05, adaptive")
        assert clib.proxtv_chunk_mode() <= 1, clib.proxtv_chunk_mode()
    finally:
        clib.proxtv_set_option(b"chunk_mode", before)
        clib.proxtv_set_option(b"deterministic", det)


def test_previous_rung3_still_exact(ptv, clib, oracle, rung3):
    """option pin = 0: rung 3 is the global-memory chunk kernel again (what fibres too long for the LDS plane get)."""
    before = clib.proxtv_set_option(b"pin", 0)
    try:
        rng = np.random.default_rng(95)
        x = np.repeat(rng.standard_normal(100), 50) + 0.2 * rng.standard_normal(5000)
        for lam in (0.5, 5.0):
            assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"gchunk lam={lam}")
        X = rng.standard_normal((400, 500))
        assert_close(ptv.tv1_2d(X, 1.0), oracle.dr2(X, 1.0)[0], tol=1e-9, what="DR on the old rung 3")
    finally:
        clib.proxtv_set_option(b"pin", before)


def test_fibres_longer_than_a_workgroup(ptv, oracle, rung3):
    """Beyond 16384 samples the fibre is spread over a grid of workgroups (pinlong.hip): the BASELINE config-#1 shape with
    pieces of any length."""
    import torch
    from proxtv_amd import device
    rng = np.random.default_rng(97)
    for n in (16385, 20000, 100000, 1000000):
        x = rng.standard_normal(n) + np.repeat(rng.standard_normal(n // 3000 + 1), 3000)[:n]
        for lam in (0.5, 30.0, 2000.0):
            assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"n={n} lam={lam}")
    for n in (8193, 50000):
        x = np.cumsum(rng.standard_normal(n)) * 0.1 + rng.standard_normal(n)
        w = rng.uniform(0.05, 4.0, n - 1)
        assert_close(ptv.tv1w_1d(x, w), oracle.tv1_weighted(x, w), tol=1e-11, what=f"weighted n={n}")
    # several long fibres in one launch, contiguous and strided
    X = rng.standard_normal((40000, 3)) + np.repeat(rng.standard_normal((20, 3)), 2000, axis=0)
    for arr, dim in ((X, 0), (np.ascontiguousarray(X.T), 1)):
        xd = device.to_colmajor(torch.from_numpy(arr).cuda())
        got = device.tv1_fibres(xd, 7.0, dim).cpu().numpy()
        want = np.apply_along_axis(lambda v: oracle.tv1_hybrid(np.ascontiguousarray(v), 7.0), dim, arr)
        assert_close(got, want, tol=1e-11, what=f"3 fibres of 40000, dim {dim}")


def _periodic_families(n):
    k = np.arange(n)
    yield "zigzag", np.where(k % 2 == 0, 1.0, -1.0)             # every sample a bend: n / 2 levels without a cap
    yield "period3", np.array([0.0, 2.0, -1.0])[k % 3]
    yield "sawtooth7", (k % 7).astype(float)
    yield "stripes", np.repeat(np.where(np.arange(n // 4 + 1) % 2 == 0, 3.0, -3.0), 4)[:n]


def test_level_cap_hands_periodic_fibres_to_the_walker(ptv, clib, oracle, rung3):
    """Data with exact periodic ties peel one knot per segment end and level; the kernels give such a fibre up after
    kPinMaxLevels = 64 levels and a gated sequential sweep finishes it (pin.hpp).  Exact either way, and bounded."""
    for n in (1000, 4096, 16384):
        for name, x in _periodic_families(n):
            for lam in (0.1, 0.6):
                before = clib.proxtv_debug_counter(b"pin_sweeps")
                got = ptv.tv1_1d(x, lam)
                assert_close(got, oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"{name} n={n} lam={lam}")
                # the pinning rung took the sweep (and gave the fibre up after kPinMaxLevels levels, whatever the data: the bound on
                # its cost is structural -- uncapped, these data took minutes -- so no wall clock is asserted here)
                assert clib.proxtv_debug_counter(b"pin_sweeps") - before == 1, (name, n, lam)


def test_level_cap_batched_and_strided(oracle, rung3):
    """A batch in which only some fibres are periodic (those alone go to the walker), both sweep directions, and the
    fused DR operators through a whole solve on a striped image."""
    import torch
    import proxtv_amd
    from proxtv_amd import device
    rng = np.random.default_rng(98)
    X = rng.standard_normal((2048, 96)) + np.repeat(rng.standard_normal((32, 96)), 64, axis=0)
    X[:, ::5] = np.where(np.arange(2048) % 2 == 0, 1.0, -1.0)[:, None]
    for arr, dim in ((X, 0), (np.ascontiguousarray(X.T), 1)):
        xd = device.to_colmajor(torch.from_numpy(arr).cuda())
        got = device.tv1_fibres(xd, 0.3, dim).cpu().numpy()
        want = np.apply_along_axis(lambda v: oracle.tv1_hybrid(np.ascontiguousarray(v), 0.3), dim, arr)
        assert_close(got, want, tol=1e-11, what=f"mixed batch, dim {dim}")
    S = np.where((np.arange(600)[:, None] + np.arange(500)[None, :]) % 2 == 0, 1.0, -1.0)   # checkerboard
    assert_close(proxtv_amd.tv1_2d(S, 0.2), oracle.dr2(S, 0.2)[0], tol=1e-9, what="DR on a checkerboard, rung 3")


def test_level_cap_long_fibre_takes_the_next_rung(ptv, clib, oracle, rung3):
    """Beyond one workgroup (pinlong.hip) a capped sweep writes nothing and the global-memory chunk kernels take it."""
    n = 70000
    handed_on = 0
    for name, x in _periodic_families(n):
        before = clib.proxtv_debug_counter(b"pin_cap_next_rung")
        got = ptv.tv1_1d(x, 0.4)
        assert_close(got, oracle.tv1_hybrid(x, 0.4), tol=1e-11, what=f"{name} n={n}")
        handed_on += clib.proxtv_debug_counter(b"pin_cap_next_rung") - before
    # (which families exhaust the 64 levels is the data's business; the zigzag -- every sample a bend -- surely does.  What bounds the
    # cost is that the capped sweep is handed on at all, not a wall clock: a shared or cold device must not turn this suite red.)
    assert handed_on >= 1


@pytest.mark.parametrize("seed", [2, 1, 0])
def test_a_priori_pins(ptv, clib, oracle, rung3, seed):
    """The switch of the pinning rung against the oracle: pin_seed (the levels start from the knots known a priori -- |dy| > 4 lambda,
    weighted r_{j+1} + 2 r_j + r_{j-1} -- instead of the fibre ends alone).  Tall images (4096+ strided fibres); lambdas with many,
    few and no seeds; weighted, both forms of the DR iteration, PD2."""
    rng = np.random.default_rng(300 + 2 * seed)
    before = (clib.proxtv_set_option(b"pin_seed", seed),)
    try:
        for (M, N), lam in (((4100, 300), 0.6), ((4200, 130), 0.15), ((4128, 97), 2.5)):
            X = rng.standard_normal((M, N))
            assert_close(ptv.tv1_2d(X, lam, max_iters=5), oracle.dr2(X, lam, max_iters=5)[0], tol=1e-9, what=f"dr2 {M}x{N} lam {lam}")
            W1, W2 = rng.uniform(0.3 * lam, 1.7 * lam, (M - 1, N)), rng.uniform(0.3 * lam, 1.7 * lam, (M, N - 1))
            assert_close(ptv.tv1w_2d(X, W1, W2, max_iters=4), oracle.dr2w(X, W1, W2, max_iters=4)[0], tol=1e-9, what=f"dr2w {M}x{N} lam {lam}")
            assert_close(ptv.tv1_2d(X, lam, method="pd", max_iters=3), oracle.pd2(X, [lam, lam], [1, 2], max_iters=3)[0], tol=1e-9,
                         what=f"pd2 {M}x{N} lam {lam}")
        for n in (96, 1025, 4097, 9000):       # single fibres of every group geometry, seeds on
            for name, x in _families(rng, n):
                for lam in (0.05, 0.4):
                    assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"{name} n={n} lam={lam}")
    finally:
        clib.proxtv_set_option(b"pin_seed", before[0])


def test_knots_known_by_windows(ptv, clib, oracle, rung3):
    """pin_seed = 2 (the default): the levels also start from the deepest knots of windows of 4 / 16 / 64 knots (pincore.hpp: seed_phase1 /
    seed_phase2; fibres of up to 4096 samples, one penalty).  Images whose fibres fill the 256-lane group, half of it, a wave and less, with
    lengths on and off the lane / window grids; penalties around the noise level, where no jump reaches 4 lambda and windows find a third
    of the string's knots; every outer loop; the result may not depend on the switch beyond the rounding of the running sums."""
    rng = np.random.default_rng(310)
    assert clib.proxtv_set_option(b"pin_seed", 2) in (0, 1, 2)
    try:
        for (M, N), lam in (((4096, 150), 1.0), ((4095, 70), 0.8), ((2048, 300), 1.3), ((1041, 200), 0.7), ((1024, 256), 1.0), ((1000, 333), 2.0),
                            ((300, 4096), 1.0), ((80, 500), 0.9)):
            X = rng.standard_normal((M, N))
            want = oracle.dr2(X, lam, max_iters=4)[0]
            got = ptv.tv1_2d(X, lam, max_iters=4)
            assert_close(got, want, tol=1e-9, what=f"dr2 {M}x{N} lam {lam}")
            clib.proxtv_set_option(b"pin_seed", 1)
            plain = ptv.tv1_2d(X, lam, max_iters=4)
            clib.proxtv_set_option(b"pin_seed", 2)
            assert np.abs(got - plain).max() <= 1e-10, (M, N, lam, np.abs(got - plain).max())
            assert_close(ptv.tv1_2d(X, lam, method="pd", max_iters=3), oracle.pd2(X, [lam, lam], [1, 2], max_iters=3)[0], tol=1e-9,
                         what=f"pd2 {M}x{N} lam {lam}")
        for n in (17, 64, 96, 1000, 1025, 2048, 4000, 4096):
            for name, x in _families(rng, n):
                for lam in (0.05, 0.4, 1.0, 3.0):
                    assert_close(ptv.tv1_1d(x, lam), oracle.tv1_hybrid(x, lam), tol=1e-11, what=f"{name} n={n} lam={lam}")
        # quarter-integer samples: depths that sit exactly on the threshold
        for n, lam in ((4096, 0.5), (4096, 1.0), (1024, 0.25), (2048, 2.0)):
            X = rng.integers(-8, 9, (n, 64)) / 4.0
            assert_close(ptv.tv1_2d(X, lam, max_iters=3), oracle.dr2(X, lam, max_iters=3)[0], tol=1e-9, what=f"ties {n} lam {lam}")
    finally:
        clib.proxtv_set_option(b"pin_seed", 2)


def test_knots_known_by_windows_weighted(ptv, clib, oracle, rung3):
    """The windows on weighted fibres (per-edge penalties: each wall against the line through its own ends, a window's threshold the tube's
    width at its wider end).  Weighted DR on images whose fibres fill the group geometries of sixteen knots a lane, penalties around the noise
    level; free edges (penalty 0); weighted single fibres; the switch may not move the result beyond the rounding of the running sums."""
    rng = np.random.default_rng(330)
    assert clib.proxtv_set_option(b"pin_seed", 2) in (0, 1, 2)
    try:
        for (M, N), lam in (((4096, 60), 1.0), ((2048, 200), 1.5), ((1000, 300), 0.8), ((300, 4096), 1.0), ((4095, 33), 2.5)):
            X = rng.standard_normal((M, N))
            W1, W2 = rng.uniform(0.4 * lam, 1.6 * lam, (M - 1, N)), rng.uniform(0.4 * lam, 1.6 * lam, (M, N - 1))
            if M == 1000:
                W1[rng.integers(0, M - 1, 50), rng.integers(0, N, 50)] = 0.0
                W2[rng.integers(0, M, 50), rng.integers(0, N - 1, 50)] = 0.0
            before = clib.proxtv_debug_counter(b"pin_sweeps")
            got = ptv.tv1w_2d(X, W1, W2, max_iters=4)
            assert clib.proxtv_debug_counter(b"pin_sweeps") > before
            assert_close(got, oracle.dr2w(X, W1, W2, max_iters=4)[0], tol=1e-9, what=f"dr2w {M}x{N} lam {lam}")
            clib.proxtv_set_option(b"pin_seed", 1)
            plain = ptv.tv1w_2d(X, W1, W2, max_iters=4)
            clib.proxtv_set_option(b"pin_seed", 2)
            assert np.abs(got - plain).max() <= 1e-10, (M, N, lam, np.abs(got - plain).max())
        for n in (64, 1000, 2048, 4096):
            for name, x in _families(rng, n):
                for lam in (0.3, 1.0, 3.0):
                    w = rng.uniform(0.3 * lam, 1.7 * lam, n - 1)
                    assert_close(ptv.tv1w_1d(x, w), oracle.tv1_weighted(x, w), tol=1e-11, what=f"weighted {name} n={n} lam={lam}")
    finally:
        clib.proxtv_set_option(b"pin_seed", 2)
