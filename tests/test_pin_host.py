"""The pinning solver of long pieces (proxtv_amd/csrc/pincore.hpp: "pin the worst violator of every segment, on both walls,
until every chord fits its tube"), compiled for the host with its group of lanes emulated one after the other, against
the oracle -- no GPU needed.  Exact for every input by construction; the tolerance is that of its running sums."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_pin_"), "libpin_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_pin_fibre.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]
    lib.host_pin_fibre.restype = C.c_int
    lib.host_pin_fibre_seeded.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]
    lib.host_pin_fibre_seeded.restype = C.c_int
    lib.host_pin_fibre_windows.argtypes = [C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_void_p]
    lib.host_pin_fibre_windows.restype = C.c_int
    lib.host_pin_fibre_windows_weighted.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    lib.host_pin_fibre_windows_weighted.restype = C.c_int
    lib.host_pin_fibre_long.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
    lib.host_pin_fibre_long.restype = C.c_int
    lib.host_pin_fibre_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int]
    lib.host_pin_fibre_threads.restype = C.c_int
    return lib


def pin(lib, y, lam, w=None, P=16):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.full(y.size, np.nan)
    levels = lib.host_pin_fibre(y.ctypes.data, None if w is None else w.ctypes.data, lam, x.ctypes.data, y.size, P)
    assert levels >= 1
    return x, levels


def pin_seeded(lib, y, lam, w=None, P=16):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.full(y.size, np.nan)
    levels = lib.host_pin_fibre_seeded(y.ctypes.data, None if w is None else w.ctypes.data, lam, x.ctypes.data, y.size, P)
    assert levels >= 1
    return x, levels


def pin_windows(lib, y, lam):
    """seeded with the knots known by windows as well (the kernel's pin_seed = 2): values, levels, knots the levels started from"""
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.full(y.size, np.nan)
    seeds = C.c_int(0)
    levels = lib.host_pin_fibre_windows(y.ctypes.data, lam, x.ctypes.data, y.size, C.byref(seeds))
    assert levels >= 1
    return x, levels, seeds.value


def tol(y):
    """running sums of n centred samples: 1e-16 * n * spread, relative to the data's scale"""
    y = np.asarray(y, dtype=float)
    return 4e-16 * y.size * max(1.0, float(np.abs(y - y.mean()).max())) + 1e-13 * max(1.0, float(np.abs(y).max()))


FAMILIES = {
    "noise": lambda r, n: r.standard_normal(n),
    "blocks": lambda r, n: np.repeat(r.standard_normal(n // 37 + 1), 37)[:n] + 0.2 * r.standard_normal(n),
    "integers": lambda r, n: r.integers(-2, 3, n).astype(float),
    "walk": lambda r, n: np.cumsum(r.standard_normal(n)) * 0.3,
    "ramp": lambda r, n: np.linspace(0, 10, n) + 0.1 * r.standard_normal(n),
    "sine": lambda r, n: 5 * np.sin(np.arange(n) / 40.0),
    "constant": lambda r, n: np.full(n, 3.0),
    "offset": lambda r, n: 1000.0 + r.standard_normal(n),
}


def test_pinning_equals_oracle(harness, oracle):
    rng = np.random.default_rng(0)
    worst_levels = 0
    for trial in range(400):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([1, 2, 3, 5, 16, 17, 31, 33, 64, 100, 257, 1000, 1025, 4096, 4097, 5000]))
        y = FAMILIES[name](rng, n)
        lam = float(10 ** rng.uniform(-2, 2))
        want = oracle.tv1_linearized(y.copy(), lam)
        for P in (16, 64) if trial % 3 else (4, 32):
            x, levels = pin(harness, y, lam, P=P)
            worst_levels = max(worst_levels, levels)
            assert np.abs(x - want).max() <= tol(y), (name, n, lam, P, np.abs(x - want).max())
    assert worst_levels <= 64, worst_levels


def test_pinning_weighted_equals_oracle(harness, oracle):
    rng = np.random.default_rng(1)
    for trial in range(200):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([2, 3, 17, 64, 257, 1000, 4096]))
        y = FAMILIES[name](rng, n)
        w = 10 ** rng.uniform(-2, 1.5) * rng.uniform(0.2, 1.0, n - 1)
        if trial % 5 == 0:
            w[rng.integers(0, n - 1, max(1, n // 10))] = 0.0     # free edges: the string is cut there
        want = oracle.tv1_weighted(y.copy(), w)
        x, _ = pin(harness, y, 0.0, w=w, P=16)
        assert np.abs(x - want).max() <= tol(y), (name, n, np.abs(x - want).max())


def test_levels_stay_logarithmic_on_the_workloads(harness):
    """What makes the solver a rung of the ladder: the level count barely moves between white noise, the long pieces of
    a DR solve at large lambda, and a single flat piece."""
    rng = np.random.default_rng(2)
    y = rng.standard_normal(4096)
    counts = {lam: pin(harness, y, lam)[1] for lam in (0.1, 1.0, 3.0, 10.0, 100.0)}
    assert max(counts.values()) <= 24, counts
    assert counts[100.0] <= 4, counts


def test_special_values(harness, oracle):
    for y, lam in (([3.0], 1.0), ([1.0, 5.0], 1.0), ([1.0, 5.0], 10.0), ([2.0, 2.0, 2.0, 2.0], 0.5),
                   ([1.0, 4.0, 2.0, 8.0, 3.0], 0.0), ([0.0, 10.0, 0.0, 10.0, 0.0], 2.0)):
        y = np.array(y)
        x, _ = pin(harness, y, lam)
        np.testing.assert_allclose(x, oracle.tv1_linearized(y.copy(), lam), atol=1e-13)


def test_ties_split_segments_in_the_middle(harness, oracle):
    """Exact ties (stripes, staircases): the lane nearest the middle of a segment wins the claim, so a segment of equal
    violations is halved, not peeled from one end.  (A regular zigzag whose every sample is a bend still peels -- the worst
    violator of a sloped chord is always next to a pin: n / 2 levels; such data have one-sample pieces and belong to the
    chunk kernels, which is where the policy sends them.)"""
    n = 4096
    y = 3.0 * (-1.0) ** np.arange(n)
    x, levels = pin(harness, y, 2.999)          # one flat piece per pair is not possible: the string is pinned everywhere ...
    assert np.abs(x - oracle.tv1_linearized(y.copy(), 2.999)).max() <= tol(y)
    x, levels = pin(harness, y, 3.0)            # ... at lambda = amplitude every knot ties
    assert levels <= 24, levels
    assert np.abs(x - oracle.tv1_linearized(y.copy(), 3.0)).max() <= tol(y)
    stairs = np.repeat(np.arange(n // 64), 64).astype(float)
    x, levels = pin(harness, stairs, 3.0)
    assert levels <= 32, levels
    assert np.abs(x - oracle.tv1_linearized(stairs.copy(), 3.0)).max() <= tol(stairs)


def test_long_fibres_wide_keys(harness, oracle):
    """Fibres beyond one workgroup's LDS (pinlong.hip: the lanes of a grid of workgroups, 64-bit claim keys): same protocol,
    emulated the same way."""
    rng = np.random.default_rng(6)
    for n, lam in ((70000, 0.5), (200000, 5.0), (1000000, 30.0), (300000, 1000.0)):
        y = rng.standard_normal(n) + np.repeat(rng.standard_normal(n // 5000 + 1), 5000)[:n]
        x = np.full(n, np.nan)
        levels = harness.host_pin_fibre_long(y.ctypes.data, None, lam, x.ctypes.data, n)
        assert 1 <= levels <= 40, levels
        assert np.abs(x - oracle.tv1_linearized(y.copy(), lam)).max() <= tol(y), (n, lam)
    n = 50000
    y = np.cumsum(rng.standard_normal(n)) * 0.1
    w = rng.uniform(0.1, 3.0, n - 1)
    x = np.full(n, np.nan)
    harness.host_pin_fibre_long(y.ctypes.data, w.ctypes.data, 0.0, x.ctypes.data, n)
    assert np.abs(x - oracle.tv1_weighted(y.copy(), w)).max() <= tol(y)


def test_protocol_under_real_concurrency(harness):
    """The slot protocol (one buffer: knots cleared in scan, maxima in update, three barriers per level) with the lanes on
    8 host threads, relaxed atomics and thread barriers: bit-identical to the sequential emulation, level for level."""
    rng = np.random.default_rng(8)
    for trial in range(40):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([100, 1000, 4096, 20000, 70000]))
        y = FAMILIES[name](rng, n)
        lam = float(10 ** rng.uniform(-1.5, 1.5))
        w = 10 ** rng.uniform(-1, 1) * rng.uniform(0.2, 1.0, n - 1) if trial % 5 == 0 else None
        a = np.full(n, np.nan)
        la = harness.host_pin_fibre_long(y.ctypes.data, None if w is None else w.ctypes.data, lam, a.ctypes.data, n)
        b = np.full(n, np.nan)
        lb = harness.host_pin_fibre_threads(y.ctypes.data, None if w is None else w.ctypes.data, lam, b.ctypes.data, n, 8)
        assert la == lb, (name, n, lam, la, lb)
        np.testing.assert_array_equal(a, b)


def test_seeded_pinning_equals_oracle_and_saves_levels(harness, oracle):
    """Starting from the knots known a priori (|dy| > 4 lambda, weighted: r_{j+1} + 2 r_j + r_{j-1}; PinLane::seed) the solver returns
    the same string -- those knots ARE on it -- in fewer levels wherever there are any."""
    rng = np.random.default_rng(11)
    saved = total = 0
    for trial in range(300):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([2, 3, 17, 33, 64, 257, 1000, 1025, 4096, 5000]))
        y = FAMILIES[name](rng, n)
        lam = float(10 ** rng.uniform(-2, 0.5))
        weighted = trial % 3 == 0
        if weighted:
            w = lam * rng.uniform(0.2, 1.8, n - 1)
            if trial % 6 == 0:
                w[rng.integers(0, n - 1, max(1, n // 10))] = 0.0
            want = oracle.tv1_weighted(y.copy(), w)
        else:
            w = None
            want = oracle.tv1_linearized(y.copy(), lam)
        for P in (16, 32) if trial % 2 else (4, 64):
            x, levels = pin_seeded(harness, y, lam, w=w, P=P)
            _, plain = pin(harness, y, lam, w=w, P=P)
            assert np.abs(x - want).max() <= tol(y), (name, n, lam, P, weighted, np.abs(x - want).max())
            assert levels <= plain + 1, (name, n, lam, P, levels, plain)
            saved += plain - levels
            total += plain
    assert saved > 0.1 * total, (saved, total)


def test_pinning_on_fibres_with_zero_jump_knots(harness, oracle):
    """Fibres built backwards from solutions whose string touches the tube to the last bit (knots with zero jump, rows inside pieces that
    touch a wall): the operands of late Dykstra / DR iterations, which the seeded policy sends to this solver from lambda ~ 0.75 on.
    Where the taut string is degenerate the pins may fall differently; the values may not."""
    from test_chunk_host import _zero_jump_fibre
    rng = np.random.default_rng(9)
    for trial in range(300):
        n = int(rng.choice([17, 64, 100, 257, 1000, 1025, 4096, 5000]))
        lam = float(rng.choice([0.05, 0.5, 3.0, 6.0]) * (0.5 + rng.random()))
        y, x_built = _zero_jump_fibre(rng, n, lam)
        want = oracle.tv1_linearized(y.copy(), lam)
        for P in (16, 64) if trial % 3 else (4, 32):
            x, levels = pin(harness, y, lam, P=P)
            assert np.abs(x - want).max() <= tol(y), (n, lam, P, np.abs(x - want).max())
            assert levels <= 64
            xs, _ = pin_seeded(harness, y, lam, P=P)
            assert np.abs(xs - want).max() <= tol(y), ("seeded", n, lam, P, np.abs(xs - want).max())


def _dr_operands(oracle, rng, n, lam, iterations):
    """Operands of the column / row sweeps of a DR solve on an n x n unit-noise image (oracle/tvnd_oracle.c: dr_generic), a few fibres each."""
    U = rng.standard_normal((n, n))
    t = np.full((n, n), 2 * U.mean())
    prox = lambda T: np.stack([oracle.tv1_linearized(np.ascontiguousarray(T[:, j]), lam) for j in range(n)], axis=1)
    out = []
    for it in range(1, iterations + 1):
        out += [np.ascontiguousarray(t[:, j]) for j in range(0, n, n // 3)]
        s = 2 * (t - prox(t)) - t
        v = U - s
        out += [np.ascontiguousarray(v[j, :]) for j in range(0, n, n // 3)]
        tb = 2 * (U - (v - prox(v.T.copy()).T)) - s
        t = 0.5 * (t + tb)
    return out


def test_window_seeds_equal_oracle(harness, oracle):
    """Knots known by windows (pincore.hpp: seed_phase1 / seed_phase2 -- the deepest knot of a window of 4 / 16 / 64 knots touches its
    wall when it lies more than the tube's width below the line through the wall at the window's ends): same string, every family,
    lengths around the lane (16), wave (1024) and window grids, penalties from a hundredth to a hundred times the noise."""
    rng = np.random.default_rng(21)
    for trial in range(500):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([2, 5, 16, 17, 18, 31, 33, 63, 64, 65, 80, 100, 257, 1000, 1023, 1024, 1025, 1041, 2048, 3000, 4095, 4096]))
        y = FAMILIES[name](rng, n)
        lam = float(10 ** rng.uniform(-2, 2))
        want = oracle.tv1_linearized(y.copy(), lam)
        x, levels, seeds = pin_windows(harness, y, lam)
        assert np.abs(x - want).max() <= tol(y), (name, n, lam, seeds, np.abs(x - want).max())
        assert levels <= 64


def test_window_seeds_on_ties_and_slivers(harness, oracle):
    """Where depths sit ON the threshold: quarter-integer samples with penalties that make 2 lambda W a sum of them (the rule is strict: a
    knot exactly the tube's width deep is not pinned by it -- and pinning it would be right as well), and fibres built backwards from
    strings that touch their tube to the last bit."""
    from test_chunk_host import _zero_jump_fibre
    rng = np.random.default_rng(22)
    for trial in range(300):
        n = int(rng.choice([64, 100, 257, 1000, 1025, 4096]))
        y = rng.integers(-8, 9, n) / 4.0
        lam = float(rng.choice([0.125, 0.25, 0.5, 0.75, 1.0, 1.5, 2.0, 4.0]))
        want = oracle.tv1_linearized(y.copy(), lam)
        x, _, _ = pin_windows(harness, y, lam)
        assert np.abs(x - want).max() <= tol(y), ("ties", n, lam, np.abs(x - want).max())
    for trial in range(200):
        n = int(rng.choice([64, 100, 257, 1000, 1025, 4096]))
        lam = float(rng.choice([0.05, 0.5, 3.0, 6.0]) * (0.5 + rng.random()))
        y, _ = _zero_jump_fibre(rng, n, lam)
        want = oracle.tv1_linearized(y.copy(), lam)
        x, _, _ = pin_windows(harness, y, lam)
        assert np.abs(x - want).max() <= tol(y), ("zero jumps", n, lam, np.abs(x - want).max())


def test_window_seeds_halve_the_levels_of_dr_operands(harness, oracle):
    """What they are for: the operands of DR sweeps at lambda ~ the noise level have no jump above 4 lambda to start from (10-12 levels);
    windows find a third to a half of the knots of the string before the first level."""
    rng = np.random.default_rng(23)
    plain = seeded = 0
    for y in _dr_operands(oracle, rng, 1024, 1.0, 4):
        want = oracle.tv1_linearized(y.copy(), 1.0)
        _, l1 = pin_seeded(harness, y, 1.0)
        x, l2, seeds = pin_windows(harness, y, 1.0)
        assert np.abs(x - want).max() <= tol(y)
        pieces = 1 + np.count_nonzero(np.abs(np.diff(want)) > 1e-12)
        assert seeds <= pieces - 1 + 8, (seeds, pieces)     # (seeds are knots of the string; a few touch a wall without a bend)
        plain += l1
        seeded += l2
    assert seeded <= 0.62 * plain, (seeded, plain)


def test_window_seeds_with_per_edge_penalties(harness, oracle):
    """The same windows on weighted fibres: the walls are no copies of the sums any more (upper wall S + r, lower S - r, each against the line
    through its own ends), a window's threshold is the tube's width at its wider end.  Penalties spread over a decade, free edges (r = 0: the
    string is cut there), penalties around the noise level where the windows find most."""
    rng = np.random.default_rng(24)
    found = plain_levels = seeded_levels = 0
    for trial in range(400):
        name = list(FAMILIES)[trial % len(FAMILIES)]
        n = int(rng.choice([5, 17, 33, 64, 65, 100, 257, 1000, 1024, 1025, 2048, 4000, 4096]))
        y = FAMILIES[name](rng, n)
        w = 10 ** rng.uniform(-1.5, 1.0) * rng.uniform(0.3, 1.7, n - 1)
        if trial % 4 == 0:
            w[rng.integers(0, n - 1, max(1, n // 20))] = 0.0
        if trial % 7 == 0:
            w = np.full(n - 1, float(w[0]))          # one penalty, through the weighted code
        want = oracle.tv1_weighted(y.copy(), w)
        x = np.full(n, np.nan)
        seeds = C.c_int(0)
        levels = harness.host_pin_fibre_windows_weighted(y.ctypes.data, w.ctypes.data, x.ctypes.data, n, C.byref(seeds))
        assert np.abs(x - want).max() <= tol(y), (name, n, float(w.mean()), seeds.value, np.abs(x - want).max())
        _, l1 = pin_seeded(harness, y, 0.0, w=w)
        found += seeds.value
        plain_levels += l1
        seeded_levels += levels
    assert found > 0 and seeded_levels < 0.9 * plain_levels, (found, seeded_levels, plain_levels)
