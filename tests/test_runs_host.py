"""The known-runs path of the along-fibre kernel on the host (proxtv_amd/csrc/chunkcore.hpp "known runs"; tests/host_harness.cpp:
host_runs_fibre mirrors the four phases of sweep_along_kernel RUNS with the lanes of a wave emulated one after the other): interior
segments of a fibre are cut at the bends known a priori (|dy| > 4 lambda); runs of one and two samples are settled by rule, longer
runs are walked one per lane from their first bend to the closing one; the rebuild values the pieces.

What is checked against the oracle: the result wherever a segment was solved that way (and that such segments exist in numbers on
the data the policy sends there), every row written exactly once, and that what does not fit -- long runs, more than 64 runs in a
segment, no known bend at the segment's ends -- falls back instead of going wrong.  Twice, like the chunk tests: IEEE quotients and
the device's table reciprocals."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

from test_chunk_host import _zero_jump_fibre

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["quotients", "table reciprocals"])
def harness(request):
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_runs_"), "libchunk_host.so")
    flags = ["-DPTV_TABLE_RECIP"] if request.param == "table reciprocals" else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", *flags, "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_runs_fibre.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p]
    return lib


def runs(lib, y, lam):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.full(y.size, np.nan)
    stats = np.zeros(10, dtype=np.int32)
    rc = lib.host_runs_fibre(y.ctypes.data, lam, y.size, x.ctypes.data, stats.ctypes.data)
    assert rc == 0, f"{-rc} rows of a solved segment not written exactly once"
    return x, stats


def test_unit_noise_at_the_headlines_penalty(harness, oracle):
    """lambda = 0.1 on N(0, 1): 78 % of the edges are bends known a priori -- (all but) every interior segment is solved run by run, ~44 runs of
    three and more samples each, and the result is the prox."""
    rng = np.random.default_rng(1)
    solved = fell = walked = 0
    for t in range(40):
        n = int(rng.choice([4096, 2300, 5000, 1200, 3265]))
        y = rng.standard_normal(n)
        x, st = runs(harness, y, 0.1)
        truth = oracle.tv1_linearized(y, 0.1)
        assert np.max(np.abs(x - truth)) <= 1e-13 * max(1.0, np.max(np.abs(y))), (t, n)
        solved, fell, walked = solved + st[0], fell + st[1], walked + st[2]
        assert st[3] <= 64
    assert solved >= 80 and fell <= 0.05 * solved     # (no bend known within two samples of a segment's start: 0.22^3 = 1 % of them)
    assert 35 * solved < walked < 55 * solved


def test_every_penalty_and_family_is_exact_or_falls_back(harness, oracle):
    """Scales, ties, blocks, walks; penalties from "every edge is known" to "none is": whatever a segment decides, the rows are right."""
    rng = np.random.default_rng(2)
    solved = fell = 0
    for t in range(300):
        n = int(rng.integers(1100, 4500))
        kind = int(rng.integers(0, 6))
        if kind == 0:   y = rng.standard_normal(n)
        elif kind == 1: y = np.repeat(rng.standard_normal(n // 5 + 1), 5)[:n] + 0.05 * rng.standard_normal(n)
        elif kind == 2: y = rng.integers(-2, 3, n).astype(float)                       # exact ties, jumps of exactly 4 lambda at lambda = 0.25
        elif kind == 3: y = np.cumsum(rng.standard_normal(n)) * 0.3
        elif kind == 4: y = rng.standard_normal(n) * rng.choice([1e-3, 1.0, 1e3])
        else:           y = np.round(rng.standard_normal(n) * 3) * 0.5
        lam = float(rng.choice([0.01, 0.05, 0.1, 0.15, 0.25, 0.5, 2.0]) * rng.choice([1.0, 1.0, 0.731]))
        x, st = runs(harness, y, lam)
        truth = oracle.tv1_linearized(y, lam)
        assert np.max(np.abs(x - truth)) <= 1e-13 * max(1.0, np.max(np.abs(y))), (t, kind, lam, st)
        solved, fell = solved + st[0], fell + st[1]
    assert solved > 150 and fell > 50      # (both ways of leaving a segment are exercised)


def test_late_iterates_full_of_zero_jump_knots(harness, oracle):
    """Operands as the late iterations of a Dykstra / DR loop make them (test_chunk_host._zero_jump_fibre): the string touches the tube to
    the last bit at knots whose jump is zero.  The rule for two-sample runs and the walk may cut such a fibre differently; both cuts are
    the prox to rounding."""
    rng = np.random.default_rng(3)
    solved = 0
    for t in range(200):
        n = int(rng.integers(1150, 3500))
        lam = float(rng.choice([0.02, 0.05, 0.1]) * (0.5 + rng.random()))
        y, x_true = _zero_jump_fibre(rng, n, lam)
        y = y + (rng.standard_normal(n) * (rng.random(n) < 0.7))     # (most edges known a priori, the built-in ties in between)
        x, st = runs(harness, y, lam)
        truth = oracle.tv1_linearized(y, lam)
        assert np.max(np.abs(x - truth)) <= 1e-12 * max(1.0, np.max(np.abs(y))), (t, lam, st)
        solved += st[0]
    assert solved > 50


def test_runs_that_cross_segment_boundaries(harness, oracle):
    """A flat stretch laid across every segment boundary (samples 1080 .. 1095 and so on): the run that comes in from before a segment
    starts further back than lane 0 looks, the one that leaves it ends further ahead than the look-ahead rows -- those segments fall
    back; with the stretch one or two samples long they are solved, by rule or by lane 0's walk from before the segment."""
    rng = np.random.default_rng(4)
    for width, want_solved in ((16, False), (2, True), (3, True), (1, True)):
        for t in range(20):
            y = rng.standard_normal(4500) * 2.0
            for s in range(1088, 4400, 1088):
                y[s - width // 2 - 1: s - width // 2 - 1 + width + 1] = y[s] + 0.01 * rng.standard_normal(width + 1)   # width + 1 samples = width flat edges
            x, st = runs(harness, y, 0.1)
            truth = oracle.tv1_linearized(y, 0.1)
            assert np.max(np.abs(x - truth)) <= 1e-13 * np.max(np.abs(y)), (width, t, st)
            if want_solved: assert st[0] >= 2, (width, st)
            else:           assert st[1] >= 2, (width, st)


def test_edges_within_a_hair_of_four_lambda(harness, oracle):
    """Round 6, found by the certifier in a full-size PD2 solve (tests/golden/sliver_edge_fibre.npz): the inner edge of a two-sample run
    that jumps by -4.00000006 lambda between two FLOOR bends must bend CEIL.  With certain_bend_before's threshold of 4.0000001 lambda
    in the edge masks it was neither a bend known a priori nor covered by the rule for two-sample runs ("an edge not known a priori
    jumps by 4 lambda at most"): 3e-9 off in two rows.  The masks cut at 4 lambda exactly.  The captured fibre, and planted ones: inner
    edges at 4 lambda (1 +- a few 1e-8, and exactly) in every combination of bend types around them."""
    g = np.load(os.path.join(HERE, "golden", "sliver_edge_fibre.npz"))
    y, lam = np.ascontiguousarray(g["y"]), float(g["lam"])
    x, st = runs(harness, y, lam)
    assert st[0] >= 3
    assert np.max(np.abs(x - g["expected"])) <= 1e-13
    rng = np.random.default_rng(5)
    for t in range(60):
        lam = float(rng.choice([0.1, 0.25, 0.07]))
        y = rng.standard_normal(3400) * 2.0
        for k in range(40, 3300, 23):
            s_in, s_out = rng.choice([-1.0, 1.0], 2)
            eps = float(rng.choice([0.0, 6e-9, -6e-9, 5e-8, -5e-8, 2e-7]))
            y[k] = y[k - 1] + s_in * (5.0 + rng.random()) * lam          # a bend known a priori into the run
            y[k + 1] = y[k] + float(rng.choice([-1.0, 1.0])) * 4.0 * lam * (1.0 + eps)   # the inner edge, a hair from 4 lambda
            y[k + 2] = y[k + 1] + s_out * (5.0 + rng.random()) * lam     # ... and one out of it
        x, st = runs(harness, y, lam)
        truth = oracle.tv1_linearized(y, lam)
        assert np.max(np.abs(x - truth)) <= 1e-13 * np.max(np.abs(y)), (t, lam, st)
