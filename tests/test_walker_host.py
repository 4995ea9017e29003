"""The per-lane state machine of the HIP kernels (proxtv_amd/csrc/walker.hpp), compiled for the host by
tests/host_harness.cpp and checked against the oracle without a GPU: same arithmetic, so the bar is bit equality."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_hw_"), "libwalker_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_walk.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int]
    lib.host_walk_from.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.host_walk_blocked.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_int, C.c_int, C.c_int]
    lib.policy_sim.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.policy_sim_pin.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p]
    return lib


def walk(lib, x, lam, w=None):
    x = np.ascontiguousarray(x, dtype=np.float64)
    out = np.zeros_like(x)
    lib.host_walk(x.ctypes.data, None if w is None else w.ctypes.data, lam, out.ctypes.data, x.size)
    return out


def signals(rng, count):
    for t in range(count):
        n = int(rng.integers(1, 80))
        kind = t % 4
        if kind == 0:
            x = rng.standard_normal(n)
        elif kind == 1:
            x = np.repeat(rng.standard_normal(n // 7 + 1), 7)[:n] + 0.1 * rng.standard_normal(n)
        elif kind == 2:
            x = rng.integers(-2, 3, n).astype(float)
        else:
            x = np.cumsum(rng.standard_normal(n))
        yield x


def test_walker_equals_oracle_unweighted(harness, oracle):
    rng = np.random.default_rng(0)
    for x in signals(rng, 4000):
        lam = float(abs(rng.standard_normal()) * rng.choice([0.0, 0.1, 1, 3, 30]))
        np.testing.assert_array_equal(walk(harness, x, lam), oracle.tv1_linearized(x, lam))


def test_walker_equals_oracle_negative_lambda(harness, oracle):
    """Negative penalties reach the solver through tvgen (prox_tv_test.py:202-209); behaviour must be the reference's."""
    rng = np.random.default_rng(1)
    for x in signals(rng, 2000):
        lam = -float(abs(rng.standard_normal()) * rng.choice([0.1, 1, 3]))
        np.testing.assert_array_equal(walk(harness, x, lam), oracle.tv1_linearized(x, lam))


def test_walker_equals_oracle_weighted(harness, oracle):
    rng = np.random.default_rng(2)
    for x in signals(rng, 3000):
        if x.size < 2:
            continue
        w = rng.uniform(0, 2, x.size - 1) * float(rng.choice([0.0, 0.1, 1, 5]))
        np.testing.assert_array_equal(walk(harness, x, 0.0, w), oracle.tv1_weighted(x, w))


def test_walker_vs_hybrid_and_condat(harness, oracle):
    """Against the library default (hybrid, may switch to the classic algorithm) and Condat: same minimiser."""
    rng = np.random.default_rng(3)
    for x in signals(rng, 1500):
        lam = float(abs(rng.standard_normal()) * rng.choice([0.1, 1, 3]))
        got = walk(harness, x, lam)
        scale = max(np.max(np.abs(x)), 1.0)
        assert np.max(np.abs(got - oracle.tv1_hybrid(x, lam, 0.5))) <= 1e-12 * scale
        assert np.max(np.abs(got - oracle.tv1_condat(x, lam))) <= 1e-12 * scale


def test_speculative_start_synchronises(harness, oracle):
    """The fact the chunked kernels rely on: a walk started mid-fibre from a free end coincides with the true walk
    from its first bend whose restart index is also a restart of the true walk -- in particular, for noisy data and
    small lambda, after a few samples."""
    rng = np.random.default_rng(4)
    n = 400
    for trial in range(200):
        x = rng.standard_normal(n)
        lam = float(rng.choice([0.05, 0.1, 0.3]))
        truth = oracle.tv1_linearized(x, lam)
        start = int(rng.integers(20, 300))
        spec = np.full(n, np.nan)
        harness.host_walk_from(x.ctypes.data, None, lam, spec.ctypes.data, n, start, start + 80)
        covered = ~np.isnan(spec)
        assert covered[start:start + 80].all()
        agree = spec == truth
        # find the first agreeing sample after which everything covered agrees
        idx = np.where(covered)[0]
        bad = idx[~agree[idx]]
        first_ok = (bad.max() + 1) if bad.size else start
        assert first_ok - start <= 32, (trial, first_ok - start)     # synchronised within the warm-up zone


def test_blocked_walk_is_bit_identical(harness, oracle):
    """walker_run_blocked (global-memory kernels: K samples fetched per block) against walker_run: same bits, same
    bends, for full walks, negative penalties, weighted walks and speculative starts that stop early."""
    rng = np.random.default_rng(5)
    for t, x in enumerate(signals(rng, 6000)):
        n = x.size
        a, b = np.zeros(n), np.zeros(n)
        if t % 3 == 2 and n >= 2:
            w = rng.uniform(0, 2, n - 1) * float(rng.choice([0.0, 0.1, 1, 5]))
            lam, wp = 0.0, w.ctypes.data
        else:
            w = None
            lam, wp = float(rng.standard_normal() * rng.choice([0.0, 0.1, 1, 3, 30])), None
            if lam < 0 and n < 3:
                continue
        na = harness.host_walk(x.ctypes.data, wp, lam, a.ctypes.data, n)
        nb = harness.host_walk_blocked(x.ctypes.data, wp, lam, b.ctypes.data, n, 0, -1)
        np.testing.assert_array_equal(a, b)
        assert na == nb
        if n >= 20 and lam >= 0:
            start = int(rng.integers(1, n - 10))
            until = min(n - 1, start + 9)
            a[:], b[:] = np.nan, np.nan
            na = harness.host_walk_from(x.ctypes.data, wp, lam, a.ctypes.data, n, start, until)
            nb = harness.host_walk_blocked(x.ctypes.data, wp, lam, b.ctypes.data, n, start, until)
            np.testing.assert_array_equal(a, b)
            assert na == nb


# ---- the geometry policy (proxtv_amd/csrc/policy.hpp) under a cost model ---------------------------------------------------
def _simulate(lib, cost, frac, solves=4, sweeps=35, switch_at=10**6, length=4096, weighted=False, start=0, cost2=None, frac2=None):
    table_c = np.array(list(cost) + list(cost2 if cost2 is not None else cost), dtype=np.float64)
    table_f = np.array(list(frac) + list(frac2 if frac2 is not None else frac), dtype=np.float64)
    total = np.zeros(1)
    trace = np.zeros(sweeps, dtype=np.int32)
    mode = lib.policy_sim(table_c.ctypes.data, table_f.ctypes.data, switch_at, solves, sweeps, length, int(weighted), start,
                          total.ctypes.data, trace.ctypes.data)
    return mode, float(total[0]), trace


# sweep time (ms, chunk + repair kernels) and rewritten-chunk fraction per mode, as measured on 4096^2 DR iterates
REGIMES = {
    # name: (cost[0..5], frac[0..5], best mode)
    "headline lam=0.1": ([0.15, 0.16, 0.45, 1.5, 2.0, 6.0], [0, 0, 0, 0, 0, -1], 0),
    "lam=0.3": ([0.24, 0.17, 0.5, 1.5, 2.0, 6.0], [1.5e-5, 1e-7, 0, 0, 0, -1], 1),
    "lam=0.7": ([0.85, 0.57, 0.75, 1.3, 1.6, 6.0], [0.02, 3e-3, 1e-5, 0, 0, -1], 1),
    "lam=1": ([7.0, 3.7, 1.25, 1.2, 1.6, 6.0], [0.3, 0.1, 4e-4, 0, 0, -1], 3),
    "lam=3 columns": ([14.0, 12.0, 9.0, 12.0, 8.0, 4.3], [1.0, 1.0, 0.9, 0.9, 0.13, -1], 5),
    "lam=30": ([10.4, 10.4, 11.5, 9.0, 15.0, 4.3], [1.0, 1.0, 1.0, 1.0, 1.0, -1], 5),
}


def test_policy_finds_the_fastest_geometry_from_anywhere(harness):
    for name, (cost, frac, best) in REGIMES.items():
        for start in range(6):
            mode, total, trace = _simulate(harness, cost, frac, solves=5, start=start)
            assert cost[mode] <= 1.12 * cost[best], (name, start, mode, best)
            # steady state: a solve costs little more than 35 sweeps of the best mode (one trial sweep at most)
            assert total <= 35 * cost[mode] + 2 * max(cost) and total <= 1.5 * 35 * cost[best] + max(cost), (name, start, total)


def test_policy_headline_pays_nothing(harness):
    cost, frac, _ = REGIMES["headline lam=0.1"]
    mode, total, trace = _simulate(harness, cost, frac, solves=3)
    assert mode == 0 and (trace == 0).all() and abs(total - 35 * cost[0]) < 1e-12


def test_policy_follows_a_drift_inside_a_solve(harness):
    """DR iterates at large lambda grow longer pieces sweep after sweep: what is best early in the solve is not later."""
    early = ([14.0, 12.0, 5.3, 4.9, 5.5, 5.9], [0.6, 0.3, 1e-3, 3e-4, 0, -1])
    late = ([14.0, 12.0, 14.0, 12.0, 8.5, 7.5], [1.0, 1.0, 0.3, 0.06, 0.1, -1])
    mode, total, trace = _simulate(harness, early[0], early[1], solves=3, switch_at=12, cost2=late[0], frac2=late[1])
    assert trace[-1] == 5 and (trace[-10:] == 5).all(), trace          # the late phase ends on the sequential walk
    assert set(trace[4:10]) <= {2, 3, 4, 5}, trace                        # ... after running a chunked geometry early on
    assert total <= 1.35 * (12 * 4.9 + 23 * 7.5), total


def test_policy_respects_unavailable_geometries(harness):
    cost, frac, _ = REGIMES["lam=1"]
    cost_w = list(cost); cost_w[2] = 0.01                                 # would win if it existed for weighted sweeps
    mode, total, trace = _simulate(harness, cost_w, frac, weighted=True, solves=4)
    assert mode != 2 and 2 not in trace
    cost_s = [14.0, 12.0, 9.0, 12.0, 0.01, 4.3]                            # 1024-sample zones need 1024-sample fibres
    mode, total, trace = _simulate(harness, cost_s, [1, 1, 0.9, 0.9, 0.1, -1], length=700, solves=4)
    assert mode != 4 and 4 not in trace


def test_policy_one_sweep_solves_explore_across_calls(harness):
    cost, frac, best = REGIMES["lam=0.3"]
    mode, total, trace = _simulate(harness, cost, frac, solves=6, sweeps=1, start=0)
    assert mode == best
    cost, frac, best = REGIMES["headline lam=0.1"]
    mode, total, trace = _simulate(harness, cost, frac, solves=8, sweeps=1, start=5)
    assert mode == best


def _simulate_pin(lib, cost, frac, cost2=None, frac2=None, switch_at=10 ** 6, solves=4, sweeps=35, length=4096, start=0):
    table_c = np.array(list(cost) + list(cost2 if cost2 is not None else cost), dtype=np.float64)
    table_f = np.array(list(frac) + list(frac2 if frac2 is not None else frac), dtype=np.float64)
    total = np.zeros(1)
    trace = np.zeros(sweeps, dtype=np.int32)
    mode = lib.policy_sim_pin(table_c.ctypes.data, table_f.ctypes.data, switch_at, solves, sweeps, length, 0, start, 1,
                              total.ctypes.data, trace.ctypes.data)
    return mode, float(total[0]), trace


def test_policy_with_a_pinning_rung(harness):
    """Rung 3 = the pinning solver: sweep time independent of the data, frac = pieces per sample of its result.
    (i) a fresh workload opens on it, so long pieces never see a chunk kernel -- whose repair walks are what costs;
    (ii) short pieces send the policy down to the chunk kernels; (iii) a chunk geometry that drifts past the pinning
    rung's time inside a solve gives way to it."""
    # 4096^2 DR iterates at lambda = 3: rungs 0-2 cost tens of ms in repairs
    long_cost, long_frac = [57.0, 40.0, 9.0, 0.25, 8.0, 4.3], [0.4, 0.3, 0.01, 0.01, 0.1, -1]
    for start in range(6):
        mode, total, trace = _simulate_pin(harness, long_cost, long_frac, solves=1, start=start)
        assert mode == 3 and (trace == 3).all(), (start, mode, trace)          # first solve, from any stale state: rung 3 only
        assert abs(total - 35 * 0.25) < 1e-9
    # headline data: white noise, lambda = 0.1 -- one piece per sample
    short_cost, short_frac = [0.09, 0.10, 0.45, 0.55, 2.0, 6.0], [0, 0, 0, 0.9, 0, -1]
    mode, total, trace = _simulate_pin(harness, short_cost, short_frac, solves=3)
    assert mode == 0 and (trace == 0).all() and abs(total - 35 * 0.09) < 1e-9
    mode, total, trace = _simulate_pin(harness, short_cost, short_frac, solves=1)
    assert mode == 0 and total <= 35 * 0.09 + 0.55 + 0.10 + 0.01                    # first solve: one look at rung 3, one at rung 1
    # lambda = 0.7: rung 1 starts a solve fast and ends it slow, the pinning rung takes 0.45 ms throughout
    early, late = [0.5, 0.30, 0.6, 0.45, 2.0, 6.0], [1.2, 0.62, 0.9, 0.45, 2.0, 6.0]
    fr = [0.01, 1e-4, 0, 0.25, 0, -1]
    mode, total, trace = _simulate_pin(harness, early, fr, cost2=late, frac2=[0.02, 6e-4, 0, 0.15, 0, -1], switch_at=15, solves=4)
    assert trace[-1] == 3 and total <= 15 * 0.45 + 20 * 0.45 + 2.0, (trace, total)


def test_policy_one_sweep_solves_react_to_harder_data(harness):
    """Batched 1-D prox calls are one-sweep solves: with a pinning rung every call is timed, and a call that takes longer than
    the rung's yardstick is the last slow one (a 10^6-sample fibre: 0.2 ms at lambda = 0.5 on rung 1, 800 ms there at lambda =
    30, 0.9 ms on the pinning rung whatever lambda)."""
    easy_c, easy_f = [0.25, 0.2, 1.0, 0.9, 50.0, 800.0], [1e-5, 0, 0, 0.9, 0, -1]
    hard_c, hard_f = [900.0, 800.0, 700.0, 0.9, 50.0, 800.0], [0.5, 0.4, 0.3, 0.001, 0.01, -1]
    # 12 easy calls settle on a chunk rung; from call 12 on the data are hard: one slow call, then the pinning rung
    for calls, want_mode, want_last in ((12, None, None), (13, None, None), (14, 3, 0.9), (20, 3, 0.9)):
        mode, last, trace = _simulate_pin(harness, easy_c, easy_f, cost2=hard_c, frac2=hard_f, switch_at=12, solves=calls, sweeps=1)
        if calls == 12:
            assert mode <= 1 and last <= 0.25, (mode, last)
        elif want_mode is not None:
            assert mode == want_mode and abs(last - want_last) < 1e-9, (calls, mode, last)
    # and back: on the pinning rung with short pieces (data got easy again) the policy looks below within a call or two
    mode, last, trace = _simulate_pin(harness, hard_c, hard_f, cost2=easy_c, frac2=easy_f, switch_at=6, solves=10, sweeps=1)
    assert mode <= 1 and last <= 0.25, (mode, last)


def test_both_forms_of_the_dr_iteration_on_the_host(harness, oracle):
    """The arithmetic of the two forms of the DR iteration (proxtv_amd/csrc/ops.hpp: OP_DR_COL / OP_DR_ROW -- the reference's
    split -- and OP_DR_COL_V / OP_DR_ROW_V -- the column sweep leaves v = U - s' and s, the row sweep returns t = s + prox(v)),
    restated with numpy around the host build of the device walker: both must reproduce the reference's DR2_TV, and each
    other to rounding, for any iteration count (weighted form included: same recurrence, src/TV2DWopt.cpp:114-126)."""
    rng = np.random.default_rng(77)

    def prox(A, lam, axis, W=None):
        A = np.ascontiguousarray(np.moveaxis(A, axis, -1))
        Wm = None if W is None else np.ascontiguousarray(np.moveaxis(W, axis, -1))
        out = np.empty_like(A)
        for j in range(A.shape[0]):
            out[j] = walk(harness, A[j], lam, None if Wm is None else np.ascontiguousarray(Wm[j]))
        return np.moveaxis(out, -1, axis)

    def dr(U, l1, l2, its, form, W1=None, W2=None):
        t = np.full(U.shape, 2 * (U.sum() / U.size))
        for _ in range(its):
            p = prox(t, l1, 0, W1)
            s = t - p
            sp = 2 * s - t
            v = U - sp
            x = prox(v, l2, 1, W2)
            t = (s + x) if form else 0.5 * (t + (sp + 2 * x))
        s = t - prox(t, l1, 0, W1)
        v = U - s
        return (U - (v - prox(v, l2, 1, W2))) - s

    for (M, N), lam in (((37, 53), 0.3), ((64, 20), 1.5), ((9, 120), 0.05)):
        U = np.asfortranarray(rng.standard_normal((M, N)) * 2)
        for its in (1, 4, 35):
            a, b = dr(U, lam, lam, its, 0), dr(U, lam, lam, its, 1)
            ref = oracle.dr2(U, lam, max_iters=its)[0]
            scale = np.max(np.abs(U))
            assert np.max(np.abs(a - ref)) <= 1e-12 * scale and np.max(np.abs(b - ref)) <= 1e-12 * scale, (M, N, lam, its)
        W1, W2 = rng.uniform(0, 2 * lam, (M - 1, N)), rng.uniform(0, 2 * lam, (M, N - 1))
        ref = oracle.dr2w(U, W1, W2)[0]
        for form in (0, 1):
            assert np.max(np.abs(dr(U, 0.0, 0.0, 35, form, W1, W2) - ref)) <= 1e-12 * np.max(np.abs(U)), (M, N, form)


def test_seeded_policy_rungs_and_sample_schedule(harness):
    """policy.hpp's pure functions: the rung a sampled certain fraction asks for -- general thresholds, the Dykstra operands' (x + p walks like
    noisier data than its certain fraction says), small sweeps' (a repair launch costs what it costs whatever the image) -- and the iterations
    before whose sweeps a loop samples its operands again."""
    lib = harness
    lib.policy_rung.argtypes = [C.c_double, C.c_int, C.c_int, C.c_int]
    lib.policy_reprobe_at.argtypes = [C.c_int, C.c_int]
    rung = lambda f, dykstra=0, small=0, weighted=0: lib.policy_rung(f, dykstra, small, weighted)
    assert [rung(f) for f in (0.78, 0.45, 0.44, 0.03, 0.029, 0.0)] == [0, 0, 1, 1, 3, 3]
    assert [rung(f, dykstra=1) for f in (0.5, 0.022, 0.009, 0.004, 0.0039, 0.001)] == [0, 1, 1, 1, 3, 3]
    assert [rung(f, small=1) for f in (0.5, 0.16, 0.06, 0.059, 0.034)] == [0, 1, 1, 3, 3]
    assert [rung(f, dykstra=1, small=1) for f in (0.16, 0.022)] == [1, 3]           # (small sweeps: one threshold for all operands)
    assert [rung(f, weighted=1) for f in (0.5, 0.095, 0.055, 0.054, 0.038)] == [0, 1, 1, 3, 3]
    assert [it for it in range(1, 40) if lib.policy_reprobe_at(it, 0)] == [2, 3, 5, 9, 17, 33]
    assert [it for it in range(1, 40) if lib.policy_reprobe_at(it, 1)] == [2, 3, 5, 9, 13, 17, 21, 25, 29, 33, 37]
