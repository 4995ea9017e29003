"""The replay path of the along-fibre kernel on the host (proxtv_amd/csrc/chunkcore.hpp: replay_lane; tests/host_harness.cpp:
host_replay_fibre runs a fibre through it the way the kernel does, lane after lane): a recorded structure -- piece ends and bend
types of an EARLIER fibre -- is verified against the optimality conditions of the prox on the new data instead of walking.
The property that makes it safe: whatever the candidate is (the fibre's own structure, a neighbour iterate's, a corrupted one, an
unrelated one), a segment that verifies IS exact.  And what makes it pay: a fibre's own structure always verifies."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CH, SEG = 17, 64 * 17


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_rp_"), "libchunk_host.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_structure.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    lib.host_replay_fibre.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    return lib


def structure(lib, y, lam):
    nc = (y.size + CH - 1) // CH
    e, t = np.zeros(nc, np.uint32), np.zeros(nc, np.uint32)
    lib.host_structure(y.ctypes.data, lam, y.size, CH, e.ctypes.data, t.ctypes.data)
    return e, t


FORMS = (0, 1)   # 0: a verification pass of its own (replay_lane), every interior segment ; 1: the check rides on the rebuild
                 # (rebuild_owned FULL = 3: what sweep_along_kernel runs), segments with nothing recorded between the known bends and them


def replay(lib, y, lam, e, t, fused=1):
    x = np.full(y.size, np.nan)
    nseg = (y.size + SEG - 1) // SEG
    ok = np.zeros(nseg, np.int32)
    n = lib.host_replay_fibre(y.ctypes.data, lam, y.size, e.ctypes.data, t.ctypes.data, x.ctypes.data, ok.ctypes.data, fused)
    assert n == int(ok.sum())
    return x, ok


def check_verified(x, ok, truth, what):
    scale = max(1.0, np.max(np.abs(truth)))
    for sg in np.flatnonzero(ok):
        a, b = sg * SEG, (sg + 1) * SEG
        err = np.max(np.abs(x[a:b] - truth[a:b]))
        assert err <= 1e-13 * scale, (what, int(sg), err)
    return int(ok.sum())


def families(rng, n, kind):
    if kind == 0:
        return rng.standard_normal(n)
    if kind == 1:
        return np.repeat(rng.standard_normal(n // 7 + 1), 7)[:n] + 0.3 * rng.standard_normal(n)
    if kind == 2:
        return np.cumsum(rng.standard_normal(n)) * 0.2 + rng.standard_normal(n)
    return rng.standard_normal(n) * 3.0


def interior_segments(n):
    return sum(1 for sg in range((n + SEG - 1) // SEG) if sg * SEG + SEG + 8 <= n - 1)


def test_own_structure_always_verifies_and_is_exact(harness, oracle):
    rng = np.random.default_rng(11)
    total = 0
    for trial in range(40):
        n = int(rng.integers(2300, 6000))
        y = families(rng, n, trial % 4)
        lam = float(rng.choice([0.05, 0.1, 0.2, 0.3]))
        truth = oracle.tv1_hybrid(y, lam)
        e, t = structure(harness, y, lam)
        x, ok = replay(harness, y, lam, e, t, fused=0)
        got = check_verified(x, ok, truth, f"own structure trial {trial}")
        # every interior segment that has its two bends known a priori verifies: where most edges are such bends, nearly all do
        total += got
        if np.mean(np.abs(np.diff(y)) > 4.0000001 * lam) >= 0.5:
            assert got >= interior_segments(n) - 1, (trial, got, interior_segments(n))
        xf, okf = replay(harness, y, lam, e, t, fused=1)
        gotf = check_verified(xf, okf, truth, f"own structure, fused, trial {trial}")
        assert np.all(okf <= ok)          # the fused form takes a subset of the segments ...
        fused_total = globals().setdefault("_fused_total", [0])
        fused_total[0] += gotf
    assert total > 80
    assert globals()["_fused_total"][0] > 0.4 * total   # ... most of them on noisy data


def test_neighbouring_iterate_verified_segments_are_exact(harness, oracle):
    """The candidate is the structure of a slightly different fibre (the previous iteration of a splitting loop): segments whose
    structure did not change verify, the others do not -- and every one that verifies is exact."""
    rng = np.random.default_rng(12)
    seen = {eps: [0, 0] for eps in (1e-4, 1e-3, 1e-2, 1e-1, 1.0)}
    for trial in range(60):
        n = int(rng.integers(2300, 6000))
        y_old = families(rng, n, trial % 4)
        lam = float(rng.choice([0.05, 0.1, 0.3]))
        e, t = structure(harness, y_old, lam)
        for eps in seen:
            y = y_old + eps * rng.standard_normal(n)
            truth = oracle.tv1_hybrid(y, lam)
            for form in FORMS:
                x, ok = replay(harness, y, lam, e, t, fused=form)
                got = check_verified(x, ok, truth, f"eps {eps} trial {trial} form {form}")
                if form == 0:
                    seen[eps][0] += got
                    seen[eps][1] += interior_segments(n)
    assert seen[1e-4][0] > 0.5 * seen[1e-4][1]      # small moves keep most segments' structure
    assert seen[1.0][0] < 0.1 * seen[1.0][1]        # another fibre's structure is (all but) never this fibre's


def test_corrupted_and_unrelated_candidates_never_verify_wrongly(harness, oracle):
    rng = np.random.default_rng(13)
    accepted = 0
    for trial in range(80):
        n = int(rng.integers(2300, 4500))
        y = families(rng, n, trial % 4)
        lam = float(rng.choice([0.05, 0.1, 0.3, 0.6]))
        truth = oracle.tv1_hybrid(y, lam)
        e, t = structure(harness, y, lam)
        for mode in range(5):
            e2, t2 = e.copy(), t.copy()
            if mode == 0:      # a handful of flipped piece ends
                for _ in range(int(rng.integers(1, 6))):
                    e2[rng.integers(0, e2.size)] ^= np.uint32(1 << int(rng.integers(0, CH)))
            elif mode == 1:    # flipped bend types
                for _ in range(int(rng.integers(1, 6))):
                    t2[rng.integers(0, t2.size)] ^= np.uint32(1 << int(rng.integers(0, CH)))
            elif mode == 2:    # every sample its own piece
                e2[:] = (1 << CH) - 1
            elif mode == 3:    # nothing recorded (a fresh buffer)
                e2[:] = 0
                t2[:] = 0
            else:              # the structure of an unrelated fibre
                e2, t2 = structure(harness, families(rng, n, (trial + 1) % 4), lam)
            for form in FORMS:
                x, ok = replay(harness, y, lam, e2, t2, fused=form)
                accepted += check_verified(x, ok, truth, f"corruption {mode} trial {trial} form {form}")
                if mode == 3:
                    assert ok.sum() == 0
    assert accepted > 0   # (flips that land outside a segment leave it verifiable: the test must have seen acceptances too)


def test_a_piece_end_recorded_on_the_segments_last_sample(harness, oracle):
    """The hole the GPU found in the first fused form: the record ends a piece ON the segment's last sample and another one on the next
    sample (two one-sample pieces), the new data merge the two, and the first bend known a priori lies one sample further on.  The
    piece [seg_e, kR) and the knot before it then belong to no lane of the segment: such a segment must walk."""
    rng = np.random.default_rng(14)
    lam = 0.1
    hits = 0
    for trial in range(200):
        n = 2 * SEG + 300
        y_old = rng.standard_normal(n)
        y_old[SEG - 1], y_old[SEG], y_old[SEG + 1] = -1.0, 1.0, 3.0       # knots at SEG and SEG + 1 in the record
        e, t = structure(harness, y_old, lam)
        assert (e[(SEG - 1) // CH] >> ((SEG - 1) % CH)) & 1 and (e[SEG // CH] >> (SEG % CH)) & 1
        y = y_old + 1e-3 * rng.standard_normal(n)
        y[SEG - 1], y[SEG] = 0.30 + 0.01 * rng.standard_normal(), 0.31 + 0.01 * rng.standard_normal()   # ... merged in the new data
        truth = oracle.tv1_hybrid(y, lam)
        for form in FORMS:
            x, ok = replay(harness, y, lam, e, t, fused=form)
            hits += check_verified(x, ok, truth, f"boundary trial {trial} form {form}")
    assert hits >= 0
