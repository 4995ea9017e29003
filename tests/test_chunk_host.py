"""The per-lane logic of the speculative-chunk kernels (proxtv_amd/csrc/chunkcore.hpp: certain-bend starts, the
branch-free interior walk, the ownership rebuild), compiled for the host and run lane after lane the way a workgroup
column does it, against the oracle -- no GPU needed."""
import ctypes as C
import os
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module", params=["quotients", "table reciprocals"])
def harness(request):
    """Twice: every span quotient an IEEE division (one rounding for all walks), and -DPTV_TABLE_RECIP -- the chunk walks and the rebuild
    multiply by the rounded reciprocal of the span as the device's table loops do, while walker_run (the reference of the unproven-lane
    check, the device's repair walks) divides: the device's two roundings, ties cut differently."""
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_ch_"), "libchunk_host.so")
    flags = ["-DPTV_TABLE_RECIP"] if request.param == "table reciprocals" else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", *flags, "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_whole_fibre.argtypes = [C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_void_p]
    lib.host_chunk_fibre.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                     C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def run(lib, y, lam, w=None, H=16, T=8, NW=8, past=0, seed=0):
    y = np.ascontiguousarray(y, dtype=np.float64)
    x = np.full(y.size, np.nan)
    fb, we = C.c_int(0), C.c_int(0)
    lib.host_chunk_fibre(y.ctypes.data, None if w is None else w.ctypes.data, lam, y.size, H, T, NW, past, seed,
                         x.ctypes.data, C.byref(fb), C.byref(we))
    return x, fb.value, we.value


def families(rng, n):
    kind = int(rng.integers(0, 5))
    if kind == 0:
        return rng.standard_normal(n)
    if kind == 1:
        return np.repeat(rng.standard_normal(n // 5 + 1), 5)[:n] + 0.05 * rng.standard_normal(n)
    if kind == 2:
        return rng.integers(-2, 3, n).astype(float)
    if kind == 3:
        return np.cumsum(rng.standard_normal(n)) * 0.3
    return rng.standard_normal(n) * rng.choice([1e-3, 1.0, 1e3])


def check(x, fb, we, truth, H, scale):
    assert we == 0, f"{we} rows not written exactly once"
    upto = truth.size if fb < 0 else fb
    err = np.max(np.abs(x[:upto] - truth[:upto])) if upto else 0.0
    assert err <= 1e-13 * scale, (err, fb)
    return upto


def test_chunked_walk_equals_oracle_unweighted(harness, oracle):
    rng = np.random.default_rng(0)
    covered = total = 0
    for t in range(1500):
        n = int(rng.integers(1, 700))
        y = families(rng, n)
        lam = float(rng.choice([0.0, 0.02, 0.1, 0.3, 1.0]) * abs(rng.standard_normal()))
        truth = oracle.tv1_linearized(y, lam)
        for (H, T, NW, past) in ((16, 8, 8, 0), (16, 8, 8, 1), (64, 64, 8, 0), (16, 8, 3, 0), (16, 8, 64, 0), (64, 64, 64, 0), (16, 8, 32, 0), (64, 64, 16, 0),
                                    (16, 64, 64, 0), (16, 64, 32, 0)):   # (64 look-ahead rows: the robust along-fibre geometry, chunks of 31)
            x, fb, we = run(harness, y, lam, H=H, T=T, NW=NW, past=past, seed=t)
            covered += check(x, fb, we, truth, H, max(1.0, np.max(np.abs(y))))
            total += n
    assert covered > 0.6 * total    # the scheme proves most links on this mix (what it cannot prove goes to the repair kernel)


def test_chunked_walk_headline_regime_is_fully_proven(harness, oracle):
    rng = np.random.default_rng(1)
    for t in range(200):
        n = int(rng.integers(100, 3000))
        y = rng.standard_normal(n)
        truth = oracle.tv1_linearized(y, 0.1)
        for NW in (8, 64, 32, 16):     # 64-fibre tile geometry; along-fibre geometries (64 / 32 / 16 chunks of 17 samples)
            x, fb, we = run(harness, y, 0.1, NW=NW, seed=t)
            assert fb < 0 and we == 0
            assert np.max(np.abs(x - truth)) <= 1e-14 * np.max(np.abs(y))


def test_chunked_walk_equals_oracle_weighted(harness, oracle):
    rng = np.random.default_rng(2)
    covered = total = 0
    for t in range(800):
        n = int(rng.integers(2, 600))
        y = families(rng, n)
        w = rng.uniform(0.0, 1.0, n - 1) * float(rng.choice([0.0, 0.05, 0.2, 1.0]))
        truth = oracle.tv1_weighted(y, w)
        for past in (0, 1):
            x, fb, we = run(harness, y, 0.0, w=w, past=past, seed=t)
            covered += check(x, fb, we, truth, 16, max(1.0, np.max(np.abs(y))))
            total += n
    assert covered > 0.6 * total


def test_short_fibres_whole_in_window(harness, oracle):
    """sweep_whole_kernel's per-lane loop: outputs 32 samples at a time, every 32 restarting at the last bend at or
    before its first sample; every row written exactly once."""
    rng = np.random.default_rng(3)
    for t in range(3000):
        n = int(rng.integers(2, 97))
        y = np.ascontiguousarray(families(rng, n))
        lam = float(rng.choice([0.0, 0.02, 0.1, 0.5, 2.0, 50.0]) * abs(rng.standard_normal()))
        x = np.full(n, np.nan)
        bad = harness.host_whole_fibre(y.ctypes.data, lam, n, t & 1, x.ctypes.data)
        assert bad == 0, (t, n, lam, bad)
        assert np.max(np.abs(x - oracle.tv1_linearized(y, lam))) <= 1e-13 * max(1.0, np.max(np.abs(y))), (t, n, lam)



def test_unproven_lane_is_right_about_its_own_rows_degenerate_knot(harness, oracle):
    """tests/golden/degenerate_knot_fibre.npz (make_degenerate_knot.py): a knot whose jump is zero up to rounding between samples 93 and 94.
    On the device the repair walk bends there and the chunk's own walk does not; the repair walk hands over to the chunk because both
    came from the same bend at 80, and rows 94, 95 are then the chunk's -- right only if an UNPROVEN lane values its first piece from the
    piece's true first row (chunkcore.hpp rebuild_owned, a0 / w0).  The harness checks that for every unproven lane (write_errors)."""
    g = np.load(os.path.join(HERE, "golden", "degenerate_knot_fibre.npz"))
    y, lam = np.ascontiguousarray(g["y"]), float(g["lam"])
    assert np.max(np.abs(oracle.tv1_linearized(y, lam) - g["expected"])) == 0.0
    unproven = 0
    for seed in range(4):
        for (H, T, NW) in ((16, 8, 64), (16, 8, 32), (16, 8, 8), (64, 64, 64)):
            x, fb, we = run(harness, y, lam, H=H, T=T, NW=NW, seed=seed)
            assert we == 0, (seed, H, T, NW, we)
            unproven += fb >= 0
    assert unproven > 0   # (at this lambda nothing is known a priori: the links fail and the repair walk is what the device runs)


def _zero_jump_fibre(rng, n, lam):
    """A fibre built backwards from its solution: x piecewise constant with some knots whose jump is exactly zero, dual z with |z| <= 1,
    z = +-1 at every knot (the zero-jump ones too) and at some rows inside pieces; y = x + lam (z_{i-1} - z_i).  In floating point the
    string touches the tube to the last bit at those places: a bend to one rounding of the walk, none to another -- what the operands
    of the late iterations of a Dykstra / DR loop look like."""
    x = np.empty(n)
    z = rng.uniform(-0.95, 0.95, n - 1) if n > 1 else np.zeros(0)
    i, v = 0, float(rng.standard_normal())
    while i < n:
        L = int(rng.integers(1, 41))
        x[i:i + L] = v
        e = i + L - 1                      # the piece's last row; edge e joins it to the next piece
        if e < n - 1:
            kind = rng.random()
            if kind < 0.4:                 # zero jump, the wall touched all the same
                z[e] = float(rng.choice([-1.0, 1.0]))
                nv = v
            else:
                step = float(abs(rng.standard_normal())) * rng.choice([1e-3, 1.0])
                up = rng.random() < 0.5
                z[e] = 1.0 if up else -1.0
                nv = v + step if up else v - step
            for k in range(i, min(e, n - 1)):   # rows inside the piece that touch a wall without bending
                if rng.random() < 0.05:
                    z[k] = float(rng.choice([-1.0, 1.0]))
            v = nv
        i += L
    zz = np.concatenate(([0.0], z, [0.0]))
    return x + lam * (zz[:-1] - zz[1:]), x


def test_chunked_walk_on_fibres_with_zero_jump_knots(harness, oracle):
    """Ties everywhere (see _zero_jump_fibre): walks that round differently may cut such a fibre differently, and every place where the
    scheme joins what two walks found has to survive that -- the links (code equality), the rebuild's ownership, and an unproven lane's
    own rows, which a repair walk may hand over to (round 5: tests/golden/degenerate_knot_fibre.npz).  Values agree to rounding whatever
    the cut."""
    rng = np.random.default_rng(7)
    covered = total = 0
    for t in range(400):
        n = int(rng.integers(2, 900))
        lam = float(rng.choice([0.05, 0.5, 3.0]) * (0.5 + rng.random()))
        y, x = _zero_jump_fibre(rng, n, lam)
        truth = oracle.tv1_linearized(y, lam)
        scale = max(1.0, np.max(np.abs(y)))
        assert np.max(np.abs(truth - x)) <= 1e-12 * scale      # (the construction: x IS the prox of y)
        for (H, T, NW, past) in ((16, 8, 8, 0), (16, 8, 64, 0), (16, 8, 32, 0), (64, 64, 64, 0), (64, 64, 8, 0), (16, 8, 3, 1)):
            got, fb, we = run(harness, y, lam, H=H, T=T, NW=NW, past=past, seed=t)
            covered += check(got, fb, we, truth, H, scale)
            total += n
    assert covered > 0.2 * total    # (long pieces at these penalties: most links are the repair kernel's)


def test_short_fibres_with_zero_jump_knots(harness, oracle):
    """sweep_whole_kernel's loop (outputs 32 samples at a time, each 32 restarting at the last bend at or before its first sample) on short
    fibres full of ties."""
    rng = np.random.default_rng(13)
    for t in range(1500):
        n = int(rng.integers(2, 97))
        lam = float(rng.choice([0.05, 0.5, 3.0]) * (0.5 + rng.random()))
        y = np.ascontiguousarray(_zero_jump_fibre(rng, n, lam)[0])
        x = np.full(n, np.nan)
        bad = harness.host_whole_fibre(y.ctypes.data, lam, n, t & 1, x.ctypes.data)
        assert bad == 0, (t, n, lam, bad)
        assert np.max(np.abs(x - oracle.tv1_linearized(y, lam))) <= 1e-10 * max(1.0, np.max(np.abs(y))), (t, n, lam)
