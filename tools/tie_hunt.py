#!/usr/bin/env python3
"""CPU only: the operands the 1-D proxes see LATE in a Dykstra (PD2_TV) or Douglas-Rachford (DR2_TV) loop -- where the string has knots whose
jump is zero up to rounding -- through the host model of the chunk + repair stage with the DEVICE's two roundings (tests/repair_model_host.cpp,
model_set_table: chunk walks multiply by the rounded reciprocal of a span, repair walks divide).  The outer loops are emulated in numpy with
the oracle's 1-D prox and checked against the oracle's own PD2_TV / DR2_TV.  Every fibre of every operand must come out exact from all four
repairs; `--legacy` runs the rebuild's semantics of rounds 1-4 instead (it finds the failure of round 5's soak, and how rare it is).

    python tools/tie_hunt.py [--legacy] [images] [seed]
    python tools/tie_hunt.py [--legacy] --fuzz-case SEED INDEX      # the image of case INDEX of `tools/fuzz.py <t> SEED` (PD2 / DR on it, iterations 12-35)
"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu

so = os.path.join(tempfile.gettempdir(), "repair_model_tie.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(ROOT, "tests", "repair_model_host.cpp")], check=True)
lib = C.CDLL(so)
lib.model_fibres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
for f in (lib.model_set_table, lib.model_set_legacy):
    f.argtypes = [C.c_int]


def data(rng, kind, shape):
    M, N = shape
    if kind == 0: return rng.standard_normal(shape)
    if kind == 1: return np.kron(rng.standard_normal((M // 16 + 1, N // 16 + 1)), np.ones((16, 16)))[:M, :N] + 0.2 * rng.standard_normal(shape)
    if kind == 2: return np.add.outer(np.linspace(-3, 3, M), np.linspace(2, -2, N)) + 0.05 * rng.standard_normal(shape)
    if kind == 3: return np.full(shape, 1.5) + (rng.random(shape) < 0.01) * 8.0
    if kind == 4: return np.cumsum(rng.standard_normal(shape), axis=int(rng.integers(0, 2))) * 0.3
    return np.round(rng.standard_normal(shape) * 3)


def fibres(A, axis):
    return np.ascontiguousarray(A.T if axis == 0 else A)     # rows of the result = the fibres along `axis`


def check(F, lam, tally, what):
    for Cn in (17, 16):     # along-fibre kernel / tiles
        out, worst = np.zeros(12, dtype=np.int64), np.zeros(4)
        first = lib.model_fibres(F.ctypes.data, None, F.shape[0], F.shape[1], lam, Cn, 16, 128, 4, out.ctypes.data, worst.ctypes.data)
        tally["fibres"] += F.shape[0]
        tally["in doubt"] += int(out[0])
        tally["links"] += int(out[1])
        bad = int(out[3:7].max())
        if bad:
            tally["wrong"] += bad
            print(f"    WRONG: {what}, chunks of {Cn}: fibres wrong after seq old / seq new / jobs / jobs+guard {out[3:7]}, worst {worst}, first {first}", flush=True)


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    legacy = "--legacy" in sys.argv
    images = int(args[0]) if args and "--fuzz-case" not in sys.argv else 6
    rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 5)
    orc = cpu.oracle()
    lib.model_set_table(1)
    lib.model_set_legacy(1 if legacy else 0)
    tally = {"fibres": 0, "in doubt": 0, "links": 0, "wrong": 0}
    late = (12, 24, 31, 35)
    fuzz_case = None
    if "--fuzz-case" in sys.argv:
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import make_degenerate_knot as mk
        fuzz_case = mk.case(int(args[0]), int(args[1]))
        images, late = 1, tuple(range(12, 36))
    for im in range(images):
        if fuzz_case:
            X, lam, kind = np.asfortranarray(fuzz_case[0]), fuzz_case[1], -1
            M, N = X.shape
        else:
            M, N = (int(v) for v in rng.choice([257, 400, 700], 2))
            kind = int(rng.integers(0, 6))
            X = np.asfortranarray(data(rng, kind, (M, N)))
            lam = float(10 ** rng.uniform(-1, 1))
        prox = lambda A, axis: np.asfortranarray(np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), lam), axis, A))
        print(f"image {im}: {M} x {N}, family {kind}, lambda {lam:.4g}", flush=True)
        # proximal Dykstra (src/TV2Dopt.cpp:59-302): z = prox_1(x + p), p += x - z ; x' = prox_2(z + q), q += z - x'
        x, p, q = X.copy(), np.zeros_like(X), np.zeros_like(X)
        for k in range(1, 36):
            a_in = x + p
            z = prox(a_in, 0); p = p + (x - z)
            b_in = z + q
            xn = prox(b_in, 1); q = q + (z - xn)
            x = xn
            if k in late:
                check(fibres(a_in, 0), lam, tally, f"PD2 iteration {k}, columns")
                check(fibres(b_in, 1), lam, tally, f"PD2 iteration {k}, rows")
        ref = orc.pd2(X, [lam, lam], [1, 2])[0]
        assert int(orc.pd2(X, [lam, lam], [1, 2])[1][0]) < 35 or np.max(np.abs(ref - x)) <= 1e-12 * max(1.0, np.max(np.abs(X))), "PD2 emulation"
        # Douglas-Rachford (src/TV2Dopt.cpp:420-560): s' = 2 (t - prox_c(t)) - t ; v = U - s' ; t <- t / 2 + prox_r(v) + s' / 2
        t = np.full_like(X, X.sum() / X.size)
        for k in range(1, 36):
            sp = 2.0 * (t - prox(t, 0)) - t
            v = X - sp
            if k in late:
                check(fibres(t, 0), lam, tally, f"DR iteration {k}, columns")
                check(fibres(v, 1), lam, tally, f"DR iteration {k}, rows")
            t = 0.5 * t + prox(v, 1) + 0.5 * sp
        s = t - prox(t, 0)
        out = prox(X - s, 1)
        e = np.max(np.abs(out - orc.dr2(X, lam)[0])) / max(1.0, np.max(np.abs(X)))
        print(f"    (DR emulation against the oracle's DR2_TV: {e:.1e})", flush=True)
    print(f"# {tally['fibres']} fibre runs, {tally['in doubt']} with a link in doubt ({tally['links']} links), WRONG: {tally['wrong']}"
          f"   [{'rounds 1-4 semantics' if legacy else 'current semantics'}, device roundings]")
    return 1 if tally["wrong"] and not legacy else 0


if __name__ == "__main__":
    sys.exit(main())
