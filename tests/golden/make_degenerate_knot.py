#!/usr/bin/env python3
"""Regenerates tests/golden/degenerate_knot_fibre.npz -- CPU only, oracle only.

The fibre is column 797 of the operand of the first 1-D prox of iteration 31 of proximal Dykstra (PD2_TV, lambda = 6.0085...) on case
1260 of `tools/fuzz.py 80 111` (400 x 1100, 16-blocks + 0.2 noise).  Late in a Dykstra loop the operand x + p reproduces the previous
result on whole stretches, so the taut string has knots whose jump is EXACTLY zero up to rounding (here between samples 93 and 94):
a bend to one walk, none to another that rounds differently.  Round 5's soak found the hand-over of a repair walk to an unproven
chunk wrong there (rows 94, 95 off by 0.028): profiles/NOTES_r05.md, "session 17".
"""
import os, sys
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import cpu

SIZES = (2, 3, 17, 95, 96, 97, 130, 257, 400, 700, 1100)


def data(rng, kind, shape):   # tools/fuzz.py data(), same draws
    M, N = shape
    if kind == 0: return rng.standard_normal(shape)
    if kind == 1: return np.kron(rng.standard_normal((M // 16 + 1, N // 16 + 1)), np.ones((16, 16)))[:M, :N] + 0.2 * rng.standard_normal(shape)
    if kind == 2: return np.add.outer(np.linspace(-3, 3, M), np.linspace(2, -2, N)) + 0.05 * rng.standard_normal(shape)
    if kind == 3: return np.full(shape, 1.5) + (rng.random(shape) < 0.01) * 8.0
    if kind == 4: return np.cumsum(rng.standard_normal(shape), axis=int(rng.integers(0, 2))) * 0.3
    return np.round(rng.standard_normal(shape) * 3)


def case(seed, index):
    rng = np.random.default_rng(seed)
    for i in range(index + 1):
        M, N = (int(v) for v in rng.choice(list(SIZES), 2))
        X = data(rng, int(rng.integers(0, 6)), (M, N))
        lam = float(10 ** rng.uniform(-3, 2))
        rng.integers(-1, 6); rng.integers(0, 2); rng.integers(0, 3); rng.integers(0, 2); rng.integers(0, 2); rng.integers(0, 3); rng.integers(0, 2)
        what = int(rng.integers(0, 6))
        if what == 1: rng.uniform(0, 2 * lam, (M - 1, N)); rng.uniform(0, 2 * lam, (M, N - 1))
        elif what == 4: rng.integers(1, 40)
        elif what == 5: rng.integers(1, 3)
    return X, lam, what


def main():
    orc = cpu.oracle()
    X, lam, what = case(111, 1260)
    assert X.shape == (400, 1100) and what == 2
    prox = lambda A, axis: np.asfortranarray(np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), lam), axis, A))
    x, p, q = np.asfortranarray(X).copy(), np.zeros_like(X), np.zeros_like(X)
    for k in range(1, 32):
        a_in = x + p
        if k == 31: break
        z = prox(a_in, 0); p = p + (x - z)
        xn = prox(z + q, 1); q = q + (z - xn)
        x = xn
    y = np.ascontiguousarray(a_in[:, 797])
    if cpu.have_reference():   # (pinned: the compiled reference gives the expected values bit for bit -- checked when the fixture was made)
        assert np.array_equal(cpu.reference().tv1_linearized(y, lam), orc.tv1_linearized(y, lam))
    np.savez(os.path.join(HERE, "degenerate_knot_fibre.npz"), y=y, lam=lam, expected=orc.tv1_linearized(y, lam))
    print("samples 92..96:", y[92:97], "lambda", lam)


if __name__ == "__main__":
    main()
