#!/usr/bin/env python3
"""CPU only: sweeps END TO END on the host under the device's two roundings, by the thousand (tests/test_sweep_end_to_end_host.py is the
sample of this that the CPU suite runs).  The lanes' own code (tests/host_harness.cpp with -DPTV_TABLE_RECIP: walk_interior, links,
rebuild_owned) leaves outputs, codes and flags; the repair model (tests/repair_model_host.cpp) finishes; the oracle judges.  Fibres: every
third column of the operands of iterations 26 / 31 / 34 of PD2 on the image of round 5's failing soak case, then -- until the time is up --
operands of iterations 20 and 32 of emulated PD2 and DR loops on random images of tools/fuzz.py's families, and fibres built backwards from
solutions full of zero-jump knots.  Geometries: the along-fibre kernel's (64 / 32 / 16 chunks of 17; 64-sample zones: chunks of 31) and the tiles' (8 / 3 chunks of 16, zones of 16 and 64);
sequential and jobs repair; every other fibre with four second-chance rounds inside the blocks (rung 1).

A deviation above 1e-12 (relative to the fibre's largest sample) is reported with where it is: END = inside the fibre's last piece, where the
closed form of the rebuild (free end at height exactly 0) is MORE exact than the reference's last-sample tests (they leave up to EPSILON =
1e-10 in the dual: <= 1e-10 / n on that piece); INSIDE = anywhere else, which must not happen.

    python tools/e2e_host_campaign.py [seconds] [seed]
    python tools/e2e_host_campaign.py --weighted [seconds] [seed]     # weighted DR loops (tv1w_2d): tiles of 16-sample chunks, along-fibre chunks of 9
"""
import os, sys, time
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
from oracle import cpu
import test_sweep_end_to_end_host as e2e
from test_chunk_host import _zero_jump_fibre
import make_degenerate_knot as mk

GEO = ((16, 8, 64), (16, 8, 32), (16, 8, 16), (16, 8, 8), (16, 8, 3), (64, 64, 64), (64, 64, 16), (64, 64, 8))


class Request:
    param = "table reciprocals"


WGEO = ((16, 8, 8), (16, 8, 3), (16, 8, 64), (16, 8, 16))


def main():
    weighted = "--weighted" in sys.argv
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    budget = float(args[0]) if args else 600.0
    rng = np.random.default_rng(int(args[1]) if len(args) > 1 else 1)
    orc = cpu.oracle()
    stage = e2e.stage.__wrapped__(Request) if hasattr(e2e.stage, "__wrapped__") else e2e.stage.__pytest_wrapped__.obj(Request)
    stage[0].host_set_rounds.argtypes = [__import__("ctypes").c_int]
    tally = {"sweeps": 0, "flagged": 0, "end": 0, "inside": 0, "worst end": 0.0, "worst inside": 0.0, "worst": 0.0}

    def one(y, lam, tag):
        y = np.ascontiguousarray(y)
        want = orc.tv1_linearized(y, lam)
        scale = max(1.0, float(np.max(np.abs(y))))
        knots = np.nonzero(np.diff(want) != 0)[0]
        tail = int(knots[-1]) + 1 if knots.size else 0        # first row of the fibre's last piece
        for (H, T, NW) in GEO:
            for which in (0, 1):
                stage[0].host_set_rounds(4 if (tally["sweeps"] // 16) % 2 else 0)   # (every other fibre: second-chance rounds, as on rung 1)
                x, nf, _ = e2e.sweep(stage, y, lam, H, T, NW, tally["sweeps"], which)   # (odd: the op whose output needs the row's own sample)
                d = np.abs(x - want) / scale
                tally["sweeps"] += 1
                tally["flagged"] += nf
                tally["worst"] = max(tally["worst"], float(d.max()))
                if d.max() > 1e-12:
                    inside = float(d[:tail].max()) if tail else 0.0
                    if inside > 1e-12:
                        tally["inside"] += 1
                        tally["worst inside"] = max(tally["worst inside"], inside)
                        print(f"INSIDE {tag} geometry {(H, T, NW)} repair {which}: {inside:.2e} at rows {np.nonzero(d[:tail] > 1e-12)[0][:6]}", flush=True)
                    else:
                        tally["end"] += 1
                        tally["worst end"] = max(tally["worst end"], float(d.max()))

    def one_w(y, w, tag):
        y, w = np.ascontiguousarray(y), np.ascontiguousarray(w)
        want = orc.tv1_weighted(y, w)
        scale = max(1.0, float(np.max(np.abs(y))))
        knots = np.nonzero(np.diff(want) != 0)[0]
        tail = int(knots[-1]) + 1 if knots.size else 0
        for (H, T, NW) in WGEO:
            for which in (0, 1):
                x, nf, _ = e2e.sweep(stage, y, 0.0, H, T, NW, tally["sweeps"], which, w=w)
                d = np.abs(x - want) / scale
                tally["sweeps"] += 1
                tally["flagged"] += nf
                tally["worst"] = max(tally["worst"], float(d.max()))
                if d.max() > 1e-12:
                    inside = float(d[:tail].max()) if tail else 0.0
                    if inside > 1e-12:
                        tally["inside"] += 1
                        tally["worst inside"] = max(tally["worst inside"], inside)
                        print(f"INSIDE {tag} geometry {(H, T, NW)} repair {which}: {inside:.2e} at rows {np.nonzero(d[:tail] > 1e-12)[0][:6]}", flush=True)
                    else:
                        tally["end"] += 1
                        tally["worst end"] = max(tally["worst end"], float(d.max()))

    t0 = time.time()
    if weighted:
        images = 0
        while time.time() - t0 < budget:
            M, N = (int(v) for v in rng.choice([97, 130, 257], 2))
            kind = int(rng.integers(0, 6))
            X = np.asfortranarray(mk.data(rng, kind, (M, N)))
            lam = float(10 ** rng.uniform(-1, 0.7))
            W1, W2 = rng.uniform(0, 2 * lam, (M - 1, N)), rng.uniform(0, 2 * lam, (M, N - 1))
            pc = lambda A: np.asfortranarray(np.stack([orc.tv1_weighted(np.ascontiguousarray(A[:, j]), np.ascontiguousarray(W1[:, j])) for j in range(N)], axis=1))
            pr = lambda A: np.asfortranarray(np.stack([orc.tv1_weighted(np.ascontiguousarray(A[i, :]), np.ascontiguousarray(W2[i, :])) for i in range(M)], axis=0))
            t = np.full_like(X, X.sum() / X.size)
            for k in range(1, 36):
                sp = 2.0 * (t - pc(t)) - t
                v = X - sp
                if k in (2, 20, 32):
                    for j in range(0, N, 5): one_w(t[:, j], W1[:, j], f"weighted DR family {kind} lambda {lam:.3g} iteration {k} column {j}")
                    for i in range(0, M, 5): one_w(v[i, :], W2[i, :], f"weighted DR family {kind} lambda {lam:.3g} iteration {k} row {i}")
                t = 0.5 * t + pr(v) + 0.5 * sp
            s_ = t - pc(t)
            e = np.max(np.abs(pr(X - s_) - orc.dr2w(X, W1, W2)[0])) / max(1.0, np.max(np.abs(X)))
            assert e <= 1e-9, f"weighted DR emulation against the oracle's: {e:.1e}"
            images += 1
        print(f"# weighted: {images} images, {tally['sweeps']} sweeps end to end, {tally['flagged']} flagged chunks handed to the repair model; "
              f"deviations above 1e-12: {tally['end']} inside the fibre's last piece (worst {tally['worst end']:.2e}), {tally['inside']} anywhere else"
              f" (worst {tally['worst inside']:.2e})")
        return 1 if tally["inside"] else 0
    X, lam, _ = mk.case(111, 1260)
    X = np.asfortranarray(X)
    prox = lambda A, axis, l: np.asfortranarray(np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), l), axis, A))
    x, p, q = X.copy(), np.zeros_like(X), np.zeros_like(X)
    for k in range(1, 35):
        a_in = x + p
        z = prox(a_in, 0, lam); p = p + (x - z)
        xn = prox(z + q, 1, lam); q = q + (z - xn)
        x = xn
        if k in (26, 31, 34):
            for j in sorted(set(range(0, a_in.shape[1], 3)) | {232, 797}):
                one(a_in[:, j], lam, f"failing image, PD2 iteration {k}, column {j}")
    print(f"# the failing image: {tally['sweeps']} sweeps, worst deviation {tally['worst']:.2e}", flush=True)
    images = 0
    while time.time() - t0 < budget:
        M, N = (int(v) for v in rng.choice([130, 257, 400], 2))
        kind = int(rng.integers(0, 6))
        X = np.asfortranarray(mk.data(rng, kind, (M, N)))
        lam = float(10 ** rng.uniform(-1, 1))
        x, p, q = X.copy(), np.zeros_like(X), np.zeros_like(X)
        for k in range(1, 33):
            a_in = x + p
            z = prox(a_in, 0, lam); p = p + (x - z)
            b_in = z + q
            xn = prox(b_in, 1, lam); q = q + (z - xn)
            x = xn
            if k in (20, 32):
                for j in range(0, N, 7): one(a_in[:, j], lam, f"PD2 family {kind} lambda {lam:.3g} iteration {k} column {j}")
                for i in range(0, M, 7): one(b_in[i, :], lam, f"PD2 family {kind} lambda {lam:.3g} iteration {k} row {i}")
        t = np.full_like(X, X.sum() / X.size)
        for k in range(1, 33):
            sp = 2.0 * (t - prox(t, 0, lam)) - t
            v = X - sp
            if k in (20, 32):
                for j in range(0, N, 7): one(t[:, j], lam, f"DR family {kind} lambda {lam:.3g} iteration {k} column {j}")
                for i in range(0, M, 7): one(v[i, :], lam, f"DR family {kind} lambda {lam:.3g} iteration {k} row {i}")
            t = 0.5 * t + prox(v, 1, lam) + 0.5 * sp
        for _ in range(40):
            n = int(rng.integers(40, 1500))
            l2 = float(rng.choice([0.05, 0.5, 3.0]) * (0.5 + rng.random()))
            one(_zero_jump_fibre(rng, n, l2)[0], l2, "zero-jump fibre")
        images += 1
    print(f"# {images} random images on top; in all {tally['sweeps']} sweeps end to end, {tally['flagged']} flagged chunks handed to the repair model; "
          f"deviations above 1e-12: {tally['end']} inside the fibre's last piece (worst {tally['worst end']:.2e}), {tally['inside']} anywhere else"
          f" (worst {tally['worst inside']:.2e})")
    return 1 if tally["inside"] else 0


if __name__ == "__main__":
    sys.exit(main())
