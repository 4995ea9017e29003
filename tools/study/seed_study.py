#!/usr/bin/env python3
"""HOST-ONLY design study (no GPU, no oracle): temporal seeding of chunk walks inside a DR solve.

A lane that finds no bend known a priori before its chunk starts from a free end 16 samples back today.  Inside a splitting loop
it could start AT the bend the previous iteration had there (the chunk kernels already publish it).  This prices the idea on the
inputs of consecutive DR iterations (unit noise): trips per lane and per WAVE (64 consecutive chunks; a wave walks as long as its
slowest lane, and every round of second chances costs it one more chunk walk), and how many links fail.

    python tools/study/seed_study.py [lambda ...]
"""
import ctypes as C, os, sys
import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import links_study as L

lib = L.lib
lib.study_seeded.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]


def dr_pairs(U, lam, at):
    """(previous, current) inputs of the column sweeps (axis 0) and of the row sweeps (axis 1) at the given iterations"""
    t = np.full(U.shape, 2 * U.mean())
    cols, rows = {}, {}
    tp = vp = None
    for it in range(max(at) + 1):
        if it in at and tp is not None: cols[it] = (tp, t.copy())
        tp = t.copy()
        s = t - L.prox_along(t, lam, 0)
        sp = 2 * s - t
        v = U - sp
        if it in at and vp is not None: rows[it] = (vp, v.copy())
        vp = v.copy()
        t = 0.5 * (t + (sp + 2 * L.prox_along(v, lam, 1)))
    return cols, rows


def study(prev, cur, lam, Cn=17, H=16, look=14, back=64):
    n = cur.shape[1]
    nch = n // Cn
    rec = []
    for fp, f in zip(np.ascontiguousarray(prev), np.ascontiguousarray(cur)):
        out = np.zeros((nch, 8), dtype=np.int32)
        lib.study_seeded(fp.ctypes.data, f.ctypes.data, n, lam, Cn, H, look, back, out.ctypes.data)
        ok = np.flatnonzero(out[:, 0] == 1)
        ok = ok[: (len(ok) // 64) * 64]
        rec.append(out[ok].reshape(-1, 64, 8))
    return np.concatenate(rec)


def wave_cost(W):
    old, new, l_old, l_new, certain, inner, seeded = (W[:, :, k] for k in range(1, 8))
    c_old, left_old = L.rounds_cost(l_old == 0, inner)
    c_new, left_new = L.rounds_cost(l_new == 0, inner)
    return dict(lane_old=old.mean(), lane_new=new.mean(), wave_old=(old.max(axis=1) + c_old).mean(), wave_new=(new.max(axis=1) + c_new).mean(),
                fail_old=(l_old == 0).mean(), fail_new=(l_new == 0).mean(), certain=certain.mean(), seeded=seeded.mean(),
                left_old=left_old.mean(), left_new=left_new.mean())


if __name__ == "__main__":
    lams = [float(v) for v in sys.argv[1:]] or [0.4, 0.5, 0.6, 0.7, 0.8]
    its = (2, 3, 5, 8, 12, 16, 20, 25, 30, 34)
    rng = np.random.default_rng(5)
    Ucol, Urow = rng.standard_normal((2400, 96)), rng.standard_normal((96, 2400))
    print("# per WAVE = 64 consecutive 17-sample chunks: slowest lane + one chunk walk per round of second chances; 14-edge look-back for bends known a priori")
    for lam in lams:
        cols, _ = dr_pairs(Ucol, lam, its)
        _, rows = dr_pairs(Urow, lam, its)
        tot = {"col": [0.0, 0.0], "row": [0.0, 0.0]}
        print(f"lambda = {lam}")
        for it in its:
            for name, pairs, ax in (("col", cols, 0), ("row", rows, 1)):
                p, c = pairs[it]
                if ax == 0: p, c = p.T, c.T
                r = wave_cost(study(p, c, lam))
                # weight: the iterations between two sampled ones count like the later one
                tot[name][0] += r["wave_old"]; tot[name][1] += r["wave_new"]
                print(f"  it {it:2d} {name}: certain {r['certain']:5.1%} seeded-available {r['seeded']:5.1%} | lane {r['lane_old']:5.1f} -> {r['lane_new']:5.1f} |"
                      f" wave {r['wave_old']:5.1f} -> {r['wave_new']:5.1f} ({r['wave_new'] / r['wave_old'] - 1:+.0%}) | links failing {r['fail_old']:.2%} -> {r['fail_new']:.2%}"
                      f" | waves left to the repair kernel {r['left_old']:.1%} -> {r['left_new']:.1%}")
        for name in ("col", "row"):
            print(f"  sum over the sampled iterations, {name}: {tot[name][0]:.0f} -> {tot[name][1]:.0f} wave trips ({tot[name][1] / tot[name][0] - 1:+.1%})")
