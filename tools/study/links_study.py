#!/usr/bin/env python3
"""HOST-ONLY design study (no GPU, no oracle): would chunk links proven by comparing walker STATES at the chunk boundary --
instead of codes of bends the lane must first walk on to detect -- save trips, and how often would they fail?

    python tools/study/links_study.py [lambda ...]

Data: the inputs of the column sweeps (t) and row sweeps (U - s') of DR2 solves on unit noise, generated here with the host
build of the device walker; fibres of 4500 samples, chunks of 17, zones of 16, 64 consecutive chunks = one wave."""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), "links_study.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "links_study.cpp")], check=True)
lib = C.CDLL(so)
lib.study_fibre.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.study_prox.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p]
lib.study_bends.argtypes = [C.c_void_p, C.c_int, C.c_double, C.c_void_p, C.c_void_p, C.c_int]


def prox_along(A, lam, axis):
    A = np.ascontiguousarray(np.moveaxis(A, axis, -1))
    out = np.empty_like(A)
    for j in range(A.shape[0]):
        lib.study_prox(A[j].ctypes.data, A.shape[1], lam, out[j].ctypes.data)
    return np.moveaxis(out, -1, axis)


def dr_inputs(U, lam, at=(3, 12, 30)):
    """the arrays the column sweeps (axis 0) and the row sweeps (axis 1) of a DR solve see at the given iterations"""
    t = np.full(U.shape, 2 * U.mean())
    cols, rows = [], []
    for it in range(max(at) + 1):
        if it in at: cols.append(t.copy())
        s = t - prox_along(t, lam, 0)
        sp = 2 * s - t
        v = U - sp
        if it in at: rows.append(v.copy())
        t = 0.5 * (t + (sp + 2 * prox_along(v, lam, 1)))
    return cols, rows


def study(fibres, lam, Cn=17, H=16, look=14):
    """fibres: (count, len) array, one fibre per row"""
    n = fibres.shape[1]
    nch = n // Cn
    rec = []
    for f in np.ascontiguousarray(fibres):
        out = np.zeros((nch, 8), dtype=np.int32)
        lib.study_fibre(f.ctypes.data, n, lam, Cn, H, look, out.ctypes.data)
        ok = np.flatnonzero(out[:, 0] == 1)
        ok = ok[: (len(ok) // 64) * 64]              # whole waves of 64 consecutive chunks
        rec.append(out[ok].reshape(-1, 64, 8))
    return np.concatenate(rec)                       # (waves, 64, 8)


def rounds_cost(fail, chunk_trips):
    """second chances: every round, the failing lanes whose predecessor holds walk their chunk again (a wave's round costs its
    slowest such lane); a run of k consecutive failing lanes takes k rounds"""
    fail = fail.copy()
    cost = np.zeros(fail.shape[0])
    for _ in range(8):
        prev_ok = np.concatenate([np.ones((fail.shape[0], 1), bool), ~fail[:, :-1]], axis=1)
        now = fail & prev_ok
        if not now.any(): break
        cost += np.where(now, chunk_trips, 0).max(axis=1)
        fail &= ~now
    return cost, fail.any(axis=1)


def report(name, W):
    old, new, l_old, l_new, certain, inner, over = (W[:, :, k] for k in range(1, 8))
    f_old, f_new = l_old == 0, l_new == 0
    c_old, left_old = rounds_cost(f_old, inner)
    c_new, left_new = rounds_cost(f_new, inner)
    t_old, t_new = old.max(axis=1) + c_old, new.max(axis=1) + c_new
    print(f"  {name:34s} certain starts {certain.mean():5.1%} | trips per lane (mean) {old.mean():5.1f} -> {new.mean():5.1f} |"
          f" per WAVE: walk {old.max(axis=1).mean():5.1f} -> {new.max(axis=1).mean():5.1f}, with second chances {t_old.mean():5.1f} -> {t_new.mean():5.1f}"
          f" ({t_new.mean() / t_old.mean() - 1:+.0%}) | links failing {f_old.mean():.2%} -> {f_new.mean():.2%},"
          f" waves with a failure {f_old.any(axis=1).mean():.0%} -> {f_new.any(axis=1).mean():.0%}, left to the repair kernel {left_old.mean():.1%} -> {left_new.mean():.1%}"
          f" | samples past the chunk end today: mean {over.mean():.1f}, wave max {over.max(axis=1).mean():.1f}")


def pool_model(arrays, lam, seg=1088, overhead=2):
    """A wave-wide pool instead of one chunk per lane: the stretches between consecutive bends known a priori are independent
    problems (the string is pinned at both ends); 64 lanes draw them in fibre order, `overhead` trips per draw (closed-form
    restart, bookkeeping).  Returns (trips per sample of the sequential walk, mean makespan of a 1088-sample segment)."""
    import heapq
    spans, seq = [], []
    for a in arrays:
        for f in np.ascontiguousarray(a):
            n = f.size
            at, tr = np.zeros(n, np.int32), np.zeros(n, np.int64)
            nb = lib.study_bends(f.ctypes.data, n, lam, at.ctypes.data, tr.ctypes.data, n)
            at, tr = at[:nb], tr[:nb]
            known = np.flatnonzero(np.abs(np.diff(f)) > 4.0000001 * lam) + 1
            keep = np.isin(at, known)
            ca, ct = at[keep], tr[keep]
            for s0 in range(seg, n - 2 * seg, seg):
                m = (ca >= s0) & (ca < s0 + seg)
                if m.sum() < 3: continue
                cost = np.diff(ct[m])
                lanes = [0] * 64
                for c in cost:
                    heapq.heappush(lanes, heapq.heappop(lanes) + int(c) + overhead)
                spans.append(max(lanes))
                seq.append(cost.sum() / (ca[m][-1] - ca[m][0]))
    return float(np.mean(seq)), float(np.mean(spans))


if __name__ == "__main__":
    lams = [float(v) for v in sys.argv[1:]] or [0.1, 0.3, 0.4, 0.5, 0.6, 0.7]
    rng = np.random.default_rng(5)
    Ucol, Urow = rng.standard_normal((4500, 160)), rng.standard_normal((160, 4500))
    for lam in lams:
        print(f"lambda = {lam}")
        cols, _ = dr_inputs(Ucol, lam)
        _, rows = dr_inputs(Urow, lam)
        for look in (8, 14):
            report(f"column inputs, look-back {look}", np.concatenate([study(a.T, lam, look=look) for a in cols]))
            report(f"row inputs,    look-back {look}", np.concatenate([study(a, lam, look=look) for a in rows]))
        for oh in (2, 0):
            seq, span = pool_model([a.T for a in cols], lam, overhead=oh)
            print(f"  pool model, column inputs, {oh} trips per draw: sequential walk {seq:.2f} trips per sample; a wave-wide pool of the stretches"
                  f" between a-priori bends finishes a 1088-sample segment in {span:.1f} trips = {span / 17:.2f} per sample")
