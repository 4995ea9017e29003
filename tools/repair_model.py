#!/usr/bin/env python3
"""HOST-ONLY model of the repair stage behind the chunk kernels (no GPU, no oracle): is the sequential repair's jump sound, is the
jobs repair (one walk per failing link, validity decided afterwards) sound, and how much does each leave to the other?

    python tools/repair_model.py [fibres per case] [jobs per fibre at most] [weighted]

Every chunk of 16 samples is its own workgroup here, so every link is a link across workgroups.  Per data family and lambda:
fibres with a link in doubt, links in doubt per such fibre, and for the four repairs (see tests/repair_model_host.cpp) the number of fibres
that end up WRONG (absolute error above 1e-9 against the sequential walk of the whole fibre) -- the point of the exercise.
"""
import ctypes as C, os, subprocess, sys, tempfile
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
so = os.path.join(tempfile.gettempdir(), "repair_model.so")
subprocess.run(["g++", "-O2", "-std=c++17", "-shared", "-fPIC", "-o", so, os.path.join(HERE, "..", "tests", "repair_model_host.cpp")], check=True)
lib = C.CDLL(so)
lib.model_fibres.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]


def lively_flat(rng, n, m, lively=(30, 330), flat=(100, 700)):
    X = np.empty((n, m))
    for f in X:
        k, on = 0, bool(rng.integers(0, 2))
        while k < m:
            span = int(rng.integers(*lively)) if on else int(rng.integers(*flat))
            f[k:k + span] = rng.normal() * 2 + (rng.standard_normal(min(span, m - k)) if on else 0.0)
            k += span
            on = not on
    return X


def families(rng, n, m):
    yield "unit noise", rng.standard_normal((n, m))
    yield "lively 30-330 / flat 100-700", lively_flat(rng, n, m)
    yield "lively 10-60 / flat 20-200", lively_flat(rng, n, m, (10, 60), (20, 200))
    yield "1 % spikes of 8 on a constant", np.full((n, m), 1.5) + (rng.random((n, m)) < 0.01) * 8.0
    yield "blocks of 16 + 0.2 noise", np.repeat(rng.standard_normal((n, m // 16 + 1)), 16, axis=1)[:, :m] + 0.2 * rng.standard_normal((n, m))
    yield "random walk * 0.3", np.cumsum(rng.standard_normal((n, m)), axis=1) * 0.3
    yield "rounded noise (ties)", np.round(rng.standard_normal((n, m)) * 3)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
    max_jobs = int(sys.argv[2]) if len(sys.argv) > 2 else 4      # (the kernel: 4 ; a large number: the validity chain on every fibre)
    weighted = len(sys.argv) > 3 and sys.argv[3] == "weighted"  # per-edge penalties ~ U(0.3, 1.7) lambda
    m, Cn, H = 2048, 16, 16
    rng = np.random.default_rng(7)
    print(f"# {n} fibres of {m} samples per case, chunks of {Cn}, warm-up zones of {H}; jobs: window 128, at most {max_jobs} per fibre{'; penalties per edge ~ U(0.3, 1.7) lambda' if weighted else ''}")
    print(f"# {'family':32s} {'lambda':>6s} {'fibres':>7s} {'links':>6s} {'walks':>6s} | wrong: {'seq old':>7s} {'seq new':>7s} {'jobs':>6s} {'jobs+g':>6s} |"
          f" declined: {'jobs':>6s} {'jobs+g':>6s} | stale reads | worst error old / new / jobs / jobs+g")
    total = np.zeros(12, dtype=np.int64)
    for name, X in families(rng, n, m):
        X = np.ascontiguousarray(X)
        for lam in (0.3, 0.7, 1.0, 1.6, 3.0):
            out = np.zeros(12, dtype=np.int64)
            worst = np.zeros(4)
            Wt = np.ascontiguousarray(rng.uniform(0.3 * lam, 1.7 * lam, (n, m))) if weighted else None
            first = lib.model_fibres(X.ctypes.data, Wt.ctypes.data if weighted else None, n, m, lam, Cn, H, 128, max_jobs, out.ctypes.data, worst.ctypes.data)
            total += out
            print(f"  {name:32s} {lam:6.1f} {out[0]:7d} {out[1] / max(out[0], 1):6.1f} {out[2] / max(out[0], 1):6.1f} |        {out[3]:7d} {out[4]:7d} {out[5]:6d} {out[6]:6d} |"
                  f"           {out[7]:6d} {out[8]:6d} | {out[9]:11d} | {worst[0]:.1e} {worst[1]:.1e} {worst[2]:.1e} {worst[3]:.1e}"
                  + (f"   (first fibre the old scan gets wrong: {first})" if first >= 0 else ""), flush=True)
    print(f"# all cases: fibres with a link in doubt {total[0]}, links {total[1]}; wrong after seq old / seq new / jobs / jobs+guard: "
          f"{total[3]} / {total[4]} / {total[5]} / {total[6]}; stale records read by the old scan: {total[9]}; "
          f"chunks in doubt although their recorded start is the true bend: {total[11]}")


if __name__ == "__main__":
    main()
