#!/usr/bin/env python3
"""A/B matrix on ONE box: every variant (an alternative build of the library and / or a set of knobs) times the same cases,
each variant in its own process, in alternation so that drift of the box hits all of them alike.

    python tools/ab_run.py [--reps R] [--cases c2,c2@0.5,c3,...] name[=lib.so][,KNOB=v,...] ...

    name           label of the variant ; `=path` loads that build instead of proxtv_amd/libproxtv_amd.so
    KNOB=v         proxtv_set_option("knob", v) before anything runs (e.g. tile=0, dr_form=2, chunk_mode=1)

Cases (device-resident, 4096^2 unit noise unless said otherwise):
    c2[@lam]   tv1_2d DR, 35 iterations (default lambda 0.1)      c3[@scale]  weighted DR, w ~ U(0.5, 1.5) * scale * 0.1
    pd2        tv1_2d PD2                                          yang2      tv1_2d Yang
    c4 / c4y   tvgen PD_TV / Yang3 on 512 x 512 x 64               s<N>       tv1_2d DR on an N x N image (small images)
    hard[@lam] tv1_2d DR on 8 x 8 random blocks + 0.2 N(0,1) (default lambda 0.5)
    prox0 / prox1 [@lam]   one OP_PROX sweep along dim 0 / 1      wprox0 / wprox1   weighted
Per case: min and median wall time of the call, and -- from a second pass with option "profile" -- the mean launch time of the
column / row sweep families in microseconds (hipEvent pairs on the library's stream).
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(cases, reps, knobs):
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from proxtv_amd import _lib, device
    lib = _lib.require_device()
    for k, v in knobs.items():
        if lib.proxtv_set_option(k.encode(), int(v)) == -1 and k not in ("verbose",):
            # (-1 is also a legal previous value for chunk_mode)
            if k != "chunk_mode":
                raise SystemExit(f"unknown knob {k}")
    rng = np.random.default_rng(0)
    dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    cache = {}

    def image(n):
        if n not in cache:
            cache[n] = (dev(np.random.default_rng(0).standard_normal((n, n))), device.colmajor_empty((n, n)))
        return cache[n]

    def make(case):
        name, _, arg = case.partition("@")
        lam = float(arg) if arg else 0.1
        if name == "c2":
            X, out = image(4096)
            return lambda: device.tv1_2d(X, lam, out=out), 4096 * 4096
        if name == "c3":
            X, out = image(4096)
            if "w" not in cache:
                r = np.random.default_rng(1)
                cache["w"] = (r.uniform(0.5, 1.5, (4095, 4096)), r.uniform(0.5, 1.5, (4096, 4095)))
            sc = (float(arg) if arg else 1.0) * 0.1
            W1, W2 = dev(cache["w"][0] * sc), dev(cache["w"][1] * sc)
            return lambda: device.tv1w_2d(X, W1, W2, out=out), 4096 * 4096
        if name == "hard":   # bench.py's back-tracking image: 8 x 8 random blocks + 0.2 N(0,1), lambda 0.5 unless said otherwise
            if "hard" not in cache:
                r7 = np.random.default_rng(7)
                cache["hard"] = (dev(np.kron(r7.standard_normal((8, 8)), np.ones((512, 512))) + 0.2 * r7.standard_normal((4096, 4096))),
                                 device.colmajor_empty((4096, 4096)))
            Xh, hout = cache["hard"]
            hl = float(arg) if arg else 0.5
            return lambda: device.tv1_2d(Xh, hl, out=hout), 4096 * 4096
        if name in ("pd2", "yang2"):
            X, out = image(4096)
            return lambda: device.tv1_2d(X, lam, method="pd" if name == "pd2" else "yang", out=out), 4096 * 4096
        if name in ("c4", "c4y"):
            if "v" not in cache:
                cache["v"] = (dev(np.random.default_rng(0).standard_normal((512, 512, 64))), device.colmajor_empty((512, 512, 64)))
            V, vout = cache["v"]
            if name == "c4":
                return lambda: device.tvgen(V, [0.1, 0.1, 0.05], [1, 2, 3], out=vout), 512 * 512 * 64
            return lambda: device.tvgen(V, [0.1, 0.1, 0.1], [1, 2, 3], method="yang", out=vout), 512 * 512 * 64
        if name[0] == "s" and name[1:].isdigit():
            X, out = image(int(name[1:]))
            return lambda: device.tv1_2d(X, lam, out=out), int(name[1:]) ** 2
        if name in ("prox0", "prox1"):
            X, out = image(4096)
            return lambda: device.tv1_fibres(X, lam, int(name[-1]), out=out), 4096 * 4096
        if name in ("wprox0", "wprox1"):
            X, out = image(4096)
            d = int(name[-1])
            W = dev(np.random.default_rng(1).uniform(0.5 * lam, 1.5 * lam, (4095, 4096) if d == 0 else (4096, 4095)))
            return lambda: device.tv1_fibres(X, 0.0, d, weights=W, out=out), 4096 * 4096
        raise SystemExit(f"unknown case {case}")

    res = {}
    for case in cases:
        run, n = make(case)
        for _ in range(2):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter()
            run()
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        fix, mode = lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()
        lib.proxtv_set_option(b"profile", 1)
        fam = [0.0, 0.0, 0.0]
        cnt = [0, 0, 0]
        for _ in range(2):
            run()
            for f in range(3):
                fam[f] += lib.proxtv_last_kernel_ms(f)
                cnt[f] += lib.proxtv_last_kernel_launches(f)
        lib.proxtv_set_option(b"profile", 0)
        res[case] = {"min": min(ts), "med": sorted(ts)[len(ts) // 2], "fix": fix, "mode": mode,
                     "us": [1e3 * fam[f] / cnt[f] if cnt[f] else 0.0 for f in range(3)], "n": [c // 2 for c in cnt]}
    print("AB_RESULT " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=7)
    ap.add_argument("--rounds", type=int, default=2, help="passes over the list of variants (alternation)")
    ap.add_argument("--cases", default="c2")
    ap.add_argument("--child", default=None)
    ap.add_argument("variants", nargs="*")
    a = ap.parse_args()
    cases = a.cases.split(",")
    if a.child is not None:
        child(cases, a.reps, json.loads(a.child))
        return
    variants = []
    for spec in a.variants:
        head, *kn = spec.split(",")
        name, _, lib = head.partition("=")
        variants.append((name, lib, dict(k.split("=") for k in kn)))
    results = {name: [] for name, _, _ in variants}
    for _ in range(a.rounds):
        for name, lib, knobs in variants:
            env = dict(os.environ)
            if lib:
                env["PROXTV_DEBUG_ALT_LIB"], env["PROXTV_LIB"] = "1", os.path.join(ROOT, lib) if not os.path.isabs(lib) else lib
            p = subprocess.run([sys.executable, os.path.abspath(__file__), "--reps", str(a.reps), "--cases", a.cases, "--child", json.dumps(knobs)],
                               env=env, capture_output=True, text=True)
            line = [l for l in p.stdout.splitlines() if l.startswith("AB_RESULT ")]
            if p.returncode != 0 or not line:
                print(f"# {name}: FAILED rc={p.returncode}\n{p.stdout[-600:]}\n{p.stderr[-1200:]}", flush=True)
                continue
            results[name].append(json.loads(line[0][len("AB_RESULT "):]))
    print(f"{'case':12s} {'variant':22s} {'min ms':>9s} {'median':>9s} {'col us':>8s} {'row us':>8s} {'other':>8s} {'launches c/r/o':>15s} {'fixups':>7s} {'mode':>4s}")
    for case in cases:
        for name, _, _ in variants:
            rs = [r[case] for r in results[name] if case in r]
            if not rs:
                continue
            best = min(rs, key=lambda r: r["min"])
            med = sorted(r["med"] for r in rs)[len(rs) // 2]
            us = [min(r["us"][f] for r in rs) for f in range(3)]
            print(f"{case:12s} {name:22s} {best['min']:9.3f} {med:9.3f} {us[0]:8.1f} {us[1]:8.1f} {us[2]:8.1f} "
                  f"{'/'.join(str(v) for v in best['n']):>15s} {best['fix']:7d} {best['mode']:4d}")


if __name__ == "__main__":
    main()
