"""Does the policy's seed (sampled edge statistics -> rung) pick a sensible rung on data that are NOT white noise?
DR solves of n x n images of several families at several lambdas: wall time under the default policy against every pinned
rung; a line is flagged when the default is more than 1.5 x the best pinned rung.      python tools/seed_check.py [n]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
rng = np.random.default_rng(3)


def families():
    z = rng.standard_normal((n, n))
    yield "white noise", z
    h = z.copy(); h[:, n // 2:] = 0.0
    yield "left half noise, right half flat", h
    h = z.copy(); h[n // 2:, :] = 1.5
    yield "top half noise, bottom half flat", h
    s = np.zeros((n, n)); m = rng.random((n, n)) < 0.05; s[m] = 10.0 * rng.standard_normal(int(m.sum()))
    yield "5 % spikes on zero", s
    yield "128-blocks + 0.2 noise", np.kron(rng.standard_normal((n // 128, n // 128)), np.ones((128, 128))) + 0.2 * z
    g = np.add.outer(np.linspace(-3, 3, n), np.linspace(0, 2, n))
    yield "smooth ramp + 0.02 noise", g + 0.02 * z
    yield "random-walk texture", np.cumsum(np.cumsum(z, axis=0), axis=1) / n
    yield "checkerboard", np.where((np.add.outer(np.arange(n), np.arange(n)) % 2) == 0, 1.0, -1.0)
    q = z.copy(); q[n // 4: 3 * n // 4, n // 4: 3 * n // 4] *= 0.01
    yield "noise with a quiet centre", q


def timed(fn, reps=2):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); best = min(best, time.perf_counter() - t0)
    return best * 1e3


out = device.colmajor_empty((n, n))
print(f"# DR {n}^2: default policy vs pinned rungs (ms)")
for name, A in families():
    X = device.to_colmajor(torch.from_numpy(np.ascontiguousarray(A)).cuda())
    for lam in (0.1, 0.5, 2.0):
        lib.proxtv_set_option(b"chunk_mode", -1)
        t_def = timed(lambda: device.tv1_2d(X, lam, out=out)); ran = lib.proxtv_chunk_mode()
        ts = {}
        for m in (0, 1, 3):
            lib.proxtv_set_option(b"chunk_mode", m)
            ts[m] = timed(lambda: device.tv1_2d(X, lam, out=out), reps=1)
        lib.proxtv_set_option(b"chunk_mode", -1)
        best = min(ts.values())
        flag = "   <-- default > 1.5 x best" if t_def > 1.5 * best else ""
        print(f"{name:34s} lambda={lam:<4} default {t_def:8.2f} (rung {ran})   pinned 0/1/3: " + " / ".join(f"{ts[m]:8.2f}" for m in (0, 1, 3)) + flag, flush=True)
