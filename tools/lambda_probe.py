"""4096^2 DR solves over lambda with the time split per sweep family (option "profile": hipEvents around every sweep), under
the default policy and pinned to chosen rungs -- where each rung stands at each lambda.
    python tools/lambda_probe.py [--modes -1,1,3] [--lams 0.1,0.5,...] [--n 4096] [--opt key=value ...]"""
import argparse, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device

ap = argparse.ArgumentParser()
ap.add_argument("--modes", default="-1")
ap.add_argument("--lams", default="0.1,0.3,0.5,0.7,1.0,3.0,10.0,30.0")
ap.add_argument("--n", type=int, default=4096)
ap.add_argument("--opt", action="append", default=[])
args = ap.parse_args()
lib = _lib.require_device()
for kv in args.opt:
    k, v = kv.split("=")
    lib.proxtv_set_option(k.encode(), int(v))
n = args.n
X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((n, n))).cuda())
out = device.colmajor_empty((n, n))


def timed(fn, reps=3):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


print(f"# DR {n}^2, options {args.opt}")
for mode in [int(m) for m in args.modes.split(",")]:
    lib.proxtv_set_option(b"chunk_mode", mode)
    for lam in [float(a) for a in args.lams.split(",")]:
        device.tv1_2d(X, lam, out=out)
        ms = timed(lambda: device.tv1_2d(X, lam, out=out))
        fx, md = lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()
        lib.proxtv_set_option(b"profile", 1)
        device.tv1_2d(X, lam, out=out)
        fam = [lib.proxtv_last_kernel_ms(f) for f in range(3)]
        cnt = [max(lib.proxtv_last_kernel_launches(f), 1) for f in range(3)]
        lib.proxtv_set_option(b"profile", 0)
        lib.proxtv_set_option(b"why", 1)
        why = np.zeros(8, dtype=np.uint32)
        lib.proxtv_debug_why(why.ctypes.data)
        device.tv1_2d(X, lam, out=out)
        lib.proxtv_debug_why(why.ctypes.data)
        lib.proxtv_set_option(b"why", 0)
        print(f"mode {mode:2d} lambda={lam:<5} {ms:8.2f} ms  fixups {fx:6d}  ran {md}   col {fam[0] / cnt[0] * 1e3:7.1f} us x{cnt[0]}  "
              f"row {fam[1] / cnt[1] * 1e3:7.1f} us x{cnt[1]}  other {fam[2]:.2f} ms  why(off-window, in-wg, x-mismatch, x-late) {why[:4].tolist()}", flush=True)
lib.proxtv_set_option(b"chunk_mode", -1)
