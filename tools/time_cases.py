#!/usr/bin/env python3
"""Wall-clock of the BASELINE.json configurations (and a back-tracking-heavy image) on the GPU, device-resident,
with the number of fibres the chunk kernel had to repair.  Not the headline benchmark (bench.py); a survey tool.

    python tools/time_cases.py
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from proxtv_amd import _lib, device  # noqa: E402


def timed(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best * 1e3


def main():
    lib = _lib.require_device()
    rng = np.random.default_rng(0)

    def dev(a):
        return device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())

    X = dev(rng.standard_normal((4096, 4096)))
    out = device.colmajor_empty((4096, 4096))
    rows = []
    rows.append(("C2  tv1_2d DR 4096^2 lam=.1", timed(lambda: device.tv1_2d(X, 0.1, out=out)), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    W1, W2 = dev(rng.uniform(0.05, 0.15, (4095, 4096))), dev(rng.uniform(0.05, 0.15, (4096, 4095)))
    rows.append(("C3  tv1w_2d weighted DR 4096^2", timed(lambda: device.tv1w_2d(X, W1, W2, out=out)), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    rows.append(("    tv1_2d PD2 4096^2 lam=.1", timed(lambda: device.tv1_2d(X, 0.1, method="pd", out=out)), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    rows.append(("    tv1_2d Yang2 4096^2 lam=.1", timed(lambda: device.tv1_2d(X, 0.1, method="yang", out=out)), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    del W1, W2
    for m, its in (("kolmogorov", 100), ("condat", 100), ("chambolle-pock-acc", 100)):
        rows.append((f"    tv1_2d {m} 4096^2, {its} its", timed(lambda: device.tv1_2d(X, 0.1, method=m, max_iters=its, out=out), reps=1), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    V = dev(rng.standard_normal((512, 512, 64)))
    vout = device.colmajor_empty((512, 512, 64))
    rows.append(("C4  tvgen PD_TV 512x512x64", timed(lambda: device.tvgen(V, [0.1, 0.1, 0.05], [1, 2, 3], out=vout)), 512 * 512 * 64, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    rows.append(("C4' Yang 512x512x64 lam=.1", timed(lambda: device.tvgen(V, [0.1, 0.1, 0.1], [1, 2, 3], method="yang", out=vout)), 512 * 512 * 64, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    del V, vout
    B = 16
    S = dev(np.stack([np.random.default_rng(k).standard_normal((2048, 2048)) for k in range(B)], axis=2))
    sout = device.colmajor_empty((2048, 2048, B))
    rows.append((f"C5  batch {B} x 2048^2 DR", timed(lambda: device.tv1_2d_batch(S, 0.1, out=sout), reps=2), B * 2048 * 2048, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    del S, sout
    r7 = np.random.default_rng(7)
    Xh = dev(np.kron(r7.standard_normal((8, 8)), np.ones((128, 128))) + 0.2 * r7.standard_normal((1024, 1024)))
    hout = device.colmajor_empty((1024, 1024))
    rows.append(("hard blocks+noise 1024^2 DR lam=.5", timed(lambda: device.tv1_2d(Xh, 0.5, out=hout)), 1024 * 1024, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    Xl = dev(rng.standard_normal((4096, 4096)))
    rows.append(("    DR 4096^2 lam=1 (long pieces)", timed(lambda: device.tv1_2d(Xl, 1.0, out=out), reps=1), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    x1 = dev(rng.standard_normal((1_000_000,)))
    o1 = device.colmajor_empty((1_000_000,))
    rows.append(("C1  tv1 fibre 1e6 lam=.5 (device)", timed(lambda: device.tv1_fibres(x1, 0.5, 0, out=o1)), 1_000_000, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    for dim in (0, 1):
        rows.append((f"    sweep OP_PROX 4096^2 dim {dim} lam=.1", timed(lambda: device.tv1_fibres(X, 0.1, dim, out=out), reps=5), 4096 * 4096, lib.proxtv_last_fixups(), lib.proxtv_chunk_mode()))
    print(f"{'case':44s} {'ms':>10s} {'Melem/s':>10s} {'fixups':>8s} {'mode':>5s}")
    for name, ms, n, fx, mode in rows:
        print(f"{name:44s} {ms:10.2f} {n / ms / 1e3:10.1f} {fx:8d} {mode:5d}")


if __name__ == "__main__":
    main()
