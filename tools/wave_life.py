#!/usr/bin/env python3
"""Along-fibre column sweep of the headline (4096^2, lambda 0.1): how long does a wave live, by the segment it takes (option "trace"),
and how long does its workgroup hold its slot?  python tools/wave_life.py"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
x = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
y = device.colmajor_empty((4096, 4096))
for runs in (1, 0):
    lib.proxtv_set_option(b"runs", runs)
    device.tv1_fibres(x, 0.1, 0, out=y)
    lib.proxtv_set_option(b"trace", 1)
    device.tv1_fibres(x, 0.1, 0, out=y)
    buf = np.zeros((32768, 8), dtype=np.uint64)
    n = lib.proxtv_debug_trace(buf.ctypes.data, 32768)
    lib.proxtv_set_option(b"trace", 0)
    t = (buf[:n, 1:6].astype(np.int64) - int(buf[:n, 1].min())) * 0.01
    life = t[:, 4] - t[:, 0]
    seg = np.arange(n) % 4
    wg = np.arange(n) // 4
    print(f"runs={runs}: {n} waves, kernel span {t[:, 4].max():.1f} us; mean life {life.mean():.2f} us; by segment: " +
          " ".join(f"{life[seg == k].mean():.2f}" for k in range(4)))
    ph = np.diff(t, axis=1)
    for k in range(4):
        print(f"   segment {k}: stage {ph[seg == k, 0].mean():.2f} walk {ph[seg == k, 1].mean():.2f} rebuild {ph[seg == k, 2].mean():.2f} out {ph[seg == k, 3].mean():.2f}")
    wg_start = np.array([t[wg == w, 0].min() for w in range(n // 4)])
    wg_end = np.array([t[wg == w, 4].max() for w in range(n // 4)])
    held = (wg_end - wg_start)
    print(f"   a workgroup holds its slot {held.mean():.2f} us on average; its waves live {life.mean():.2f}: {100 * (1 - life.mean() / held.mean()):.0f} % of a held wave slot idles inside the workgroup")
    print(f"   wave-time {life.sum() / 1e3:.1f} ms.wave, slot-time held {4 * held.sum() / 1e3:.1f} ms.wave, slots x span {4096 * t[:, 4].max() / 1e3:.1f} ms.wave")
