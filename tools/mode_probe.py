#!/usr/bin/env python3
"""Time one DR solve (4096^2 N(0,1)) for a given lambda under each pinned chunk-geometry mode; prints repairs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
lam = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
if len(sys.argv) > 3: lib.proxtv_set_option(b"rounds", int(sys.argv[3]))
X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
out = device.colmajor_empty((4096, 4096))
ref = None
for mode in [int(m) for m in (sys.argv[2].split(',') if len(sys.argv) > 2 else '5,4,3,2,1,0,-1'.split(','))]:
    lib.proxtv_set_option(b"chunk_mode", mode)
    device.tv1_2d(X, lam, out=out); torch.cuda.synchronize()
    t0 = time.perf_counter(); device.tv1_2d(X, lam, out=out); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    lib.proxtv_set_option(b"profile", 1); device.tv1_2d(X, lam, out=out); torch.cuda.synchronize(); lib.proxtv_set_option(b"profile", 0)
    fam = [lib.proxtv_last_kernel_ms(k) for k in range(3)]
    o = out.clone()
    if ref is None: ref = o
    print(f"lam={lam} mode={mode:2d}: {dt*1e3:9.2f} ms  fibres repaired={lib.proxtv_last_fixups():7d}  policy now={lib.proxtv_chunk_mode()}  max|diff vs first|={float((o-ref).abs().max()):.2e}  col/row/other ms = {fam[0]:.1f}/{fam[1]:.1f}/{fam[2]:.1f}")
