"""Time one prox sweep along the LAST dimension of 512x512xL volumes (short fibres) for a few chunk_min_len settings."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(0)
for L in (32, 64, 90):
    V = device.to_colmajor(torch.from_numpy(rng.standard_normal((512, 512, L))).cuda())
    out = device.colmajor_empty((512, 512, L))
    row = []
    ref = None
    for ml in (256,):
        lib.proxtv_set_option(b"chunk_min_len", ml)
        for lam in (0.1, 1.0, 5.0):
            device.tv1_fibres(V, lam, 2, out=out); device.tv1_fibres(V, lam, 2, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): device.tv1_fibres(V, lam, 2, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            key = (lam,)
            if ml == 256: ref = ref or {}; ref[key] = out.clone()
            d = float((out - ref[key]).abs().max())
            row.append(f"min_len={ml} lam={lam}: {dt*1e3:7.3f} ms (d={d:.0e}, mode {lib.proxtv_chunk_mode()})")
    print(f"L={L:4d} ({512*512*L/1e6:.1f} Msamples)  " + "  ".join(row), flush=True)
