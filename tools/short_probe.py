"""One prox sweep along the LAST dimension of 512x512xL volumes (short fibres: 16 <= L < chunk_min_len) under the three
short-fibre kernels (option "whole": 1 = one block of the chunk kernel, 2 = whole fibre per lane in LDS, 0 = sequential), and
along dimension 0 of L x 512 x 512 (contiguous short fibres)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device
lib = _lib.require_device()
rng = np.random.default_rng(0)
for shape, dim in (((512, 512, 32), 2), ((512, 512, 64), 2), ((512, 512, 90), 2), ((64, 512, 512), 0)):
    V = device.to_colmajor(torch.from_numpy(rng.standard_normal(shape)).cuda())
    out = device.colmajor_empty(shape)
    row, ref = [], {}
    for whole in (2, 1, 0):
        lib.proxtv_set_option(b"whole", whole)
        for lam in (0.1, 1.0):
            device.tv1_fibres(V, lam, dim, out=out); device.tv1_fibres(V, lam, dim, out=out); torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(5): device.tv1_fibres(V, lam, dim, out=out)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
            ref.setdefault(lam, out.clone())
            d = float((out - ref[lam]).abs().max())
            row.append(f"whole={whole} lam={lam}: {dt*1e3:6.3f} ms ({V.numel()*16/dt/8e12:.2f} of HBM peak; d={d:.0e}, fixups {lib.proxtv_last_fixups()})")
    lib.proxtv_set_option(b"whole", 1)
    print(f"{shape} dim {dim}: " + " | ".join(row), flush=True)
