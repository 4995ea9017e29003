import os, sys
import numpy as np
sys.path.insert(0, '/root/repo')
import proxtv_amd as ptv
from proxtv_amd import _lib
lib = _lib.require_device()
rng = np.random.default_rng(94)
X = rng.standard_normal((2048, 2048))
for lam in (3.0, 1.0, 0.5, 0.05):
    for k in range(6):
        print(f"--- lam {lam} solve {k}", file=sys.stderr, flush=True)
        ptv.tv1_2d(X, lam)
    print("mode", lib.proxtv_chunk_mode(), file=sys.stderr)
