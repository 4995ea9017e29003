import numpy as np, torch, sys, os
sys.path.insert(0, '/root/repo')
from proxtv_amd import _lib, device
from oracle import cpu
orc = cpu.oracle()
lib = _lib.require_device()
rng = np.random.default_rng(42)
X = rng.standard_normal((700, 900))
xd = device.to_colmajor(torch.from_numpy(X).cuda())
for it in (1, 2, 0):
    want = orc.dr2(X, 1.0, max_iters=it)[0]
    for mode in (0, 1, 2, 3, 5):
        lib.proxtv_set_option(b"chunk_mode", mode)
        w, info = device.tv1_2d(xd, 1.0, max_iters=it)
        err = np.abs(w.cpu().numpy() - want)
        print(os.environ.get("PROXTV_LIB", "main"), "iters", it, "mode", mode, "max err %.3g" % err.max(), "nbad", int((err > 1e-9).sum()), "fixups", lib.proxtv_last_fixups())
