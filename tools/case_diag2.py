#!/usr/bin/env python3
"""Second step of tools/case_diag.py: the plain 1-D prox of the operands it saved, one fibre in isolation, under the pinned rungs and
kernel options.    python tools/case_diag2.py <first_bad.npz> <column> <outdir>"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import proxtv_amd as ptv
from proxtv_amd import _lib
from oracle import cpu

d = np.load(sys.argv[1])
col = int(sys.argv[2])
out = sys.argv[3]
lib = _lib.require_device()
orc = cpu.oracle()
A, lam = np.asfortranarray(d["a_in"]), float(d["lam"])
truth = np.asfortranarray(np.apply_along_axis(lambda f: orc.tv1_hybrid(np.ascontiguousarray(f), lam), 0, A))
defaults = {}


def opt(k, v):
    before = lib.proxtv_set_option(k, v)
    defaults.setdefault(k, before)


def why():
    buf = np.zeros(8, dtype=np.uint32)
    lib.proxtv_debug_why(buf.ctypes.data)
    return buf.tolist()


def run(label, M):
    g = ptv.tvgen(M, [lam], [1], [1])
    t = truth if M.shape == A.shape else np.repeat(truth[:, col:col + 1], M.shape[1], axis=1)
    e = np.abs(g - t)
    rr, cc = np.nonzero(e > 1e-12)
    print(f"{label:60s} error {e.max():.3e} rows {sorted(set(rr.tolist()))[:6]} cols {sorted(set(cc.tolist()))[:6]} why {why()}", flush=True)
    return g


opt(b"why", 1)
opt(b"chunk_mode", 0)
opt(b"deterministic", 0)
opt(b"pin_seed", 0)
opt(b"repair_jobs", 0)
g = run("whole operand, mode 0", A)
np.save(os.path.join(out, "z_gpu_cols.npy"), g[:, max(0, col - 2):col + 3])
one = np.asfortranarray(np.repeat(A[:, col:col + 1], 64, axis=1))
run("the fibre alone x 64, mode 0", one)
run("the fibre alone x 1, mode 0", np.asfortranarray(A[:, col:col + 1]))
near = np.asfortranarray(A[:, col - 29:col + 35])
run("64 columns around it, mode 0", near)
for m in (1, 2, 3, 4, 5, -1):
    opt(b"chunk_mode", m)
    run(f"whole operand, mode {m}", A)
opt(b"chunk_mode", 0)
for k, v in ((b"along", 0), (b"xlink", 0), (b"whole", 0), (b"repair_jobs", 2), (b"tile", 0), (b"replay", 1)):
    before = lib.proxtv_set_option(k, v)
    run(f"whole operand, mode 0, {k.decode()}={v}", A)
    lib.proxtv_set_option(k, before)
# rows as fibres (the strided tile kernel): the transposed operand along dimension 2
At = np.asfortranarray(A.T)
g2 = ptv.tvgen(At, [lam], [2], [1])
e = np.abs(g2.T - truth)
rr, cc = np.nonzero(e > 1e-12)
print(f"transposed, dimension 2, mode 0: error {e.max():.3e} rows {sorted(set(rr.tolist()))[:6]} cols {sorted(set(cc.tolist()))[:6]} why {why()}")
for k, v in defaults.items():
    lib.proxtv_set_option(k, v)
