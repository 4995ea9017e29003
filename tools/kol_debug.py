"""Debug aid: first iteration count at which the HIP Kolmogorov2 / CCP loops differ bitwise from the CPU oracle."""
import sys, os, time
sys.path.insert(0, os.getcwd())
import numpy as np
import proxtv_amd as ptv
from oracle import cpu
o = cpu.oracle()
g = np.load("tests/golden/golden_2d_primal_dual.npz")
for name in g["names"][:4]:
    X, lam = g[f"{name}/X"], float(g[f"{name}/lam"])
    first = None
    for it in list(range(1, 12)) + [20, 40, 80, 130, 136, 140]:
        a = ptv.tv1_2d(X, lam, method="kolmogorov", max_iters=it)
        b = o.kolmogorov2(X, lam, it)[0]
        if not np.array_equal(a, b):
            first = (it, float(np.abs(a - b).max()))
            break
    print(name, "kolmogorov first bitwise difference:", first, "ref iters", g[f"{name}/kol_info"][0], flush=True)
    first = None
    for it in list(range(1, 8)) + [20, 40, 80]:
        a = ptv.tv1_2d(X, lam, method="chambolle-pock-acc", max_iters=it)
        b = o.ccp2(X, lam, 2, it)[0]
        if not np.array_equal(a, b):
            first = (it, float(np.abs(a - b).max()))
            break
    print(name, "cp-acc first bitwise difference:", first, flush=True)
rng = np.random.default_rng(0)
bad = 0
for t in range(2000):
    n = int(rng.integers(2, 8)); x = rng.standard_normal(n); lam = float(rng.choice([0.05, 0.1, 0.7, 1.5, 3.0]))
    a = ptv.tv1_1d(x, lam); b = o.tv1_hybrid(x, lam)
    if not np.array_equal(a, b):
        bad += 1
        if bad < 4: print("1-D mismatch", n, lam, x, a - b)
print("1-D short fibres bitwise mismatches:", bad, "of 2000", flush=True)
