#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the compiled reference (oracle/_ref/libproxtv_ref.so).

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference to have built
oracle/_ref via `make -C oracle ref`).  The fixtures are pure data: seeded inputs and the outputs /
info[] / return codes the reference's own C entry points produced for them.  The reference has no golden
vectors of its own (prox_tv/prox_tv_test.py holds only unseeded cross-method consistency tests), so these
pin both the oracle restatement and the HIP path.

Usage:  python oracle/gen_golden.py [--large]     (--large adds the BASELINE.json full-size cases)
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import cpu  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")


def signals_1d(rng):
    """name -> (x, lambda).  Mix of the reference tests' regimes and the edge cases it lacks."""
    cases = {}
    # known answers (all reference solvers agree)
    cases["ka_single"] = (np.array([3.0]), 1.0)
    cases["ka_pair"] = (np.array([1.0, 5.0]), 1.0)
    cases["ka_pair_big"] = (np.array([1.0, 5.0]), 10.0)
    cases["ka_const"] = (np.array([2.0, 2.0, 2.0, 2.0]), 0.5)
    cases["ka_lam0"] = (np.array([1.0, 4.0, 2.0, 8.0, 3.0]), 0.0)
    cases["ka_zigzag"] = (np.array([0.0, 10.0, 0.0, 10.0, 0.0]), 2.0)
    for n in (2, 3, 5, 17, 64, 65, 100, 257, 1000):
        cases[f"randn_n{n}"] = (rng.standard_normal(n), 0.5)
    cases["randn_small_lam"] = (rng.standard_normal(300), 0.01)
    cases["randn_huge_lam"] = (rng.standard_normal(300), 1e4)
    cases["scaled_like_ref_tests"] = (100 * rng.standard_normal(29), 13.7)  # prox_tv_test.py:37-44
    cases["int_input"] = ((100 * rng.standard_normal(23)).astype("int").astype(float), 7.3)  # :47-53
    blocks = np.repeat(rng.standard_normal(12), 40) + 0.2 * rng.standard_normal(480)
    cases["blocks_noise"] = (blocks, 0.5)          # heavy back-tracking regime
    cases["ramp"] = (np.linspace(-3, 3, 400), 0.7)
    cases["walk"] = (np.cumsum(rng.standard_normal(500)), 2.0)
    cases["neg_lambda"] = (rng.standard_normal(40), -0.3)  # prox_tv_test.py:202-209 feeds negative weights
    return cases


def gen_1d(ref):
    rng = np.random.default_rng(20260926)
    out = {}
    names = []
    for name, (x, lam) in signals_1d(rng).items():
        names.append(name)
        out[f"{name}/x"] = x
        out[f"{name}/lam"] = np.float64(lam)
        out[f"{name}/hybrid"] = ref.tv1_hybrid(x, lam)
        out[f"{name}/hybrid_1p2"] = ref.tv1_hybrid(x, lam, 1.2)   # prox_tv_test.py:61
        out[f"{name}/hybrid_0p5"] = ref.tv1_hybrid(x, lam, 0.5)   # forces the classic switch early
        out[f"{name}/linearized"] = ref.tv1_linearized(x, lam)
        if lam >= 0:
            out[f"{name}/classic"] = ref.tv1_classic(x, lam)
            out[f"{name}/condat"] = ref.tv1_condat(x, lam)
        if x.size >= 2 and lam >= 0:
            w = rng.uniform(0.0, 2 * lam + 0.05, x.size - 1)
            out[f"{name}/w"] = w
            out[f"{name}/weighted"] = ref.tv1_weighted(x, w)
            wu = np.full(x.size - 1, lam)
            out[f"{name}/weighted_uniform"] = ref.tv1_weighted(x, wu)   # prox_tv_test.py:26-34
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "golden_1d.npz"), **out)
    print("golden_1d:", len(names), "cases")


def gen_2d(ref):
    rng = np.random.default_rng(20260927)
    out = {}
    names = []
    shapes = [(2, 2), (2, 3), (3, 2), (5, 7), (16, 16), (24, 31), (64, 48), (65, 130)]
    for (M, N) in shapes:
        for lam in (0.1, 1.5):
            name = f"randn_{M}x{N}_l{lam}"
            names.append(name)
            X = np.asfortranarray(rng.standard_normal((M, N)))
            out[f"{name}/X"] = X
            out[f"{name}/lam"] = np.float64(lam)
            y, info, rc = ref.dr2(X, lam)
            out[f"{name}/dr2"], out[f"{name}/dr2_info"], out[f"{name}/dr2_rc"] = y, info, np.int64(rc)
            y, info, rc = ref.dr2(X, lam, max_iters=7)
            out[f"{name}/dr2_it7"], out[f"{name}/dr2_it7_info"] = y, info
            y, info, rc = ref.dr2(X, lam, 0.5 * lam)     # different penalties along columns / rows (tvp_2d path)
            out[f"{name}/dr2_aniso"] = y
            W1 = np.asfortranarray(rng.uniform(0.0, 2 * lam, (M - 1, N)))
            W2 = np.asfortranarray(rng.uniform(0.0, 2 * lam, (M, N - 1)))
            out[f"{name}/W1"], out[f"{name}/W2"] = W1, W2
            y, info, rc = ref.dr2w(X, W1, W2)
            out[f"{name}/dr2w"], out[f"{name}/dr2w_info"], out[f"{name}/dr2w_rc"] = y, info, np.int64(rc)
            y, info, rc, _ = ref.pd2(X, [lam, lam], [1, 2])              # tv1_2d(method='pd') and 2-penalty tvgen
            out[f"{name}/pd2"], out[f"{name}/pd2_info"], out[f"{name}/pd2_rc"] = y, info, np.int64(rc)
            y, info, rc, _ = ref.pd2(X, [lam, lam], [1, 2], max_iters=3)
            out[f"{name}/pd2_it3"], out[f"{name}/pd2_it3_info"] = y, info
            y, info, rc, _ = ref.pd2(X, [lam], [2])                       # single penalty along rows
            out[f"{name}/pd2_single"], out[f"{name}/pd2_single_info"] = y, info
            y, info, rc, _ = ref.pd(X, [lam], [1])                        # 1-penalty tvgen -> PD_TV
            out[f"{name}/pd_single"], out[f"{name}/pd_single_info"] = y, info
            y, info, rc = ref.yang2(X, lam)
            out[f"{name}/yang2"], out[f"{name}/yang2_info"], out[f"{name}/yang2_rc"] = y, info, np.int64(rc)
    # the reference's only fixed-input test (prox_tv_test.py:169-178): integer weight arrays
    a = -np.array([[1, 2, 3], [4, 5, 6], [7, 8, 9]]) / 10.0
    out["emengd/X"] = np.asfortranarray(a)
    out["emengd/dr2w_it100"] = ref.dr2w(a, np.ones((2, 3)), np.ones((3, 2)), max_iters=100)[0]
    out["emengd/dr2"] = ref.dr2(a, 1.0)[0]
    # heavy back-tracking image
    blocks = np.kron(rng.standard_normal((6, 5)), np.ones((16, 16))) + 0.2 * rng.standard_normal((96, 80))
    out["blocks/X"] = np.asfortranarray(blocks)
    out["blocks/dr2_l0p5"] = ref.dr2(blocks, 0.5)[0]
    out["blocks/pd2_l0p5"], out["blocks/pd2_l0p5_info"] = ref.pd2(blocks, [0.5, 0.5], [1, 2])[:2]
    # several penalties on the same dimension (prox_tv_test.py:212-226): tvgen -> PD_TV with 5 terms
    Xm = np.asfortranarray(100 * rng.standard_normal((14, 19)))
    w = 9.3
    out["multireg/X"] = Xm
    out["multireg/lams"] = np.array([w / 2, w / 2, w / 3, w / 3, w / 3])
    out["multireg/dims"] = np.array([1, 1, 2, 2, 2])
    y, info, rc, lam_after = ref.pd(Xm, out["multireg/lams"], out["multireg/dims"], max_iters=1000)
    out["multireg/pd_it1000"], out["multireg/pd_it1000_info"], out["multireg/lams_after"] = y, info, lam_after
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "golden_2d.npz"), **out)
    print("golden_2d:", len(names), "cases (+ emengd, blocks, multireg)")


def gen_1d_other_methods(ref):
    """What the reference returns for tv1_1d's other method names ('pn', 'kolmogorov', 'condattautstring', 'dp'):
    the HIP surface serves them with the exact solver, these vectors show how close that is."""
    rng = np.random.default_rng(20260929)
    out = {}
    names = []
    cases = {"ka_zigzag": (np.array([0.0, 10.0, 0.0, 10.0, 0.0]), 2.0), "ka_pair": (np.array([1.0, 5.0]), 1.0)}
    for n in (3, 17, 100, 257, 1000):
        cases[f"randn_n{n}"] = (rng.standard_normal(n), 0.5)
    cases["scaled_like_ref_tests"] = (100 * rng.standard_normal(29), 13.7)
    cases["blocks_noise"] = (np.repeat(rng.standard_normal(12), 40) + 0.2 * rng.standard_normal(480), 0.5)
    cases["walk"] = (np.cumsum(rng.standard_normal(500)), 2.0)
    for name, (x, lam) in cases.items():
        names.append(name)
        out[f"{name}/x"], out[f"{name}/lam"] = x, np.float64(lam)
        out[f"{name}/hybrid"] = ref.tv1_hybrid(x, lam)
        for m in ("pn", "kolmogorov", "condattautstring", "dp"):
            out[f"{name}/{m}"] = ref.tv1_other_method(x, lam, m)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "golden_1d_other_methods.npz"), **out)
    print("golden_1d_other_methods:", len(names), "cases")


def gen_2d_primal_dual(ref):
    """Kolmogorov2_TV and CondatChambollePock2_TV (the other tv1_2d methods): full 2500-iteration runs and truncated ones."""
    rng = np.random.default_rng(20260928)
    out = {}
    names = []
    for (M, N) in [(2, 2), (2, 3), (3, 2), (5, 7), (16, 16), (24, 31), (64, 48), (65, 130)]:
        for lam in (0.1, 1.5):
            name = f"randn_{M}x{N}_l{lam}"
            names.append(name)
            X = np.asfortranarray(rng.standard_normal((M, N)))
            out[f"{name}/X"] = X
            out[f"{name}/lam"] = np.float64(lam)
            y, info, rc = ref.kolmogorov2(X, lam)
            out[f"{name}/kol"], out[f"{name}/kol_info"], out[f"{name}/kol_rc"] = y, info, np.int64(rc)
            y, info, rc = ref.kolmogorov2(X, lam, max_iters=40)
            out[f"{name}/kol_it40"], out[f"{name}/kol_it40_info"] = y, info
            for alg in (0, 1, 2):
                y, info, rc = ref.ccp2(X, lam, alg)
                out[f"{name}/ccp{alg}"], out[f"{name}/ccp{alg}_info"], out[f"{name}/ccp{alg}_rc"] = y, info, np.int64(rc)
                y, info, rc = ref.ccp2(X, lam, alg, max_iters=60)
                out[f"{name}/ccp{alg}_it60"], out[f"{name}/ccp{alg}_it60_info"] = y, info
    # a constant image is a fixed point of the first primal step: the loops exit through their `stop > 0` test
    C = np.asfortranarray(np.full((6, 9), 3.25))
    out["const/X"] = C
    for alg in (0, 1, 2):
        y, info, rc = ref.ccp2(C, 0.7, alg)
        out[f"const/ccp{alg}"], out[f"const/ccp{alg}_info"] = y, info
    y, info, rc = ref.kolmogorov2(C, 0.7)
    out["const/kol"], out["const/kol_info"] = y, info
    # invalid algorithm selector
    y, info, rc = ref.ccp2(C, 0.7, 5)
    out["const/ccp_bad_info"], out["const/ccp_bad_rc"] = info, np.int64(rc)
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "golden_2d_primal_dual.npz"), **out)
    print("golden_2d_primal_dual:", len(names), "cases (+ const)")


def gen_nd(ref):
    rng = np.random.default_rng(20260928)
    out = {}
    names = []
    for shape in [(4, 5, 6), (12, 10, 8), (7, 3, 2, 5), (64, 5, 3)]:
        name = "randn_" + "x".join(map(str, shape))
        names.append(name)
        X = np.asfortranarray(rng.standard_normal(shape))
        nd = len(shape)
        lams = [0.3, 0.2, 0.1, 0.25][:nd]
        dims = list(range(1, nd + 1))
        out[f"{name}/X"] = X
        out[f"{name}/lams"] = np.array(lams)
        y, info, rc, lam_after = ref.pd(X, lams, dims)
        out[f"{name}/pd"], out[f"{name}/pd_info"], out[f"{name}/pd_rc"] = y, info, np.int64(rc)
        out[f"{name}/pd_lams_after"] = lam_after       # PD_TV scales lambdas in caller memory (TVNDopt.cpp:100-101)
        y, info, rc, _ = ref.pd(X, lams, dims, max_iters=4)
        out[f"{name}/pd_it4"], out[f"{name}/pd_it4_info"] = y, info
        y, info, rc, _ = ref.pdr(X, lams, dims)
        out[f"{name}/pdr"], out[f"{name}/pdr_info"], out[f"{name}/pdr_rc"] = y, info, np.int64(rc)
        y, info, rc, _ = ref.pd2(X, [lams[0], lams[-1]], [1, nd])   # penalise first and last dims only
        out[f"{name}/pd2_first_last"], out[f"{name}/pd2_first_last_info"] = y, info
        y, info, rc, _ = ref.pd2(X, [lams[1], lams[1]], [2, 2])     # same dim twice
        out[f"{name}/pd2_22"] = y
        if nd == 3:
            y, info, rc = ref.yang3(X, 0.2)
            out[f"{name}/yang3"], out[f"{name}/yang3_info"], out[f"{name}/yang3_rc"] = y, info, np.int64(rc)
            y, info, rc = ref.yang3(X, 0.2, max_iters=5)
            out[f"{name}/yang3_it5"], out[f"{name}/yang3_it5_info"] = y, info
    # colour-image idiom: penalise dims 1,2 of a 3-D array (demo_filter_image_color.py:22)
    Xc = np.asfortranarray(rng.standard_normal((20, 24, 3)))
    out["color/X"] = Xc
    out["color/pd2_12"], out["color/pd2_12_info"] = ref.pd2(Xc, [0.15, 0.15], [1, 2])[:2]
    out["names"] = np.array(names)
    np.savez_compressed(os.path.join(GOLD, "golden_nd.npz"), **out)
    print("golden_nd:", len(names), "cases (+ color)")


def digest(a, step):
    """Compact fingerprint of a big output: strided subsample + moments."""
    flat = np.asarray(a).ravel(order="F")
    return {
        "sub": flat[::step].copy(),
        "sum": np.float64(flat.sum()),
        "abs": np.float64(np.abs(flat).sum()),
        "sq": np.float64((flat * flat).sum()),
        "max": np.float64(np.abs(flat).max()),
    }


def put(out, key, d):
    for k, v in d.items():
        out[f"{key}/{k}"] = v


def gen_large(ref, threads):
    """BASELINE.json configs at full size; inputs are re-created from seeds by the tests."""
    out = {"step": np.int64(4099)}
    t0 = time.time()
    # C1: tv1_1d Condat 1e6, lambda 0.5
    x = np.random.default_rng(0).standard_normal(1_000_000)
    put(out, "c1/x", digest(x, 4099))
    put(out, "c1/condat", digest(ref.tv1_condat(x, 0.5), 4099))
    put(out, "c1/hybrid", digest(ref.tv1_hybrid(x, 0.5), 4099))
    print("c1 done", time.time() - t0)
    # C2: DR 4096^2, lambda 0.1
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((4096, 4096)))
    put(out, "c2/X", digest(X, 4099))
    y, info, rc = ref.dr2(X, 0.1, n_threads=threads)
    put(out, "c2/dr2", digest(y, 4099)); out["c2/dr2_info"] = info
    print("c2 done", time.time() - t0)
    # C3: weighted DR 4096^2
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.standard_normal((4096, 4096)))
    W1 = np.asfortranarray(rng.uniform(0.05, 0.15, (4095, 4096)))
    W2 = np.asfortranarray(rng.uniform(0.05, 0.15, (4096, 4095)))
    put(out, "c3/W1", digest(W1, 4099)); put(out, "c3/W2", digest(W2, 4099))
    y, info, rc = ref.dr2w(X, W1, W2, n_threads=threads)
    put(out, "c3/dr2w", digest(y, 4099)); out["c3/dr2w_info"] = info
    del W1, W2
    print("c3 done", time.time() - t0)
    # C4: 512x512x64 volume (f32-cast then f64), lambda [.1,.1,.05]: what tvgen runs (PD_TV) + Yang3 scalar
    V = np.asfortranarray(np.random.default_rng(0).standard_normal((512, 512, 64)).astype(np.float32).astype(np.float64))
    put(out, "c4/V", digest(V, 4099))
    y, info, rc, _ = ref.pd(V, [0.1, 0.1, 0.05], [1, 2, 3], n_threads=threads)
    put(out, "c4/pd", digest(y, 4099)); out["c4/pd_info"] = info
    print("c4 pd done", time.time() - t0)
    y, info, rc = ref.yang3(V, 0.1)
    put(out, "c4/yang3", digest(y, 4099)); out["c4/yang3_info"] = info
    print("c4 yang3 done", time.time() - t0)
    # C5 items: DR 2048^2, seeds 0..2
    for k in range(3):
        X = np.asfortranarray(np.random.default_rng(k).standard_normal((2048, 2048)))
        y, info, rc = ref.dr2(X, 0.1, n_threads=threads)
        put(out, f"c5/{k}/dr2", digest(y, 4099))
    print("c5 done", time.time() - t0)
    # hard variant of C2 at 1024^2: 8x8 random blocks + 0.2 N(0,1), lambda 0.5 (BASELINE.md plan item 1)
    rng = np.random.default_rng(7)
    Xh = np.asfortranarray(np.kron(rng.standard_normal((8, 8)), np.ones((128, 128))) + 0.2 * rng.standard_normal((1024, 1024)))
    put(out, "hard/X", digest(Xh, 1031))
    y, info, rc = ref.dr2(Xh, 0.5, n_threads=threads)
    put(out, "hard/dr2", digest(y, 1031))
    print("hard done", time.time() - t0)
    np.savez_compressed(os.path.join(GOLD, "golden_large.npz"), **out)


def gen_large_extra(ref, threads):
    """Cases the bench line reports beside the headline (round 6), ADDED to the existing golden_large.npz (its other entries are kept
    byte for byte): `hard4096` = SURVEY 8(d)'s back-tracking-heavy image at the headline size (8 x 8 random blocks of 512 x 512 + 0.2
    N(0,1), lambda = 0.5) and `lam1` = the headline image at lambda = 1 (pieces of ~100 samples: the long-piece rungs)."""
    path = os.path.join(GOLD, "golden_large.npz")
    with np.load(path) as g:
        out = {k: g[k] for k in g.files}
    t0 = time.time()
    rng = np.random.default_rng(7)
    Xh = np.asfortranarray(np.kron(rng.standard_normal((8, 8)), np.ones((512, 512))) + 0.2 * rng.standard_normal((4096, 4096)))
    put(out, "hard4096/X", digest(Xh, 4099))
    y, info, rc = ref.dr2(Xh, 0.5, n_threads=threads)
    put(out, "hard4096/dr2", digest(y, 4099)); out["hard4096/dr2_info"] = info
    print("hard4096 done", time.time() - t0)
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((4096, 4096)))
    y, info, rc = ref.dr2(X, 1.0, n_threads=threads)
    put(out, "lam1/dr2", digest(y, 4099)); out["lam1/dr2_info"] = info
    print("lam1 done", time.time() - t0)
    np.savez_compressed(path, **out)


def gen_p2(ref):
    """TV-L2 (p = 2) fibres: what the compiled reference returns -- single-threaded (its fibres warm-start each other per
    OpenMP thread, so its output depends on the thread count) and to its own accuracy (duality gap 1e-5).  The exact
    solvers of this repo are checked against these within that accuracy, not to 1e-6."""
    rng = np.random.default_rng(20260927)
    out, names1, names2 = {}, [], []
    for n in (1, 2, 3, 10, 64, 100, 257, 1000, 4096):
        for lam in (0.05, 1.0, 7.0, 300.0):
            name = f"n{n}_lam{lam}"
            names1.append(name)
            x = rng.standard_normal(n) * (3.0 if n % 2 else 1.0)
            out[f"{name}/x"] = x
            out[f"{name}/lam"] = np.float64(lam)
            out[f"{name}/tv2"], out[f"{name}/tv2_info"] = ref.tv(x, lam, 2)
    for shape, lam in (((33, 47), 0.4), ((96, 64), 1.5), ((128, 200), 0.2)):
        name = f"img{shape[0]}x{shape[1]}"
        names2.append(name)
        X = rng.standard_normal(shape)
        out[f"{name}/X"] = X
        out[f"{name}/lam"] = np.float64(lam)
        for n1, n2 in ((2, 2), (1, 2), (2, 1)):
            out[f"{name}/dr2_{n1}{n2}"] = ref.dr2(X, lam, 0.7 * lam, n_threads=1, norm1=n1, norm2=n2)[0]
            y, info, rc, _ = ref.pd2(X, [lam, 0.7 * lam], [1, 2], norms=[n1, n2], n_threads=1)
            out[f"{name}/pd2_{n1}{n2}"], out[f"{name}/pd2_{n1}{n2}_info"] = y, info
    V = rng.standard_normal((24, 30, 12))
    out["vol/X"] = V
    y, info, rc, _ = ref.pd(V, [0.3, 0.2, 0.4], [1, 2, 3], norms=[2, 1, 2], n_threads=1)
    out["vol/pd_212"], out["vol/pd_212_info"] = y, info
    out["names1"] = np.array(names1)
    out["names2"] = np.array(names2)
    np.savez_compressed(os.path.join(GOLD, "golden_p2.npz"), **out)
    print("golden_p2.npz:", len(names1), "1-D cases,", len(names2), "images, 1 volume")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--large", action="store_true")
    ap.add_argument("--only-large", action="store_true")
    ap.add_argument("--only-large-extra", action="store_true", help="add the round-6 bench cases to golden_large.npz")
    ap.add_argument("--only-primal-dual", action="store_true", help="regenerate golden_2d_primal_dual.npz only")
    ap.add_argument("--only-p2", action="store_true", help="regenerate golden_p2.npz only")
    ap.add_argument("--threads", type=int, default=os.cpu_count() or 1)
    args = ap.parse_args()
    if not cpu.have_reference():
        cpu.build_reference(quiet=False)
    ref = cpu.reference()
    os.makedirs(GOLD, exist_ok=True)
    if args.only_p2:
        gen_p2(ref)
        return
    if args.only_large_extra:
        gen_large_extra(ref, args.threads)
        return
    if args.only_primal_dual:
        gen_1d_other_methods(ref)
        gen_2d_primal_dual(ref)
        return
    if not args.only_large:
        gen_1d(ref)
        gen_2d(ref)
        gen_1d_other_methods(ref)
        gen_2d_primal_dual(ref)
        gen_nd(ref)
        gen_p2(ref)
    if args.large or args.only_large:
        gen_large(ref, args.threads)
        gen_large_extra(ref, args.threads)


if __name__ == "__main__":
    main()
