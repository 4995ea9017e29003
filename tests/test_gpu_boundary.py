"""The drop-in boundary under conditions the reference's callers can create: several host threads
solving at once, outputs that alias inputs, device tensors of the wrong kind, a device that does not exist."""
import threading

import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def test_concurrent_host_threads_mixed_solvers(ptv, oracle):
    """Four host threads, each with its own stream / scratch pool / geometry policy inside the library, run different
    solvers on different inputs at the same time; every result is checked against the oracle.  (Options are
    process-wide and are not touched here: see include/proxtv_amd.h.)"""
    rng = np.random.default_rng(77)
    X = [rng.standard_normal((300 + 40 * k, 500 - 30 * k)) for k in range(4)]
    W = [(rng.uniform(0.05, 0.2, (x.shape[0] - 1, x.shape[1])), rng.uniform(0.05, 0.2, (x.shape[0], x.shape[1] - 1))) for x in X]
    V = [rng.standard_normal((40 + k, 50, 24)) for k in range(4)]
    s1 = [rng.standard_normal(20000 + 777 * k) for k in range(4)]
    want = []
    for k in range(4):
        want.append({
            "dr": oracle.dr2(X[k], 0.3)[0],
            "drw": oracle.dr2w(X[k], *W[k])[0],
            "pd": oracle.pd(V[k], [0.2, 0.1, 0.3], [1, 2, 3])[0],
            "1d": oracle.tv1_hybrid(s1[k], 0.7),
            "pd2": oracle.pd2(X[k], [0.2, 0.2], [1, 2])[0] if hasattr(oracle, "pd2") else None,
        })
    got = [dict() for _ in range(4)]
    errors = []

    def work(k):
        try:
            for rep in range(3):     # interleave solver kinds differently per thread
                order = ["dr", "drw", "pd", "1d"]
                order = order[k:] + order[:k]
                for what in order:
                    if what == "dr":
                        got[k]["dr"] = ptv.tv1_2d(X[k], 0.3)
                    elif what == "drw":
                        got[k]["drw"] = ptv.tv1w_2d(X[k], *W[k])
                    elif what == "pd":
                        got[k]["pd"] = ptv.tvgen(V[k], [0.2, 0.1, 0.3], [1, 2, 3], [1, 1, 1])
                    else:
                        got[k]["1d"] = ptv.tv1_1d(s1[k], 0.7)
        except Exception as e:   # noqa: BLE001
            errors.append((k, repr(e)))

    threads = [threading.Thread(target=work, args=(k,)) for k in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for k in range(4):
        for what in ("dr", "drw", "pd", "1d"):
            assert_close(got[k][what], want[k][what], tol=1e-9, what=f"thread {k}: {what}")


def test_device_outputs_may_alias_inputs(oracle):
    """out = x on every device entry point: the same result as the out-of-place call (the library solves through a
    scratch array when it sees the overlap) -- bit for bit: which kernels a sweep runs is a function of its input
    (option "deterministic", the default), so two calls on the same data take the same path."""
    import torch
    from proxtv_amd import device
    rng = np.random.default_rng(78)
    X = rng.standard_normal((260, 340))
    ref = {}
    for method in ("dr", "pd", "yang", "kolmogorov", "condat"):
        kw = {"max_iters": 30} if method in ("kolmogorov", "condat") else {}
        xd = device.to_colmajor(torch.from_numpy(X).cuda())
        y, info = device.tv1_2d(xd, 0.25, method=method, **kw)
        ref[method] = y.cpu().numpy().copy()
        z, info2 = device.tv1_2d(xd, 0.25, method=method, out=xd, **kw)
        assert z.data_ptr() == xd.data_ptr()
        np.testing.assert_array_equal(z.cpu().numpy(), ref[method])
        assert info[0] == info2[0]
    assert_close(ref["dr"], oracle.dr2(X, 0.25)[0], tol=1e-11, what="dr")
    # weighted DR, N-D loops, single sweeps in both directions
    W1, W2 = rng.uniform(0.05, 0.3, (259, 340)), rng.uniform(0.05, 0.3, (260, 339))
    xd = device.to_colmajor(torch.from_numpy(X).cuda())
    w1, w2 = device.to_colmajor(torch.from_numpy(W1).cuda()), device.to_colmajor(torch.from_numpy(W2).cuda())
    a = device.tv1w_2d(xd, w1, w2)[0].cpu().numpy().copy()
    b = device.tv1w_2d(xd, w1, w2, out=xd)[0].cpu().numpy()
    np.testing.assert_array_equal(a, b)
    V = rng.standard_normal((30, 40, 20))
    for method in (None, "pdr", "yang"):
        vd = device.to_colmajor(torch.from_numpy(V).cuda())
        a = device.tvgen(vd, [0.2, 0.1, 0.3], [1, 2, 3], method=method)[0].cpu().numpy().copy()
        b = device.tvgen(vd, [0.2, 0.1, 0.3], [1, 2, 3], method=method, out=vd)[0].cpu().numpy()
        np.testing.assert_array_equal(a, b)
    big = rng.standard_normal((1500, 700))
    for dim in (0, 1):
        bd = device.to_colmajor(torch.from_numpy(big).cuda())
        a = device.tv1_fibres(bd, 0.4, dim).cpu().numpy().copy()
        b = device.tv1_fibres(bd, 0.4, dim, out=bd).cpu().numpy()
        np.testing.assert_array_equal(a, b)


def test_device_api_rejects_wrong_tensors():
    import torch
    from proxtv_amd import device
    x = device.to_colmajor(torch.randn(64, 48, dtype=torch.float64).cuda())
    for bad in (torch.empty(64, 48, dtype=torch.float64, device="cuda"),                # row-major
                device.colmajor_empty((64, 48), dtype=torch.float32),                   # wrong dtype
                device.colmajor_empty((64, 47)),                                        # wrong shape
                torch.empty(48, 64, dtype=torch.float64).T):                            # right layout, but on the host
        with pytest.raises(ValueError):
            device.tv1_2d(x, 0.1, out=bad)
        with pytest.raises(ValueError):
            device.tv1_fibres(x, 0.1, 0, out=bad)
        with pytest.raises(ValueError):
            device.tvgen(x, [0.1, 0.1], [1, 2], out=bad)
    w_ok_col, w_ok_row = device.colmajor_empty((63, 48)).fill_(0.1), device.colmajor_empty((64, 47)).fill_(0.1)
    device.tv1w_2d(x, w_ok_col, w_ok_row)
    with pytest.raises(ValueError):
        device.tv1w_2d(x, w_ok_row, w_ok_col)                 # swapped: would read past the buffers
    with pytest.raises(ValueError):
        device.tv1_fibres(x, 0.0, 1, weights=w_ok_col)        # weights of the other dimension
    device.tv1_fibres(x, 0.0, 1, weights=w_ok_row)
    with pytest.raises(ValueError):
        device.tv1_fibres(x, 0.1, 2)


def test_unknown_device_fails_cleanly_and_state_survives(clib, ptv, oracle):
    """proxtv_init with a device that does not exist reports failure; the thread's state on the real device is intact."""
    x = np.random.default_rng(79).standard_normal(5000)
    before = ptv.tv1_1d(x, 0.3)
    assert clib.proxtv_init(97) != 0
    assert clib.proxtv_init(0) == 0
    np.testing.assert_array_equal(ptv.tv1_1d(x, 0.3), before)
    assert_close(before, oracle.tv1_hybrid(x, 0.3), tol=1e-11)


def test_results_are_reproducible_bit_for_bit(ptv, clib, oracle):
    """The reference is bit-identical from run to run whatever its thread count; so is this library by
    default: the geometry a sweep runs is a function of (input statistics, lambda), not of what the thread solved before or
    of measured times.  Same calls, other workloads in between, three passes -- every result equals the first pass's."""
    rng = np.random.default_rng(80)
    X = rng.standard_normal((700, 900))
    B = np.kron(rng.standard_normal((7, 9)), np.ones((100, 100))) + 0.1 * rng.standard_normal((700, 900))
    x1 = rng.standard_normal(200000)
    V = rng.standard_normal((120, 100, 40))
    W1, W2 = rng.uniform(0.05, 1.0, (699, 900)), rng.uniform(0.05, 1.0, (700, 899))

    def calls():
        yield "dr 0.1", lambda: ptv.tv1_2d(X, 0.1)
        yield "1d 0.5", lambda: ptv.tv1_1d(x1, 0.5)
        yield "dr 0.6", lambda: ptv.tv1_2d(X, 0.6)
        yield "blocks", lambda: ptv.tv1_2d(B, 0.5)
        yield "dr 2.0", lambda: ptv.tv1_2d(X, 2.0)
        yield "1d 30", lambda: ptv.tv1_1d(x1, 30.0)
        yield "pd", lambda: ptv.tvgen(V, [0.3, 0.2, 0.4], [1, 2, 3], [1, 1, 1])
        yield "drw", lambda: ptv.tv1w_2d(X, W1, W2)
        yield "yang", lambda: ptv.tv1_2d(X, 0.3, method="yang")
    assert clib.proxtv_set_option(b"deterministic", 1) == 1      # the default
    first = {}
    for rep in range(3):
        order = list(calls())
        if rep:
            order = [order[k] for k in rng.permutation(len(order))]
        for name, fn in order:
            got = fn()
            if rep == 0:
                first[name] = got.copy()
            else:
                np.testing.assert_array_equal(got, first[name], err_msg=name)
    assert_close(first["dr 0.6"], oracle.dr2(X, 0.6)[0], tol=1e-10)
    assert_close(first["blocks"], oracle.dr2(B, 0.5)[0], tol=1e-10)
    assert_close(first["1d 30"], oracle.tv1_hybrid(x1, 30.0), tol=1e-10)


def test_adaptive_policy_is_still_exact(ptv, clib, oracle):
    """deterministic = 0: the hill climb on measured sweep times (seeded by the same statistics).  Results stay exact; two
    calls may differ in the last bits (different rungs round differently)."""
    before = clib.proxtv_set_option(b"deterministic", 0)
    try:
        rng = np.random.default_rng(81)
        X = rng.standard_normal((700, 900))
        for lam in (0.1, 0.6, 2.0, 0.1):
            want = oracle.dr2(X, lam)[0]
            for _ in range(3):
                assert_close(ptv.tv1_2d(X, lam), want, tol=1e-10, what=f"adaptive lam={lam}")
        x1 = rng.standard_normal(300000)
        for lam in (0.5, 30.0, 0.5):
            want = oracle.tv1_hybrid(x1, lam)
            for _ in range(4):
                assert_close(ptv.tv1_1d(x1, lam), want, tol=1e-10, what=f"adaptive 1-D lam={lam}")
    finally:
        clib.proxtv_set_option(b"deterministic", before)


def test_seed_sends_uneven_data_to_the_pinning_rung(ptv, clib, oracle):
    """The policy's seed looks at more than an average: an image whose right half is flat (or that is flat but for sparse
    spikes) has stretches no speculative walk can be proven on -- rung 3, whatever the lively half says; white noise of the
    same size takes the chunk kernels.  Results exact either way."""
    rng = np.random.default_rng(82)
    Z = rng.standard_normal((640, 900))
    half = Z.copy(); half[:, 450:] = 0.0
    spikes = np.zeros((640, 900)); m = rng.random((640, 900)) < 0.05; spikes[m] = 10.0 * rng.standard_normal(int(m.sum()))
    before = (clib.proxtv_set_option(b"chunk_mode", -1), clib.proxtv_set_option(b"deterministic", 1))   # (the suite may run pinned)
    try:
        assert_close(ptv.tv1_2d(Z, 0.1), oracle.dr2(Z, 0.1)[0], tol=1e-10)
        assert clib.proxtv_chunk_mode() == 0
        for name, X in (("half flat", half), ("spikes", spikes)):
            assert_close(ptv.tv1_2d(X, 0.1), oracle.dr2(X, 0.1)[0], tol=1e-10, what=name)
            assert clib.proxtv_chunk_mode() == 3, (name, clib.proxtv_chunk_mode())
    finally:
        clib.proxtv_set_option(b"chunk_mode", before[0])
        clib.proxtv_set_option(b"deterministic", before[1])


def test_why_counters(ptv, clib):
    """Tuning aid: what left work to the repair kernel.  Nothing on white noise at small lambda; something at lambda = 0.6."""
    import ctypes as C
    X = np.random.default_rng(83).standard_normal((1500, 1500))
    why = (C.c_uint * 8)()
    before = (clib.proxtv_set_option(b"chunk_mode", -1), clib.proxtv_set_option(b"deterministic", 1), clib.proxtv_set_option(b"xlink", 1))
    clib.proxtv_set_option(b"why", 1)
    try:
        clib.proxtv_debug_why(why)
        ptv.tv1_2d(X, 0.1)
        assert clib.proxtv_debug_why(why) == 8 and sum(why[:4]) == 0, list(why)
        ptv.tv1_2d(X, 0.6)
        assert clib.proxtv_debug_why(why) == 8 and sum(why[:4]) > 0, list(why)
    finally:
        clib.proxtv_set_option(b"why", 0)
        clib.proxtv_set_option(b"chunk_mode", before[0])
        clib.proxtv_set_option(b"deterministic", before[1])
        clib.proxtv_set_option(b"xlink", before[2])
