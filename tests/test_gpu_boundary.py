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
    scratch array when it sees the overlap).  Compared to 1e-12: two calls may run different kernel geometries (the
    adaptive policy explores across calls) -- the chunk kernels differ among themselves in the last ulps, the pinning
    solver works on running sums and agrees with them to ~1e-14 of the data's scale."""
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
        np.testing.assert_allclose(z.cpu().numpy(), ref[method], rtol=0, atol=1e-12)
        assert info[0] == info2[0]
    assert_close(ref["dr"], oracle.dr2(X, 0.25)[0], tol=1e-11, what="dr")
    # weighted DR, N-D loops, single sweeps in both directions
    W1, W2 = rng.uniform(0.05, 0.3, (259, 340)), rng.uniform(0.05, 0.3, (260, 339))
    xd = device.to_colmajor(torch.from_numpy(X).cuda())
    w1, w2 = device.to_colmajor(torch.from_numpy(W1).cuda()), device.to_colmajor(torch.from_numpy(W2).cuda())
    a = device.tv1w_2d(xd, w1, w2)[0].cpu().numpy().copy()
    b = device.tv1w_2d(xd, w1, w2, out=xd)[0].cpu().numpy()
    np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    V = rng.standard_normal((30, 40, 20))
    for method in (None, "pdr", "yang"):
        vd = device.to_colmajor(torch.from_numpy(V).cuda())
        a = device.tvgen(vd, [0.2, 0.1, 0.3], [1, 2, 3], method=method)[0].cpu().numpy().copy()
        b = device.tvgen(vd, [0.2, 0.1, 0.3], [1, 2, 3], method=method, out=vd)[0].cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)
    big = rng.standard_normal((1500, 700))
    for dim in (0, 1):
        bd = device.to_colmajor(torch.from_numpy(big).cuda())
        a = device.tv1_fibres(bd, 0.4, dim).cpu().numpy().copy()
        b = device.tv1_fibres(bd, 0.4, dim, out=bd).cpu().numpy()
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-12)


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
    # (not bit for bit: the adaptive policy may take its one look at the pinning rung in either call)
    np.testing.assert_allclose(ptv.tv1_1d(x, 0.3), before, rtol=0, atol=1e-12)
    assert_close(before, oracle.tv1_hybrid(x, 0.3), tol=1e-11)
