"""Device-resident entry points: operate on torch tensors that already live in HBM.

torch is plumbing here (device memory + streams); every solve is the HIP path of libproxtv_amd.so reached through
the ``proxtv_*_dev`` C entry points of include/proxtv_amd.h.  Arrays follow the library's column-major convention:
an (M, N) image is passed as a tensor whose *transpose* is contiguous (``colmajor_empty`` / ``to_colmajor`` build
such tensors); N-D arrays likewise have dimension 0 fastest.
"""
import numpy as np
import torch

from . import _lib

_N_INFO = 3


def colmajor_empty(shape, device="cuda", dtype=torch.float64):
    """Uninitialised tensor of logical `shape` stored column-major (dimension 0 fastest)."""
    rev = tuple(reversed(tuple(shape)))
    return torch.empty(rev, device=device, dtype=dtype).permute(*reversed(range(len(rev))))


def to_colmajor(t):
    """Copy `t` (any layout) into column-major storage on its device."""
    out = colmajor_empty(t.shape, device=t.device, dtype=torch.float64)
    out.copy_(t)
    return out


def _is_colmajor(t):
    return t.permute(*reversed(range(t.dim()))).is_contiguous()


def _check(t, name, like=None, shape=None):
    if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float64 and _is_colmajor(t)):
        raise ValueError(f"{name}: expected a float64 CUDA tensor in column-major storage (see to_colmajor)")
    if like is not None and t.device != like.device:
        raise ValueError(f"{name}: lives on {t.device}, the input on {like.device}")
    if shape is not None and tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name}: shape {tuple(t.shape)}, expected {tuple(shape)}")


def _out(out, x, name="out"):
    """The output tensor: a fresh column-major one on x's device, or the caller's (checked: a tensor of another
    dtype / layout / shape / device would be filled with the wrong bytes or fault).  It may alias x: the library
    detects the overlap and solves through a scratch array."""
    if out is None:
        return colmajor_empty(x.shape, device=x.device)
    _check(out, name, like=x, shape=x.shape)
    return out


def _stream(dev):
    """hipStream_t handle for the C-ABI: torch's current stream ON THE TENSOR'S DEVICE (callers run under
    ``torch.cuda.device(x.device)``: the library keeps its state per device and uses the current one).  torch's legacy
    default stream has handle 0, which the library reads as "use my own per-thread (non-blocking) stream"; that stream
    does not order against the null stream, so drain torch's producers first.  Every *_dev entry point synchronises
    before returning, so consumers need no extra fence."""
    cur = torch.cuda.current_stream(dev)
    if cur.cuda_stream == 0:
        cur.synchronize()
    return cur.cuda_stream


def tv1_2d(x, w, max_iters=0, method="dr", out=None, w_row=None):
    """tv1_2d on an HBM-resident (M, N) image.  Returns (y, info)."""
    _check(x, "x")
    with torch.cuda.device(x.device):
        return _tv1_2d(x, w, max_iters, method, _out(out, x), w_row)


def _tv1_2d(x, w, max_iters, method, y, w_row):
    lib = _lib.require_device()
    info = np.zeros(_N_INFO)
    M, N = x.shape
    w_row = w if w_row is None else w_row
    _stream = lambda: globals()["_stream"](x.device)   # noqa: E731
    if method == "dr":
        lib.proxtv_DR2_TV_dev(M, N, x.data_ptr(), float(w), float(w_row), y.data_ptr(), int(max_iters),
                              info.ctypes.data, _stream())
    elif method == "pd":
        lam = np.array([w, w_row], dtype=np.float64)
        dims = np.array([1.0, 2.0])
        ns = np.array([M, N], dtype=np.int32)
        lib.proxtv_PD2_TV_dev(x.data_ptr(), lam.ctypes.data, dims.ctypes.data, y.data_ptr(), info.ctypes.data,
                              ns.ctypes.data, 2, 2, int(max_iters), _stream())
    elif method == "yang":
        lam = np.array([w, w_row], dtype=np.float64)
        ns = np.array([M, N], dtype=np.int32)
        lib.proxtv_Yang_TV_dev(ns.ctypes.data, 2, x.data_ptr(), lam.ctypes.data, y.data_ptr(), int(max_iters),
                               info.ctypes.data, _stream())
    elif method == "kolmogorov":
        lib.proxtv_Kolmogorov2_TV_dev(M, N, x.data_ptr(), float(w), y.data_ptr(), int(max_iters), info.ctypes.data, _stream())
    elif method in ("condat", "chambolle-pock", "chambolle-pock-acc"):
        alg = {"condat": 0, "chambolle-pock": 1, "chambolle-pock-acc": 2}[method]
        lib.proxtv_CondatChambollePock2_TV_dev(M, N, x.data_ptr(), float(w), y.data_ptr(), alg, int(max_iters),
                                               info.ctypes.data, _stream())
    else:
        raise NotImplementedError(method)
    _lib.check("device.tv1_2d")
    return y, info


def tv1w_2d(x, w_col, w_row, max_iters=0, out=None):
    """Weighted DR on HBM-resident arrays: w_col (M-1, N), w_row (M, N-1), all column-major."""
    _check(x, "x")
    M, N = x.shape
    _check(w_col, "w_col", like=x, shape=(M - 1, N))
    _check(w_row, "w_row", like=x, shape=(M, N - 1))
    y = _out(out, x)
    info = np.zeros(_N_INFO)
    with torch.cuda.device(x.device):
        lib = _lib.require_device()
        lib.proxtv_DR2L1W_TV_dev(M, N, x.data_ptr(), w_col.data_ptr(), w_row.data_ptr(), y.data_ptr(), int(max_iters),
                                 info.ctypes.data, _stream(x.device))
    _lib.check("device.tv1w_2d")
    return y, info


def tv1_2d_batch(xs, w, max_iters=0, out=None):
    """DR on a stack of B images held as a column-major (M, N, B) tensor (image b = xs[:, :, b])."""
    _check(xs, "xs")
    y = _out(out, xs)
    info = np.zeros(_N_INFO)
    M, N, B = xs.shape
    with torch.cuda.device(xs.device):
        lib = _lib.require_device()
        lib.proxtv_DR2_TV_batch_dev(M, N, B, xs.data_ptr(), float(w), float(w), y.data_ptr(), int(max_iters),
                                    info.ctypes.data, _stream(xs.device))
    _lib.check("device.tv1_2d_batch")
    return y, info


def tvgen(x, ws, ds, max_iters=0, method=None, out=None):
    """N-D TV-L1 on an HBM-resident column-major array.  method: None = reference dispatch (2 terms -> 'pd2',
    otherwise 'pd'); 'pdr' = parallel Douglas-Rachford; 'yang' = Yang ADMM (2-D / 3-D, one lambda per dim)."""
    _check(x, "x")
    with torch.cuda.device(x.device):
        return _tvgen(x, ws, ds, max_iters, method, _out(out, x))


def _tvgen(x, ws, ds, max_iters, method, y):
    lib = _lib.require_device()
    info = np.zeros(_N_INFO)
    ns = np.array(x.shape, dtype=np.int32)
    lam = np.array(ws, dtype=np.float64)
    dims = np.array(ds, dtype=np.float64)
    npen = lam.size
    _stream = lambda: globals()["_stream"](x.device)   # noqa: E731
    if method is None:
        method = "pd2" if npen == 2 else "pd"
    if method == "pd2":
        lib.proxtv_PD2_TV_dev(x.data_ptr(), lam.ctypes.data, dims.ctypes.data, y.data_ptr(), info.ctypes.data,
                              ns.ctypes.data, x.dim(), npen, int(max_iters), _stream())
    elif method in ("pd", "pdr"):
        scaled = lam * npen     # the host entry points scale in caller memory (src/TVNDopt.cpp:100-101); here explicit
        fn = lib.proxtv_PD_TV_dev if method == "pd" else lib.proxtv_PDR_TV_dev
        fn(x.data_ptr(), scaled.ctypes.data, dims.ctypes.data, y.data_ptr(), info.ctypes.data, ns.ctypes.data,
           x.dim(), npen, int(max_iters), _stream())
    elif method == "yang":
        lib.proxtv_Yang_TV_dev(ns.ctypes.data, x.dim(), x.data_ptr(), lam.ctypes.data, y.data_ptr(), int(max_iters),
                               info.ctypes.data, _stream())
    else:
        raise NotImplementedError(method)
    _lib.check("device.tvgen")
    return y, info


def tv1_fibres(x, w, dim, weights=None, out=None):
    """Batched exact 1-D TV-L1 prox of every fibre of `x` along 0-based `dim` (the per-sweep kernel)."""
    _check(x, "x")
    y = _out(out, x)
    ns = np.array(x.shape, dtype=np.int32)
    if not 0 <= int(dim) < x.dim():
        raise ValueError(f"dim {dim} out of range for a {x.dim()}-D array")
    wp = 0
    if weights is not None:
        wshape = list(x.shape)
        wshape[int(dim)] -= 1              # one penalty per edge along the fibre
        _check(weights, "weights", like=x, shape=wshape)
        wp = weights.data_ptr()
    with torch.cuda.device(x.device):
        lib = _lib.require_device()
        lib.proxtv_tv1_fibres_dev(x.data_ptr(), y.data_ptr(), ns.ctypes.data, x.dim(), int(dim), float(w), wp,
                                  _stream(x.device))
    _lib.check("device.tv1_fibres")
    return y


def certify_fibres(x, y, w, dim, weights=None):
    """The certificate of ``y = tv1_fibres(x, w, dim)``: the number of fibres along `dim` for which `y` is NOT the exact TV-L1 prox of
    `x` (optimality conditions of the 1-D problem, fibre by fibre, within rounding).  0: `y` is the prox.  -1: nothing to check
    (w <= 0 without weights, or y is x)."""
    _check(x, "x")
    _check(y, "y", like=x, shape=x.shape)
    ns = np.array(x.shape, dtype=np.int32)
    if not 0 <= int(dim) < x.dim():
        raise ValueError(f"dim {dim} out of range for a {x.dim()}-D array")
    wp = 0
    if weights is not None:
        wshape = list(x.shape)
        wshape[int(dim)] -= 1
        _check(weights, "weights", like=x, shape=wshape)
        wp = weights.data_ptr()
    with torch.cuda.device(x.device):
        lib = _lib.require_device()
        failed = lib.proxtv_certify_fibres_dev(x.data_ptr(), y.data_ptr(), ns.ctypes.data, x.dim(), int(dim), float(w), wp,
                                               _stream(x.device))
    if failed == -2:
        _lib.check("device.certify_fibres")
    return int(failed)
