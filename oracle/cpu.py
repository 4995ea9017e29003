"""ctypes drivers for the CPU checkers -- TEST INFRASTRUCTURE, not product code.

Two interchangeable handles with the same method names:

* ``oracle()``    -> oracle/libtvoracle.so, this repo's plain-C restatement (symbols ``orc_<name>``);
* ``reference()`` -> oracle/_ref/libproxtv_ref.so, the unmodified reference sources compiled by
  ``make -C oracle ref`` (symbols ``<name>``; argument types per the reference's ``src/TVopt.h:88-141``).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libtvoracle.so")
REF_SO = os.path.join(HERE, "_ref", "libproxtv_ref.so")

_dp = C.c_void_p
_SIGS = {
    # name: (restype, argtypes)
    "linearizedTautString_TV1": (C.c_int, [_dp, C.c_double, _dp, C.c_int]),
    "hybridTautString_TV1": (None, [_dp, C.c_int, C.c_double, _dp]),
    "hybridTautString_TV1_custom": (None, [_dp, C.c_int, C.c_double, _dp, C.c_double]),
    "classicTautString_TV1": (C.c_int, [_dp, C.c_int, C.c_double, _dp]),
    "classicTautString_TV1_offset": (C.c_int, [_dp, C.c_int, C.c_double, _dp, C.c_double]),
    "tautString_TV1_Weighted": (C.c_int, [_dp, _dp, _dp, C.c_int]),
    "TV1D_denoise": (None, [_dp, _dp, C.c_int, C.c_double]),
    "DR2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp,
                         C.c_int, C.c_int, _dp]),
    "DR2L1W_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]),
    "PD2_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "PD_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "PDR_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _dp, C.c_int, C.c_int, C.c_int, C.c_int]),
    "Yang2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "Yang3_TV": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "Kolmogorov2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "CondatChambollePock2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_short, C.c_int, _dp]),
}


def _f64(a):
    return np.ascontiguousarray(a, dtype=np.float64) if not (
        isinstance(a, np.ndarray) and a.dtype == np.float64 and (a.flags.f_contiguous or a.flags.c_contiguous)
    ) else a


class CpuLib:
    """Thin numpy front-end over one of the two CPU libraries (column-major arrays, like the reference)."""

    def __init__(self, path, prefix, kind):
        self.path, self.kind = path, kind
        self._lib = C.CDLL(path)
        self._fn = {}
        for name, (res, args) in _SIGS.items():
            f = getattr(self._lib, prefix + name)
            f.restype, f.argtypes = res, args
            self._fn[name] = f
        if not prefix:  # the reference's other tv1_1d methods (no restatement: used to generate / check golden vectors only)
            for name, (res, args) in {
                "PN_TV1": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, _dp]),
                "SolveTVConvexQuadratic_a1_nw": (None, [C.c_int, _dp, C.c_double, _dp]),
                "TV1D_denoise_tautstring": (None, [_dp, _dp, C.c_int, C.c_double]),
                "dp": (None, [C.c_int, _dp, C.c_double, _dp]),
            }.items():
                f = getattr(self._lib, name)
                f.restype, f.argtypes = res, args
                self._fn[name] = f
        if prefix:  # restated here as well: Johnson's dynamic programme
            f = self._lib.orc_dp
            f.restype, f.argtypes = None, [C.c_int, _dp, C.c_double, _dp]
            self._fn["dp"] = f
        if prefix:  # oracle-only extension
            f = self._lib.orc_Yang3_TV_perdim
            f.restype = C.c_int
            f.argtypes = [C.c_size_t, C.c_size_t, C.c_size_t, _dp, _dp, _dp, C.c_int, _dp]
            self._fn["Yang3_TV_perdim"] = f

    # ---- 1-D -------------------------------------------------------------------------------------------------
    def _run1d(self, call, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out = np.zeros(x.size)
        call(x, out)
        return out

    def tv1_linearized(self, x, lam):
        return self._run1d(lambda a, o: self._fn["linearizedTautString_TV1"](a.ctypes.data, lam, o.ctypes.data, a.size), x)

    def tv1_hybrid(self, x, lam, bexp=None):
        if bexp is None:
            return self._run1d(lambda a, o: self._fn["hybridTautString_TV1"](a.ctypes.data, a.size, lam, o.ctypes.data), x)
        return self._run1d(lambda a, o: self._fn["hybridTautString_TV1_custom"](a.ctypes.data, a.size, lam, o.ctypes.data, bexp), x)

    def tv1_classic(self, x, lam, offset=None):
        if offset is None:
            return self._run1d(lambda a, o: self._fn["classicTautString_TV1"](a.ctypes.data, a.size, lam, o.ctypes.data), x)
        return self._run1d(lambda a, o: self._fn["classicTautString_TV1_offset"](a.ctypes.data, a.size, lam, o.ctypes.data, offset), x)

    def tv1_condat(self, x, lam):
        return self._run1d(lambda a, o: self._fn["TV1D_denoise"](a.ctypes.data, o.ctypes.data, a.size, lam), x)

    def tv1_dp(self, x, lam):
        """Johnson's dynamic programme (both libraries)."""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(x.size)
        self._fn["dp"](int(x.size), x.ctypes.data, lam, out.ctypes.data)
        return out

    def tv1_other_method(self, x, lam, method, sigma=0.05):
        """reference() only: 'pn' | 'kolmogorov' | 'condattautstring' | 'dp' (prox_tv/__init__.py:197-216)"""
        x = np.ascontiguousarray(x, dtype=np.float64)
        out = np.zeros(x.size)
        n = int(x.size)
        if method == "pn":
            info = np.zeros(3)
            self._fn["PN_TV1"](x.ctypes.data, lam, out.ctypes.data, info.ctypes.data, n, sigma, None)
        elif method == "kolmogorov":
            self._fn["SolveTVConvexQuadratic_a1_nw"](n, x.ctypes.data, lam, out.ctypes.data)
        elif method == "condattautstring":
            self._fn["TV1D_denoise_tautstring"](x.ctypes.data, out.ctypes.data, n, lam)
        elif method == "dp":
            self._fn["dp"](n, x.ctypes.data, lam, out.ctypes.data)
        else:
            raise ValueError(method)
        return out

    def tv(self, x, lam, p):
        """TV() dispatcher: p = 1 (hybrid taut string) or p = 2.  For p = 2 the two libraries differ BY DESIGN: the
        oracle solves the TV-L2 prox to convergence (orc_TV2_exact), the reference's morePG_TV2 stops at a duality gap
        of 1e-5 (the reference has 7 arguments: a NULL workspace -- no warm start -- is passed)."""
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        out, info = np.zeros(x.size), np.zeros(3)
        f = self._lib.orc_TV if self.kind == "port" else self._lib.TV
        f.restype = C.c_int
        if self.kind == "port":
            f.argtypes = [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double]
            f(x.ctypes.data, lam, out.ctypes.data, info.ctypes.data, x.size, float(p))
        else:
            f.argtypes = [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, _dp]
            f(x.ctypes.data, lam, out.ctypes.data, info.ctypes.data, x.size, float(p), None)
        return out, info

    def tv1_weighted(self, x, w):
        w = np.ascontiguousarray(w, dtype=np.float64).ravel()
        return self._run1d(lambda a, o: self._fn["tautString_TV1_Weighted"](a.ctypes.data, w.ctypes.data, o.ctypes.data, a.size), x)

    # ---- 2-D / N-D (inputs any layout; converted to column-major; outputs column-major) -----------------------
    def dr2(self, X, w1, w2=None, max_iters=0, n_threads=1, norm1=1.0, norm2=1.0):
        X = np.asfortranarray(X, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        w2 = w1 if w2 is None else w2
        rc = self._fn["DR2_TV"](X.shape[0], X.shape[1], X.ctypes.data, w1, w2, float(norm1), float(norm2), out.ctypes.data,
                                n_threads, max_iters, info.ctypes.data)
        return out, info, rc

    def dr2w(self, X, W1, W2, max_iters=0, n_threads=1):
        X = np.asfortranarray(X, dtype=np.float64)
        W1 = np.asfortranarray(W1, dtype=np.float64)
        W2 = np.asfortranarray(W2, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        rc = self._fn["DR2L1W_TV"](X.shape[0], X.shape[1], X.ctypes.data, W1.ctypes.data, W2.ctypes.data,
                                   out.ctypes.data, n_threads, max_iters, info.ctypes.data)
        return out, info, rc

    def _pd_like(self, name, X, lambdas, dims, norms=None, max_iters=0, n_threads=1):
        X = np.asfortranarray(X, dtype=np.float64)
        lam = np.array(lambdas, dtype=np.float64)  # private copy: PD_TV / PDR_TV scale it in place
        npen = lam.size
        nrm = np.ones(npen) if norms is None else np.array(norms, dtype=np.float64)
        dm = np.array(dims, dtype=np.float64)
        ns = np.array(X.shape, dtype=np.int32)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        rc = self._fn[name](X.ctypes.data, lam.ctypes.data, nrm.ctypes.data, dm.ctypes.data, out.ctypes.data,
                            info.ctypes.data, ns.ctypes.data, X.ndim, npen, n_threads, max_iters)
        return out, info, rc, lam

    def pd2(self, X, lambdas, dims, **kw):
        return self._pd_like("PD2_TV", X, lambdas, dims, **kw)

    def pd(self, X, lambdas, dims, **kw):
        return self._pd_like("PD_TV", X, lambdas, dims, **kw)

    def pdr(self, X, lambdas, dims, **kw):
        return self._pd_like("PDR_TV", X, lambdas, dims, **kw)

    def yang2(self, X, lam, max_iters=0):
        X = np.asfortranarray(X, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        rc = self._fn["Yang2_TV"](X.shape[0], X.shape[1], X.ctypes.data, lam, out.ctypes.data, max_iters, info.ctypes.data)
        return out, info, rc

    def kolmogorov2(self, X, lam, max_iters=0):
        X = np.asfortranarray(X, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        rc = self._fn["Kolmogorov2_TV"](X.shape[0], X.shape[1], X.ctypes.data, lam, out.ctypes.data, max_iters,
                                        info.ctypes.data)
        return out, info, rc

    def ccp2(self, X, lam, alg, max_iters=0):
        """alg 0 = Condat, 1 = Chambolle-Pock, 2 = accelerated Chambolle-Pock"""
        X = np.asfortranarray(X, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        rc = self._fn["CondatChambollePock2_TV"](X.shape[0], X.shape[1], X.ctypes.data, lam, out.ctypes.data, alg,
                                                 max_iters, info.ctypes.data)
        return out, info, rc

    def yang3(self, X, lam, max_iters=0):
        X = np.asfortranarray(X, dtype=np.float64)
        out = np.zeros(X.shape, order="F")
        info = np.zeros(3)
        if np.ndim(lam) == 0:
            rc = self._fn["Yang3_TV"](X.shape[0], X.shape[1], X.shape[2], X.ctypes.data, float(lam), out.ctypes.data,
                                      max_iters, info.ctypes.data)
        else:
            l3 = np.array(lam, dtype=np.float64)
            rc = self._fn["Yang3_TV_perdim"](X.shape[0], X.shape[1], X.shape[2], X.ctypes.data, l3.ctypes.data,
                                             out.ctypes.data, max_iters, info.ctypes.data)
        return out, info, rc


def build_oracle(quiet=True):
    subprocess.run(["make", "-C", HERE] + (["-s"] if quiet else []), check=True)


def build_reference(quiet=True):
    """Only possible where /root/reference exists (the build container)."""
    subprocess.run(["make", "-C", HERE, "ref"] + (["-s"] if quiet else []), check=True)


_cache = {}


def oracle():
    if "o" not in _cache:
        if not os.path.exists(ORACLE_SO):
            build_oracle()
        _cache["o"] = CpuLib(ORACLE_SO, "orc_", "port")
    return _cache["o"]


def have_reference():
    return os.path.exists(REF_SO)


def reference():
    if "r" not in _cache:
        _cache["r"] = CpuLib(REF_SO, "", "reference")
    return _cache["r"]
