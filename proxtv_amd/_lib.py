"""ctypes binding of libproxtv_amd.so (the C-ABI declared in include/proxtv_amd.h).

The library is the product: there is no Python or CPU fallback.  If the shared object is missing it is built
in-tree with hipcc (proxtv_amd/build.py); if that is impossible, or no gfx950 device is usable at call time,
the caller gets an exception -- never a silently different code path.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libproxtv_amd.so")
# A/B measurements (tools/ab*.sh) load an alternative build of the library.  The override is honoured only together with
# an explicit debug switch, so that a stray PROXTV_LIB in a production environment cannot redirect the load.
if os.environ.get("PROXTV_DEBUG_ALT_LIB") == "1" and os.environ.get("PROXTV_LIB"):
    LIB_PATH = os.environ["PROXTV_LIB"]

_dp = C.c_void_p      # double* (host or device, per entry point)
_ip = C.c_void_p      # int*

SIGNATURES = {
    # ---- part 1: drop-in entry points (reference: src/TVopt.h:88-141) ----
    "TV": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "linearizedTautString_TV1": (C.c_int, [_dp, C.c_double, _dp, C.c_int]),
    "classicTautString_TV1": (C.c_int, [_dp, C.c_int, C.c_double, _dp]),
    "classicTautString_TV1_offset": (C.c_int, [_dp, C.c_int, C.c_double, _dp, C.c_double]),
    "hybridTautString_TV1": (None, [_dp, C.c_int, C.c_double, _dp]),
    "hybridTautString_TV1_custom": (None, [_dp, C.c_int, C.c_double, _dp, C.c_double]),
    "tautString_TV1_Weighted": (C.c_int, [_dp, _dp, _dp, C.c_int]),
    "TV1D_denoise": (None, [_dp, _dp, C.c_int, C.c_double]),
    "TV1D_denoise_tautstring": (None, [_dp, _dp, C.c_int, C.c_double]),
    "dp": (None, [C.c_int, _dp, C.c_double, _dp]),
    "PN_TV1": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "PN_TV1_Weighted": (C.c_int, [_dp, _dp, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "SolveTVConvexQuadratic_a1_nw": (None, [C.c_int, _dp, C.c_double, _dp]),
    "SolveTVConvexQuadratic_a1": (None, [C.c_int, _dp, _dp, _dp]),
    "GP_TVp": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "OGP_TVp": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "FISTA_TVp": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "FW_TVp": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "GPFW_TVp": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_double, C.c_void_p]),
    "DR2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp,
                         C.c_int, C.c_int, _dp]),
    "DR2L1W_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, _dp, _dp, _dp, C.c_int, C.c_int, _dp]),
    "PD2_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_int]),
    "PD_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_int]),
    "PDR_TV": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_int]),
    "Yang2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "Yang3_TV": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "Kolmogorov2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp]),
    "CondatChambollePock2_TV": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_short, C.c_int, _dp]),
    "more_TV2": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int]),
    "morePG_TV2": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int, C.c_void_p]),
    "PG_TV2": (C.c_int, [_dp, C.c_double, _dp, _dp, C.c_int]),
    "newWorkspace": (C.c_void_p, [C.c_int]),
    "resetWorkspace": (None, [C.c_void_p]),
    "freeWorkspace": (None, [C.c_void_p]),
    "newWorkspaces": (C.c_void_p, [C.c_int, C.c_int]),
    "freeWorkspaces": (None, [C.c_void_p, C.c_int]),
    # ---- part 2: MI355X-native extensions ----
    "proxtv_init": (C.c_int, [C.c_int]),
    "proxtv_version": (C.c_char_p, []),
    "proxtv_last_error": (C.c_char_p, []),
    "proxtv_release_scratch": (None, []),
    "proxtv_set_option": (C.c_int, [C.c_char_p, C.c_int]),
    "proxtv_DR2_TV_dev": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, _dp, C.c_int, _dp, C.c_void_p]),
    "proxtv_DR2L1W_TV_dev": (C.c_int, [C.c_size_t, C.c_size_t, _dp, _dp, _dp, _dp, C.c_int, _dp, C.c_void_p]),
    "proxtv_PD2_TV_dev": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "proxtv_PD_TV_dev": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "proxtv_PDR_TV_dev": (C.c_int, [_dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "proxtv_Yang_TV_dev": (C.c_int, [_ip, C.c_int, _dp, _dp, _dp, C.c_int, _dp, C.c_void_p]),
    "proxtv_Kolmogorov2_TV_dev": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_int, _dp, C.c_void_p]),
    "proxtv_CondatChambollePock2_TV_dev": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, _dp, C.c_short, C.c_int, _dp,
                                                     C.c_void_p]),
    "proxtv_tv1_fibres_dev": (C.c_int, [_dp, _dp, _ip, C.c_int, C.c_int, C.c_double, _dp, C.c_void_p]),
    "proxtv_certify_fibres_dev": (C.c_long, [_dp, _dp, _ip, C.c_int, C.c_int, C.c_double, _dp, C.c_void_p]),
    "proxtv_DR2_TV_batch_dev": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, _dp,
                                          C.c_int, _dp, C.c_void_p]),
    "proxtv_DR2_TV_batch": (C.c_int, [C.c_size_t, C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, _dp, C.c_int, _dp]),
    "proxtv_tvp_fibres_dev": (C.c_int, [_dp, _dp, _ip, C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p]),
    "proxtv_DR2_TVp_dev": (C.c_int, [C.c_size_t, C.c_size_t, _dp, C.c_double, C.c_double, C.c_double, C.c_double, _dp, C.c_int,
                                     _dp, C.c_void_p]),
    "proxtv_PD_TVp_dev": (C.c_int, [C.c_int, _dp, _dp, _dp, _dp, _dp, _dp, _ip, C.c_int, C.c_int, C.c_int, C.c_void_p]),
    "proxtv_last_fixups": (C.c_long, []),
    "proxtv_chunk_mode": (C.c_int, []),
    "proxtv_debug_trace": (C.c_long, [C.c_void_p, C.c_long]),
    "proxtv_debug_why": (C.c_int, [C.c_void_p]),
    "proxtv_debug_counter": (C.c_long, [C.c_char_p]),
    "proxtv_calib_copy_dev": (C.c_int, [_dp, _dp, C.c_long, C.c_void_p]),
    "proxtv_last_kernel_ms": (C.c_double, [C.c_int]),
    "proxtv_last_kernel_launches": (C.c_long, [C.c_int]),
}

_lib = None


class ProxTVError(RuntimeError):
    """A HIP-path failure reported by libproxtv_amd (no device, HIP error, unsupported argument)."""


def _preload_hip_runtime():
    """One HIP runtime per process.

    libproxtv_amd.so needs `libamdhip64.so.7`.  The PyTorch-ROCm wheel bundles its own copy of that runtime (same
    SONAME, private directory); two HIP/HSA runtimes in one process cannot both open the GPU.  Whenever torch is
    installed, load ITS runtime first (without importing torch): the dynamic loader then binds our library to the
    already-loaded SONAME, and a later `import torch` finds the very same file.  Without torch the system runtime
    under /opt/rocm is picked up through the library's RUNPATH as usual.
    """
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return  # torch already brought its runtime in
    try:
        spec = importlib.util.find_spec("torch")
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def load(build_if_missing=True):
    """Return the ctypes handle with argtypes set; builds the library in-tree if it is not there."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise ImportError("libproxtv_amd.so is not built (run `python -m proxtv_amd.build`)")
        from . import build as _build
        _build.build()
    _preload_hip_runtime()
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)   # AttributeError here == the C-ABI lost a symbol
        except AttributeError:
            # (A/B runs against an OLDER build of the library -- the explicit debug switch above -- may lack this round's additions)
            if os.environ.get("PROXTV_DEBUG_ALT_LIB") == "1" and name in ("proxtv_certify_fibres_dev",):
                continue
            raise
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def last_error():
    msg = load().proxtv_last_error()
    return msg.decode() if msg else ""


def require_device():
    """Raise unless a gfx950 device is usable.  Called by the Python surface before any solve."""
    lib = load()
    if lib.proxtv_init(-1) != 0:
        raise ProxTVError("proxtv_amd: " + last_error())
    return lib


def check(what):
    """Raise if the last C call on this thread recorded an error (the C-ABI itself never throws)."""
    msg = last_error()
    if msg:
        raise ProxTVError(f"{what}: {msg}")
