"""No-GPU checks of the drop-in boundary: the shared library loads, exports every symbol include/proxtv_amd.h declares,
fails loudly without a device (no CPU fallback), and the Python surface mirrors the reference's argument checks."""
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _has_gpu():
    return os.path.exists("/dev/kfd")


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "proxtv_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"^\s*(?:const\s+)?(?:int|void|double|long|char|Workspace)\s*\**\s*(\w+)\s*\(", text, flags=re.M)
    return sorted(set(names))


def test_library_loads_and_exports_every_declared_symbol():
    from proxtv_amd import _lib
    lib = _lib.load()
    names = declared_symbols()
    # the hot-path entry points the reference's cffi cdef / mex gateways bind (prox_tv/prox_tv_build.py:13-76)
    for must in ["TV", "hybridTautString_TV1", "hybridTautString_TV1_custom", "linearizedTautString_TV1",
                 "classicTautString_TV1", "classicTautString_TV1_offset", "tautString_TV1_Weighted", "TV1D_denoise",
                 "DR2_TV", "DR2L1W_TV", "PD2_TV", "PD_TV", "PDR_TV", "Yang2_TV", "Yang3_TV", "newWorkspace",
                 "freeWorkspace", "proxtv_DR2_TV_batch_dev", "proxtv_tv1_fibres_dev", "proxtv_certify_fibres_dev"]:
        assert must in names, must
    for n in names:
        assert hasattr(lib, n), f"libproxtv_amd.so does not export {n}"
    assert set(_lib.SIGNATURES) == set(names), set(_lib.SIGNATURES) ^ set(names)
    assert lib.proxtv_version().startswith(b"proxtv_amd")


# Every function the reference's cffi cdef declares (prox_tv/prox_tv_build.py:13-76 -- names only: cffi's API mode resolves
# each of them when the extension module is linked, so the cdef links against libproxtv_amd.so unmodified iff all resolve).
REFERENCE_CDEF_FUNCTIONS = [
    "TV1D_denoise", "TV1D_denoise_tautstring", "dp",
    "PN_TV1", "linearizedTautString_TV1", "classicTautString_TV1", "hybridTautString_TV1", "hybridTautString_TV1_custom",
    "SolveTVConvexQuadratic_a1_nw",
    "PN_TV1_Weighted", "tautString_TV1_Weighted", "SolveTVConvexQuadratic_a1",
    "more_TV2", "PG_TV2", "morePG_TV2",
    "DR2L1W_TV",
    "PD2_TV", "DR2_TV", "CondatChambollePock2_TV", "Yang2_TV", "Kolmogorov2_TV",
    "PD_TV",
    "GP_TVp", "OGP_TVp", "FISTA_TVp", "FW_TVp", "GPFW_TVp",
]


def test_every_function_of_the_reference_cdef_resolves():
    """`nm -D` on the shipped library: all 27 cdef'd functions are defined, unmangled, in its dynamic symbol table."""
    import subprocess
    from proxtv_amd import _lib
    _lib.load()
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], check=True, capture_output=True, text=True).stdout
    defined = {line.split()[-1] for line in out.splitlines() if line.split()[1:2] == ["T"]}
    missing = [f for f in REFERENCE_CDEF_FUNCTIONS if f not in defined]
    assert not missing, missing
    assert len(REFERENCE_CDEF_FUNCTIONS) == 27
    if os.path.exists("/root/reference/prox_tv/prox_tv_build.py"):   # (this container only) the list above IS the cdef's
        text = open("/root/reference/prox_tv/prox_tv_build.py").read()
        cdef = text[text.index('ffi.cdef("""'):text.index('""")')]
        cdef = re.sub(r"/\*.*?\*/|//[^\n]*", "", cdef, flags=re.S)
        names = set(re.findall(r"\b(\w+)\s*\(", cdef)) - {"cdef"}
        assert names == set(REFERENCE_CDEF_FUNCTIONS), names ^ set(REFERENCE_CDEF_FUNCTIONS)


def test_code_object_is_gfx950_only():
    """The library carries hand-written gfx950 code objects and nothing else (no multi-arch / fallback bundles)."""
    from proxtv_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    archs = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert archs == {b"gfx950"}, archs


def test_workspace_shims_are_callable_without_gpu():
    from proxtv_amd import _lib
    lib = _lib.load()
    ws = lib.newWorkspace(128)
    assert ws
    lib.resetWorkspace(ws)
    lib.freeWorkspace(ws)
    wa = lib.newWorkspaces(64, 3)
    assert wa
    lib.freeWorkspaces(wa, 3)


@pytest.mark.skipif(_has_gpu(), reason="checks the no-device failure mode")
def test_fails_loudly_without_device(capfd):
    import proxtv_amd
    from proxtv_amd import _lib
    lib = _lib.load()
    assert lib.proxtv_init(-1) != 0
    with pytest.raises(proxtv_amd.ProxTVError):
        proxtv_amd.tv1_2d(np.zeros((4, 4)), 0.1)
    with pytest.raises(proxtv_amd.ProxTVError):
        proxtv_amd.tv1_1d(np.zeros(4), 0.1)
    # straight through the C-ABI: reference CANCEL convention -- message, info[RC] = RC_ERROR, return 0, output untouched
    X = np.asfortranarray(np.ones((4, 4)))
    out, info = np.full((4, 4), 7.0, order="F"), np.zeros(3)
    rc = lib.DR2_TV(4, 4, X.ctypes.data, 0.1, 0.1, 1.0, 1.0, out.ctypes.data, 1, 0, info.ctypes.data)
    assert rc == 0 and info[2] == 3 and (out == 7.0).all()
    lam, norms, dims, ns = np.array([.1, .1]), np.ones(2), np.array([1., 2.]), np.array([4, 4], dtype=np.int32)
    rc = lib.PD2_TV(X.ctypes.data, lam.ctypes.data, norms.ctypes.data, dims.ctypes.data, out.ctypes.data,
                    info.ctypes.data, ns.ctypes.data, 2, 2, 1, 0)
    assert rc == 0 and info[2] == 3 and (out == 7.0).all()
    assert "no CPU fallback" in capfd.readouterr().out


def test_python_surface_argument_checks():
    """Same assertions as the reference, raised before anything touches the device
    (prox_tv/__init__.py:160-161, 245-246, 400-401, 473-477, 569-572)."""
    import proxtv_amd as ptv
    x = np.zeros((4, 5))
    with pytest.raises(AssertionError):
        ptv.tv1_1d(np.zeros(5), -1.0)
    with pytest.raises(AssertionError):
        ptv.tv1_1d(np.zeros(5), 1.0, method="nope")
    with pytest.raises(AssertionError):
        ptv.tv1w_1d(np.zeros(5), np.ones(5))            # needs n-1 weights
    with pytest.raises(AssertionError):
        ptv.tv1w_1d(np.zeros(5), -np.ones(4))
    with pytest.raises(AssertionError):
        ptv.tv1_2d(x, -0.1)
    with pytest.raises(AssertionError):
        ptv.tv1_2d(x, 0.1, method="nope")
    with pytest.raises(AssertionError):
        ptv.tv1w_2d(x, np.ones((4, 5)), np.ones((4, 4)))  # w_col must be (M-1, N)
    with pytest.raises(AssertionError):
        ptv.tv1w_2d(x, -np.ones((3, 5)), np.ones((4, 4)))
    with pytest.raises(AssertionError):
        ptv.tvgen(x, [1, 2], [1], [1, 1])
    with pytest.raises(AssertionError):
        ptv.tvgen(x, [1], [1], [1], n_threads=0)
    with pytest.raises(AssertionError):
        ptv.tvgen(x, [1], [1], [1], max_iters=-1)
    with pytest.raises(AssertionError):
        ptv.tvp_2d(x, 1, 1, 0.5, 1)
    # out-of-scope solvers say so instead of silently doing something else
    # (general p; p = 1 and p = 2 are implemented)
    for call in (lambda: ptv.tvp_1d(np.zeros(5), 1.0, 1.5), lambda: ptv.tvp_2d(x, 1, 1, 2, 1.5),
                 lambda: ptv.tvgen(x, [1, 1], [1, 2], [1, 3])):
        with pytest.raises(NotImplementedError):
            call()


def test_force_float_helpers():
    import proxtv_amd as ptv
    assert isinstance(ptv.force_float_scalar(3), float)
    a = np.arange(4.0)
    assert ptv.force_float_matrix(a) is a                          # float64 passes through as the same object (Q3)
    assert ptv.force_float_matrix([1, 2]).dtype == np.float64
    assert ptv.force_float_matrix(np.arange(3)).dtype == np.float64


def test_every_documented_knob_is_accepted():
    """The keys listed in include/proxtv_amd.h's knob comment are the ones proxtv_set_option knows (options are plain host state:
    no device needed) -- a knob that is documented but misspelt in the dispatch, or the other way round, shows here."""
    from proxtv_amd import _lib
    lib = _lib.load()
    text = open(os.path.join(ROOT, "include", "proxtv_amd.h")).read()
    block = text[text.index("/* Knobs"):text.index("int proxtv_set_option")]
    keys = sorted(set(re.findall(r'"([a-z_0-9]+)"', block)))
    assert {"runs", "chunk_mode", "deterministic", "dr_form", "xlink", "verbose", "profile", "certify"} <= set(keys), keys
    assert len(keys) <= 20, f"{len(keys)} knobs: every A/B that is settled takes its switch with it"
    for k in keys:
        before = lib.proxtv_set_option(k.encode(), 12345)
        assert lib.proxtv_set_option(k.encode(), before) == 12345, f"knob {k!r} is documented but not known to proxtv_set_option"
    assert lib.proxtv_set_option(b"no_such_knob", 1) == -1


def test_every_knob_has_its_environment_variable():
    """include/proxtv_amd.h: "each also has an environment variable PROXTV_<KEY> read at load time" -- one table in common.hip serves
    both ways of setting a knob.  A fresh process per check (the variables are read once)."""
    import subprocess
    import sys
    text = open(os.path.join(ROOT, "include", "proxtv_amd.h")).read()
    block = text[text.index("/* Knobs"):text.index("int proxtv_set_option")]
    keys = sorted(set(re.findall(r'"([a-z_0-9]+)"', block)))
    assert {"tile", "pin_seed", "why", "certify", "trace", "profile"} <= set(keys), keys
    code = ("import sys, json; sys.path.insert(0, %r); from proxtv_amd import _lib; lib = _lib.load(); "
            "print(json.dumps({k: lib.proxtv_set_option(k.encode(), 0) for k in %r}))" % (ROOT, keys))
    env = dict(os.environ)
    for i, k in enumerate(keys):
        env["PROXTV_" + k.upper()] = str(700 + i)
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, check=True).stdout
    import json
    got = json.loads(out.strip().splitlines()[-1])
    assert got == {k: 700 + i for i, k in enumerate(keys)}, got


def test_build_units_match_the_header():
    """proxtv_amd/build.py compiles csrc/sweep_unit.hip once per (op, weighted) pair; the pairs are listed twice -- there and in
    PTV_SWEEP_UNITS of csrc/sweep_kernels.hpp (what sweep.hip dispatches to) -- and a pair missing on either side is a link error
    or a dead object."""
    import re
    from proxtv_amd import build
    hdr = open(os.path.join(os.path.dirname(build.__file__), "csrc", "sweep_kernels.hpp")).read()
    block = hdr[hdr.index("#define PTV_SWEEP_UNITS(X)"):hdr.index("#define PTV_DECLARE_UNIT")]
    pairs = [(op, w == "true") for op, w in re.findall(r"X\((OP_\w+), (true|false)\)", block)]
    assert pairs == build.SWEEP_UNITS
    assert len(set(pairs)) == len(pairs) == 17
