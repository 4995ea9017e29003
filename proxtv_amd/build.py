"""Build libproxtv_amd.so (HIP, gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python -m proxtv_amd.build [--force] [--report]
    python -m proxtv_amd.build --variant NAME [-- extra hipcc flags]     (A/B builds: proxtv_amd/build/lib_NAME.so)

hipcc cross-compiles for gfx950 without a GPU present.  The .so lands next to this file (git-ignored, but it
travels to the GPU box with the gpurun snapshot).

The sweep kernels are templates on (op, weighted); csrc/sweep_unit.hip is compiled once per pair (seventeen objects), so the
library builds in about a minute on eight cores instead of the 4.5 minutes the one translation unit took, and the wall time of
every object is kept in build/build_times.json (python -m proxtv_amd.build --report prints it).
"""
import json
import os
import subprocess
import sys
import time
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libproxtv_amd.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: keep a*b+c as two roundings so the device arithmetic matches the reference's CPU build
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]

# (op, weighted) pairs of csrc/sweep_kernels.hpp: PTV_SWEEP_UNITS -- tests/test_cabi_cpu.py checks the two lists agree
SWEEP_UNITS = [("OP_PROX", False), ("OP_PROX", True), ("OP_DR_COL", False), ("OP_DR_COL", True), ("OP_DR_COL_FINAL", False),
               ("OP_DR_COL_FINAL", True), ("OP_DR_ROW", False), ("OP_DR_ROW", True), ("OP_DR_ROW_FINAL", False),
               ("OP_DRW_ROW_FINAL", True), ("OP_PD2_A", False), ("OP_PD2_B", False), ("OP_YANG", False), ("OP_DR_COL_V", False),
               ("OP_DR_COL_V", True), ("OP_DR_ROW_V", False), ("OP_DR_ROW_V", True)]
PLAIN_UNITS = ["common", "sweep", "pin", "pinlong", "pointwise", "tv2", "solvers", "cabi"]


def units():
    """(object name, source file, extra flags), the heaviest first so that the pool's tail is short."""
    out = []
    for op, w in SWEEP_UNITS:
        out.append((f"sweep_unit_{op[3:].lower()}_{'w' if w else 'u'}", "sweep_unit",
                    [f"-DPTV_UNIT_OP={op}", f"-DPTV_UNIT_W={'true' if w else 'false'}"]))
    out += [(u, u, []) for u in PLAIN_UNITS]
    return out


def build_id():
    """Short content hash of the kernel sources (csrc/ + the C header): what a measurement under profiles/ is keyed to,
    so that bench.py never quotes counters of another build (there is no .git on the GPU box)."""
    import hashlib
    h = hashlib.sha1()
    files = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".hpp")))
    files.append(os.path.join(HERE, "..", "include", "proxtv_amd.h"))
    for f in files:
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()[:12]


def _newer(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _compile_all(objdir, extra, force, verbose, only=None):
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "proxtv_amd.h"))
    times = {}

    def compile_one(unit):
        name, source, flags = unit
        src, obj = os.path.join(CSRC, source + ".hip"), os.path.join(objdir, name + ".o")
        if force or _newer(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + flags + extra + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            t0 = time.perf_counter()
            subprocess.run(cmd, check=True)
            times[name] = round(time.perf_counter() - t0, 1)
        return obj

    todo = [u for u in units() if only is None or only(u)]
    workers = min(len(todo), max(1, (os.cpu_count() or 2)))
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=workers) as ex:
        objs = list(ex.map(compile_one, todo))
    if times:
        times["_wall"] = round(time.perf_counter() - t0, 1)
        times["_workers"] = workers
        with open(os.path.join(objdir, "build_times.json"), "w") as fh:
            json.dump(times, fh, indent=1, sort_keys=True)
    return objs


def build(force=False, verbose=False):
    objs = _compile_all(OBJ, [], force, verbose)
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


def build_variant(name, extra, verbose=False, sweeps_only=True):
    """A/B build: the sweep objects (or everything) recompiled with `extra` flags into build/var_<name>/, linked with the default
    build's other objects into build/lib_<name>.so.  Run with  PROXTV_DEBUG_ALT_LIB=1 PROXTV_LIB=.../lib_<name>.so  (tools/ab_run.py)."""
    build(verbose=verbose)
    vdir = os.path.join(OBJ, "var_" + name)
    pick = (lambda u: u[1] in ("sweep_unit", "sweep")) if sweeps_only else None
    # the objects of a variant are keyed on its flags: the up-to-date test of _compile_all looks at sources and headers only, so the
    # same NAME rebuilt with other flags would silently relink the old objects and an A/B run would measure the wrong code
    os.makedirs(vdir, exist_ok=True)
    stamp, want = os.path.join(vdir, "flags.txt"), " ".join(extra) + ("\n" if sweeps_only else " [all units]\n")
    stale = True
    if os.path.exists(stamp):
        with open(stamp) as fh:
            stale = fh.read() != want
    objs = _compile_all(vdir, list(extra), stale, verbose, only=pick)
    with open(stamp, "w") as fh:
        fh.write(want)
    if sweeps_only:
        objs += [os.path.join(OBJ, u + ".o") for u in PLAIN_UNITS if u != "sweep"]
    lib = os.path.join(OBJ, f"lib_{name}.so")
    subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs, check=True)
    return lib


def report():
    path = os.path.join(OBJ, "build_times.json")
    if not os.path.exists(path):
        return "no build_times.json (nothing was compiled by the last build)"
    with open(path) as fh:
        t = json.load(fh)
    rows = sorted(((v, k) for k, v in t.items() if not k.startswith("_")), reverse=True)
    lines = [f"hipcc wall time per object ({t.get('_workers')} workers, {t.get('_wall')} s in all; sum {sum(v for v, _ in rows):.0f} s)"]
    lines += [f"  {k:34s} {v:7.1f} s" for v, k in rows]
    lines.append(f"  libproxtv_amd.so {os.path.getsize(LIB) / 2**20:.1f} MiB" if os.path.exists(LIB) else "")
    return "\n".join(lines)


if __name__ == "__main__":
    argv = sys.argv[1:]
    if "--variant" in argv:
        i = argv.index("--variant")
        extra = argv[argv.index("--") + 1:] if "--" in argv else []
        print(build_variant(argv[i + 1], extra, verbose="--verbose" in argv, sweeps_only="--all-units" not in argv))
    elif "--report" in argv:
        print(report())
    else:
        print(build(force="--force" in argv, verbose=True))
