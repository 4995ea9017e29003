"""Build libproxtv_amd.so (HIP, gfx950) in-tree with hipcc.  Used by __graft_entry__.build() and by hand:

    python -m proxtv_amd.build [--force]

hipcc cross-compiles for gfx950 without a GPU present.  The .so lands next to this file (git-ignored, but it
travels to the GPU box with the gpurun snapshot).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
LIB = os.path.join(HERE, "libproxtv_amd.so")
UNITS = ["common", "sweep", "pin", "pinlong", "pointwise", "tv2", "solvers", "cabi"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# -ffp-contract=off: keep a*b+c as two roundings so the device arithmetic matches the reference's CPU build
FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fPIC", "-ffp-contract=off", "-fno-gpu-rdc",
         "-Wall", "-Wno-unused-function"]


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


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hpp")]
    headers.append(os.path.join(HERE, "..", "include", "proxtv_amd.h"))

    def compile_one(u):
        src, obj = os.path.join(CSRC, u + ".hip"), os.path.join(OBJ, u + ".o")
        if force or _newer(obj, [src] + headers):
            cmd = [HIPCC] + FLAGS + ["-c", src, "-o", obj]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.run(cmd, check=True)
        return obj

    with ThreadPoolExecutor(max_workers=len(UNITS)) as ex:
        objs = list(ex.map(compile_one, UNITS))
    if force or _newer(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
