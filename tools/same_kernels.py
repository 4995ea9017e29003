#!/usr/bin/env python3
"""Which kernels of two builds of a HIP translation unit are the same machine code?

    python tools/same_kernels.py old/sweep.o new/sweep.o [substring ...]

Unbundles the gfx950 code object of both objects (llvm-objdump --offloading), disassembles it (no addresses, no encodings: the text
of a kernel does not depend on where it sits) and compares kernel by kernel.  Prints one line per kernel whose demangled name holds
every given substring (all kernels if none) and a summary: counters measured on one build are counters of the other for exactly the
kernels reported SAME -- what lets profiles/*_pmc_traffic.json name a second build it holds for.
"""
import hashlib
import os
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def kernels(obj):
    with tempfile.TemporaryDirectory() as d:
        local = os.path.join(d, "u.o")
        with open(obj, "rb") as src, open(local, "wb") as dst:
            dst.write(src.read())
        subprocess.run([f"{LLVM}/llvm-objdump", "--offloading", local], cwd=d, check=True, capture_output=True)
        code = [f for f in os.listdir(d) if "amdgcn" in f]
        if len(code) != 1:
            raise SystemExit(f"{obj}: expected one gfx950 code object, found {code}")
        text = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--no-leading-addr", os.path.join(d, code[0])],
                              check=True, capture_output=True, text=True).stdout
    out, name, body = {}, None, []
    for line in text.splitlines():
        if line.startswith("<") and line.endswith(">:"):
            if name:
                out[name] = hashlib.sha1("\n".join(body).encode()).hexdigest()[:12]
            name, body = line[1:-2], []
        elif name:
            body.append(line.split("//")[0].rstrip())
    if name:
        out[name] = hashlib.sha1("\n".join(body).encode()).hexdigest()[:12]
    return out


def demangle(names):
    try:
        p = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True)
    except FileNotFoundError:
        return {n: n for n in names}
    return dict(zip(names, p.stdout.splitlines())) if p.returncode == 0 else {n: n for n in names}


def main():
    old, new, want = sys.argv[1], sys.argv[2], sys.argv[3:]
    a, b = kernels(old), kernels(new)
    names = demangle(sorted(set(a) | set(b)))
    same = differ = 0
    for n in sorted(set(a) | set(b)):
        verdict = "SAME" if a.get(n) == b.get(n) else ("only in one build" if n not in a or n not in b else "DIFFERENT")
        same += verdict == "SAME"
        differ += verdict != "SAME"
        pretty = names[n]
        if (want and all(w in pretty for w in want)) or (not want and verdict != "SAME"):
            print(f"{verdict:18s} {a.get(n, '-'):12s} {b.get(n, '-'):12s} {pretty[:200]}")
    print(f"# {same} kernels the same machine code, {differ} not")


if __name__ == "__main__":
    main()
