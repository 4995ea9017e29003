"""The hand-written gfx950 assembly walks (proxtv_amd/csrc/walk_asm.hpp) interpreted on the CPU, lane by lane under an exec
mask, against their specification -- the C++ walk_interior of chunkcore.hpp, run by tests/host_harness.cpp.

The interpreter reads the assembly TEXT out of the header (the string literals of the `asm volatile` statement and its operand
lists) and implements the two dozen instructions the loops use: 64 lanes, VGPRs as 64-element arrays, SGPR pairs as 64-bit masks,
VALU writes and LDS reads masked by exec, comparisons writing zeros for inactive lanes.  It knows nothing about wait states or
counters -- hazards are the GPU suite's business -- but every mask, operand order, register reuse and address computation of a
loop is executed exactly as written.  Geometry: the along-fibre kernel's (64 consecutive chunks of one fibre, one LDS window)."""
import ctypes as C
import os
import re
import struct
import subprocess
import tempfile

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
HEADER = os.path.join(os.path.dirname(HERE), "proxtv_amd", "csrc", "walk_asm.hpp")
MASK64 = (1 << 64) - 1


@pytest.fixture(scope="module")
def harness():
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_asm_"), "libhost.so")
    flags = ["-DPTV_TABLE_RECIP"] if os.environ.get("PTV_EMULATE_TABLE") else []
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", *flags, "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_walk_interior.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


def asm_lines(function):
    """instructions of the asm statement inside `function` of walk_asm.hpp, and its operand names by class"""
    src = open(HEADER).read()
    k = src.index(function + "(")
    a = src.index("asm volatile(", k)
    body = src[a:src.index(");", src.index('"memory"', a))]
    text, ops = body.split("\n        :", 1)
    lines = []
    for piece in re.findall(r'"((?:[^"\\]|\\.)*)"', text):
        for ln in piece.split("\\n"):
            ln = ln.strip()
            if ln:
                lines.append(ln)
    return lines, re.findall(r'\[(\w+)\]\s*"([^"]+)"\((\w+)\)', ops)


class Machine:
    """64 lanes.  V: name -> list of 64 python numbers ; S: name -> int (64-bit mask, or a scalar value)."""

    def __init__(self, lds):
        self.lds = lds          # list of doubles, byte address = 8 * index
        self.V, self.S = {}, {}
        self.exec = MASK64
        self.vcc = 0
        self.trips = 0

    def active(self):
        return [l for l in range(64) if (self.exec >> l) & 1]

    def src(self, tok, lane, as_float):
        neg = tok.startswith("-")
        if neg:
            tok = tok[1:]
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        if m:
            n = m.group(1)
            v = self.V[n][lane] if n in self.V else self.S[n]
        else:
            v = float(tok) if as_float else int(tok, 0)
        return -v if neg else v

    def mask(self, tok):
        if tok == "exec":
            return self.exec
        if tok == "vcc":
            return self.vcc
        m = re.fullmatch(r"%\[(\w+)\]", tok)
        return self.S[m.group(1)] if m else int(tok, 0)

    def set_mask(self, tok, v):
        v &= MASK64
        if tok == "exec":
            self.exec = v
        elif tok == "vcc":
            self.vcc = v
        else:
            self.S[re.fullmatch(r"%\[(\w+)\]", tok).group(1)] = v

    def wv(self, tok, lane, v):
        self.V[re.fullmatch(r"%\[(\w+)\]", tok).group(1)][lane] = v

    def run(self, lines, imm):
        labels = {ln[:-1]: k for k, ln in enumerate(lines) if ln.endswith(":")}
        pc = 0
        while pc < len(lines):
            ln = lines[pc]
            pc += 1
            if ln.endswith(":"):
                self.trips += 1
                assert self.trips < 100000, "runaway loop"
                continue
            op, _, rest = ln.partition(" ")
            for k, v in imm.items():
                rest = rest.replace(f"%[{k}]", str(v))
            a = [t.strip() for t in rest.split(",")] if rest else []
            u32 = lambda x: int(x) & 0xffffffff
            i32 = lambda x: ((int(x) & 0xffffffff) ^ 0x80000000) - 0x80000000
            if op in ("s_waitcnt", "s_nop"):
                continue
            if op == "s_cbranch_execnz":
                if self.exec:
                    pc = labels[a[0]]
                continue
            if op == "s_mov_b64":
                self.set_mask(a[0], self.mask(a[1]))
            elif op == "s_and_b64":
                self.set_mask(a[0], self.mask(a[1]) & self.mask(a[2]))
            elif op == "s_or_b64":
                self.set_mask(a[0], self.mask(a[1]) | self.mask(a[2]))
            elif op == "s_andn2_b64":
                self.set_mask(a[0], self.mask(a[1]) & ~self.mask(a[2]))
            elif op == "ds_read_b64":
                off = 0
                addr = a[1]
                if " offset:" in addr:
                    addr, o = addr.split(" offset:")
                    off = int(o)
                for l in self.active():
                    byte = u32(self.src(addr.strip(), l, False)) + off
                    assert byte % 8 == 0 and 0 <= byte // 8 < len(self.lds), (ln, l, byte)
                    self.wv(a[0], l, self.lds[byte // 8])
            elif op.startswith("v_cmp_"):
                _, _, rel, ty = op.split("_")[:4]
                f = {"lt": lambda x, y: x < y, "gt": lambda x, y: x > y, "ge": lambda x, y: x >= y, "le": lambda x, y: x <= y}[rel]
                conv = {"f64": float, "i32": i32, "u32": u32}[ty]
                bits = 0
                for l in self.active():
                    if f(conv(self.src(a[1], l, ty == "f64")), conv(self.src(a[2], l, ty == "f64"))):
                        bits |= 1 << l
                self.set_mask(a[0], bits)
            else:
                for l in self.active():
                    s = lambda k, fl=False: self.src(a[k], l, fl)
                    if op == "v_add_f64":
                        r = np.float64(s(1, True)) + np.float64(s(2, True))
                    elif op == "v_mul_f64":
                        r = np.float64(s(1, True)) * np.float64(s(2, True))
                    elif op == "v_fma_f64":   # one rounding
                        r = fma(s(1, True), s(2, True), s(3, True))
                    elif op == "v_min_f64":
                        r = min(s(1, True), s(2, True))
                    elif op == "v_max_f64":
                        r = max(s(1, True), s(2, True))
                    elif op == "v_rcp_f64":
                        r = 1.0 / s(1, True)
                    elif op == "v_cvt_f64_i32":
                        r = float(i32(s(1)))
                    elif op in ("v_mov_b64", "v_mov_b32"):
                        r = s(1, op == "v_mov_b64")
                    elif op == "v_add_u32":
                        r = u32(s(1) + s(2))
                    elif op == "v_sub_u32":
                        r = u32(s(1) - s(2))
                    elif op == "v_subrev_u32":
                        r = u32(s(2) - s(1))
                    elif op == "v_mad_u32_u24":
                        r = u32((u32(s(1)) & 0xffffff) * (u32(s(2)) & 0xffffff) + s(3))
                    elif op == "v_lshl_add_u32":
                        r = u32((u32(s(1)) << (s(2) & 31)) + s(3))
                    elif op == "v_lshl_or_b32":
                        r = u32((u32(s(1)) << (s(2) & 31)) | u32(s(3)))
                    elif op == "v_lshlrev_b32_e64":
                        r = u32(u32(s(2)) << (u32(s(1)) & 31))
                    elif op == "v_or_b32":
                        r = u32(s(1)) | u32(s(2))
                    elif op == "v_cndmask_b32_e64":
                        r = s(2) if (self.mask(a[3]) >> l) & 1 else s(1)
                    else:
                        raise NotImplementedError(ln)
                    self.wv(a[0], l, float(r) if isinstance(r, (float, np.floating)) else r)


def fma(a, b, c):
    """a * b + c with one rounding (exact rational arithmetic on the operands)"""
    from fractions import Fraction
    return float(Fraction(a) * Fraction(b) + Fraction(c))


C17, H, T = 17, 16, 8


def segment_case(rng, lam, weighted):
    """one 64-chunk segment of a fibre, the along-fibre kernel's window: rows [lo, hi) with lo = seg_s - H"""
    n = H + 64 * C17 + T + 40
    kind = int(rng.integers(0, 3))
    y = rng.standard_normal(n) if kind == 0 else (np.repeat(rng.standard_normal(n // 5 + 1), 5)[:n] + 0.1 * rng.standard_normal(n)
                                                  if kind == 1 else np.round(rng.standard_normal(n) * 2))
    w = rng.uniform(0.2 * lam, 1.8 * lam, n - 1) if weighted else None
    return np.ascontiguousarray(y), w


def spec_and_emulated(harness, function, y, w, lam, tab_entries=0, entry_exec=MASK64):
    n = y.size
    seg_s = H
    lo, hi = seg_s - H, seg_s + 64 * C17 + T
    lim = min(n - 1, hi)
    lines, operands = asm_lines(function)
    # ---- initial state of every lane: a free-end start H samples before its chunk (walker_start)
    st = []
    for l in range(64):
        cs = seg_s + l * C17
        start = cs - H
        r0 = w[start] if w is not None else lam
        st.append(dict(cs=cs, ce=cs + C17, lo=-r0 + y[start], hi=r0 + y[start], hlo=0.0, hhi=0.0, i=start, k0=start - 1, klo=start, khi=start))
    # ---- specification
    want = []
    for s in st:
        wk = np.array([s["lo"], s["hi"], s["hlo"], s["hhi"]])
        wi = np.array([s["i"], s["k0"], s["klo"], s["khi"]], dtype=np.int32)
        rc = np.zeros(6, dtype=np.uint32)
        harness.host_walk_interior(y.ctypes.data, None if w is None else w.ctypes.data, n, lo, lim, s["cs"], s["ce"], lam,
                                   wk.ctypes.data, wi.ctypes.data, rc.ctypes.data)
        want.append((wk, wi, rc))
    # ---- the machine: LDS = [sample plane | penalty plane | reciprocal table]
    rows = hi - lo + 2
    lds = [float(v) for v in y[lo:hi]] + [1e300, 1e300]
    wdelta = 8 * rows
    lds += ([float(v) for v in w[lo:hi]] + [0.0, 0.0]) if w is not None else []
    rtab = 8 * len(lds)
    lds += [0.0] + [1.0 / k for k in range(1, max(tab_entries, 1))]
    m = Machine(lds)
    wlo = lo
    for name in ("lo", "hi", "hlo", "hhi"):
        m.V[name] = [s[name] for s in st]
    for name in ("i", "k0", "klo", "khi"):
        m.V[name] = [s[name] - wlo for s in st]
    m.V["ai"] = [8 * (s["i"] - wlo) for s in st]
    m.V["awi"] = [8 * (s["i"] - wlo) + wdelta for s in st]
    m.V["yi"] = [float(y[s["i"]]) for s in st]
    m.V["r"] = [float(w[s["i"]]) if w is not None else 0.0 for s in st]
    m.V["sp"] = [s["i"] - s["k0"] for s in st]
    m.V["abase"] = [0] * 64
    for name in ("ends", "types", "mine", "next", "last", "doneflag"):
        m.V[name] = [0] * 64
    for name, constraint, _ in operands:
        if "v" in constraint and name not in m.V:
            m.V[name] = [0] * 64                  # temporaries ("=&v")
    # cs / ce differ from lane to lane here (one chunk per lane); the kernel passes them in SGPRs because there a wave shares
    # them -- the interpreter resolves %[csr] etc. per lane through V
    m.V["csr"] = [s["cs"] - wlo for s in st]
    m.V["cer"] = [s["ce"] - wlo for s in st]
    m.V["cem1r"] = [s["ce"] - 1 - wlo for s in st]
    m.S.update(lam=lam, nlam=-lam, lam2=2 * lam, nlam2=2 * (-lam), pbs=8, limr=lim - wlo, span=C17, wlo=wlo, wd=wdelta, rtab=rtab,
               tsz=tab_entries)
    for name in ("msave", "mlive", "mcv", "mfv", "mb", "mth", "mtl", "mdone", "m1", "m2", "m3"):
        m.S[name] = 0
    for name, constraint, _ in operands:          # every operand of the statement is bound
        assert name in m.V or name in m.S or constraint == "n", name
    m.exec = entry_exec
    m.run(lines, {"pb": 8, "pb2": 16})
    assert m.exec == entry_exec                   # the statement restores the mask it was entered with
    got = []
    for l in range(64):
        wk = np.array([m.V[k][l] for k in ("lo", "hi", "hlo", "hhi")])
        wi = np.array([m.V[k][l] + wlo for k in ("i", "k0", "klo", "khi")], dtype=np.int64)
        rc = np.array([m.V["ends"][l], m.V["types"][l], m.V["mine"][l], m.V["next"][l], m.V["last"][l], m.V["doneflag"][l]], dtype=np.uint64)
        got.append((wk, wi, rc))
    return want, got, m.trips


def compare(want, got, exact):
    for l, ((wk, wi, rc), (gk, gi, gc)) in enumerate(zip(want, got)):
        assert list(wi) == list(gi), (l, wi, gi)
        assert [int(v) for v in rc] == [int(v) for v in gc], (l, rc, gc)
        if exact:
            assert [struct.pack("d", v) for v in wk] == [struct.pack("d", float(v)) for v in gk], (l, wk, gk)
        else:
            assert np.allclose(wk, gk, rtol=1e-13, atol=1e-13), (l, wk, gk)


@pytest.mark.parametrize("lam", [0.1, 0.4, 1.0])
def test_unweighted_assembly_walk_follows_its_specification(harness, lam):
    """(the device divides by v_rcp_f64 + Newton + residual, the specification by `/`: decisions and codes must agree exactly,
    heights and slopes to rounding)"""
    rng = np.random.default_rng(int(lam * 100))
    total = 0
    for _ in range(6):
        y, _w = segment_case(rng, lam, False)
        want, got, trips = spec_and_emulated(harness, "void walk_interior_asm", y, None, lam)
        compare(want, got, exact=False)
        total += trips
    assert total > 100


@pytest.mark.parametrize("lam", [0.1, 0.5])
def test_weighted_assembly_walk_follows_its_specification(harness, lam):
    rng = np.random.default_rng(7 + int(lam * 100))
    for _ in range(4):
        y, w = segment_case(rng, lam, True)
        want, got, trips = spec_and_emulated(harness, "void walk_interior_asm_w", y, w, 0.0)
        compare(want, got, exact=False)


# ---- the staged table walks (walk_interior_asm_tab / _w_tab): against the specification built with -DPTV_TABLE_RECIP, whose
# quotient is the same single product with the correctly rounded reciprocal -- so the bar is bit equality.
@pytest.fixture(scope="module")
def harness_table():
    out = os.path.join(tempfile.mkdtemp(prefix="ptv_asmt_"), "libhost.so")
    subprocess.run(["g++", "-O2", "-std=c++17", "-ffp-contract=off", "-fPIC", "-shared", "-pthread", "-DPTV_TABLE_RECIP", "-o", out,
                    os.path.join(HERE, "host_harness.cpp")], check=True)
    lib = C.CDLL(out)
    lib.host_walk_interior.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                       C.c_void_p, C.c_void_p, C.c_void_p]
    return lib


@pytest.mark.parametrize("lam", [0.1, 0.4, 1.0])
def test_table_walk_follows_its_specification_bit_for_bit(harness_table, lam):
    rng = np.random.default_rng(11 + int(lam * 100))
    for _ in range(6):
        y, _w = segment_case(rng, lam, False)
        want, got, trips = spec_and_emulated(harness_table, "void walk_interior_asm_tab", y, None, lam, tab_entries=48)
        compare(want, got, exact=True)


@pytest.mark.parametrize("lam", [0.1, 0.5])
def test_weighted_table_walk_follows_its_specification_bit_for_bit(harness_table, lam):
    rng = np.random.default_rng(13 + int(lam * 100))
    for _ in range(4):
        y, w = segment_case(rng, lam, True)
        want, got, trips = spec_and_emulated(harness_table, "void walk_interior_asm_w_tab", y, w, 0.0, tab_entries=48)
        compare(want, got, exact=True)


def test_table_walk_leaves_the_loop_where_the_table_ends(harness_table):
    """Robust instantiations: a lane whose span reaches the table's end stops there (the slow tail takes over) -- with a short
    table on long pieces every lane's state must still be a state of the specification's walk: continue the specification from
    it and from the start, and arrive at the same place."""
    rng = np.random.default_rng(5)
    y, _w = segment_case(rng, 3.0, False)
    want, got, trips = spec_and_emulated(harness_table, "void walk_interior_asm_tab", y, None, 3.0, tab_entries=12)
    n = y.size
    lo, hi = 0, H + 64 * C17 + T
    lim = min(n - 1, hi)
    stopped = 0
    for l, ((wk, wi, rc), (gk, gi, gc)) in enumerate(zip(want, got)):
        cs = H + l * C17
        if list(wi) == list(gi):
            continue
        stopped += 1
        assert gi[0] - gi[1] >= 12 or gc[5], (l, gi)          # it left the loop because of its span
        k2, i2, c2 = np.array(gk, dtype=np.float64), np.array(gi, dtype=np.int32), np.array(gc, dtype=np.uint32)
        harness_table.host_walk_interior(y.ctypes.data, None, n, lo, lim, cs, cs + C17, 3.0, k2.ctypes.data, i2.ctypes.data, c2.ctypes.data)
        assert list(i2) == list(wi) and [int(v) for v in c2] == [int(v) for v in rc], (l, i2, wi)
    assert stopped > 0

def test_assembly_walk_in_the_tile_geometry(harness):
    """The 64-fibre tile: lane l walks ITS fibre's column of a window of pitch 64 (row r of lane l at byte 8 (64 r + l)), every
    lane the same chunk -- the addressing the strided sweeps use (`abase` differs per lane, the row pitch is 512 bytes)."""
    rng = np.random.default_rng(21)
    lam, Hh, Cc, NWw, Tt = 0.3, 16, 16, 8, 8
    rows = Hh + NWw * Cc + Tt
    n = rows + 30
    Y = np.ascontiguousarray(rng.standard_normal((64, n)))            # one fibre per lane
    lines, operands = asm_lines("void walk_interior_asm")
    for wave in (0, 3, 7):                                              # which chunk of the block the wave owns
        lo, hi = 0, rows
        cs = Hh + wave * Cc
        ce, start, lim = cs + Cc, cs - Hh, min(n - 1, rows)
        want = []
        for l in range(64):
            y = np.ascontiguousarray(Y[l])
            wk = np.array([-lam + y[start], lam + y[start], 0.0, 0.0])
            wi = np.array([start, start - 1, start, start], dtype=np.int32)
            rc = np.zeros(6, dtype=np.uint32)
            harness.host_walk_interior(y.ctypes.data, None, n, lo, lim, cs, ce, lam, wk.ctypes.data, wi.ctypes.data, rc.ctypes.data)
            want.append((wk, wi, rc))
        lds = [float(Y[l, r]) if r < rows else 1e300 for r in range(rows + 2) for l in range(64)]
        m = Machine(lds)
        m.V.update(lo=[-lam + Y[l, start] for l in range(64)], hi=[lam + Y[l, start] for l in range(64)], hlo=[0.0] * 64, hhi=[0.0] * 64,
                   i=[start] * 64, k0=[start - 1] * 64, klo=[start] * 64, khi=[start] * 64, ai=[8 * (64 * start + l) for l in range(64)],
                   yi=[float(Y[l, start]) for l in range(64)], abase=[8 * l for l in range(64)])
        for name in ("ends", "types", "mine", "next", "last", "doneflag"):
            m.V[name] = [0] * 64
        for name, constraint, _ in operands:
            if "v" in constraint and name not in m.V:
                m.V[name] = [0] * 64
        m.S.update(lam=lam, nlam=-lam, lam2=2 * lam, nlam2=2 * (-lam), pbs=512, limr=lim, csr=cs, cer=ce, cem1r=ce - 1, span=Cc, wlo=0)
        for name in ("msave", "mlive", "mcv", "mfv", "mb", "mth", "mtl", "mdone", "m1", "m2", "m3"):
            m.S[name] = 0
        m.run(lines, {"pb": 512})
        got = [(np.array([m.V[k][l] for k in ("lo", "hi", "hlo", "hhi")]), np.array([m.V[k][l] for k in ("i", "k0", "klo", "khi")]),
                np.array([m.V[k][l] for k in ("ends", "types", "mine", "next", "last", "doneflag")], dtype=np.uint64)) for l in range(64)]
        compare(want, got, exact=False)


def test_assembly_walk_entered_with_some_lanes_off(harness):
    """A wave enters the loop with the lanes that have no chunk switched off (the kernels call it inside `if (has_chunk)`): those
    lanes' registers must come out untouched, the others as before."""
    rng = np.random.default_rng(33)
    y, _w = segment_case(rng, 0.3, False)
    entry = int(rng.integers(1, 1 << 62)) | 1
    want, got, _ = spec_and_emulated(harness, "void walk_interior_asm", y, None, 0.3, entry_exec=entry)
    on = [l for l in range(64) if (entry >> l) & 1]
    compare([want[l] for l in on], [got[l] for l in on], exact=False)
    for l in range(64):
        if not (entry >> l) & 1:
            cs = H + l * C17
            assert list(got[l][1]) == [cs - H, cs - H - 1, cs - H, cs - H] and not any(int(v) for v in got[l][2]), l
