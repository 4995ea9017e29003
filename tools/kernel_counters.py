#!/usr/bin/env python3
"""Per-kernel evidence for every hot kernel of the library in one table: duration, registers / LDS / scratch, HBM traffic
(FETCH_SIZE / WRITE_SIZE, calibrated) and SQ instruction counters -- rocprofv3 --kernel-trace --pmc, one pass per counter set,
all workloads in ONE process per pass (tools/profile_cases.py a+b+c) so that a pass costs one interpreter start.

    python tools/kernel_counters.py collect gpurun_out/kc [cases]     # on the GPU box: five rocprofv3 runs
    python tools/kernel_counters.py report  gpurun_out/kc             # -> text table on stdout
    python tools/kernel_counters.py traffic gpurun_out/kc             # -> the json bench.py reads (profiles/rNN_pmc_traffic.json)

Calibration (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE are the L2's fabric request counters in KiB and, on gfx950,
FETCH_SIZE under-reports wide coalesced reads.  The sweep kernels move 8 bytes per lane, so the factor is measured in the same
process on `calib_copy_kernel` (same width, 1 GiB -- four times the Infinity Cache): true traffic 1 GiB read + 1 GiB written.
"""
import collections
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = "calib+dr0.1+c3+c4+c4y+dr0.7+dr1.0"
SETS = {
    "fetch": ["FETCH_SIZE"],
    "write": ["WRITE_SIZE"],
    "insts": ["SQ_INSTS_VALU", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD"],
    "waves": ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_LDS_BANK_CONFLICT"],
    "waits": ["SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"],
}
CALIB_BYTES = 128 * 1024 * 1024 * 8


def collect(outdir, cases):
    os.makedirs(outdir, exist_ok=True)
    only = os.environ.get("KC_SETS")   # e.g. KC_SETS=insts,waves : a subset of the passes
    for tag, counters in SETS.items():
        if only and tag not in only.split(","):
            continue
        cmd = ["rocprofv3", "--kernel-trace", "--pmc"] + counters + ["-d", os.path.join(outdir, tag), "-o", "k", "--",
                                                                      sys.executable, os.path.join(ROOT, "tools", "profile_cases.py"), cases, "2"]
        print("+", " ".join(cmd), flush=True)
        subprocess.run(cmd, check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=600)
    with open(os.path.join(outdir, "cases.txt"), "w") as fh:
        fh.write(cases + "\n")


def _db(outdir, tag):
    for dirpath, _, files in os.walk(os.path.join(outdir, tag)):
        for f in files:
            if f.endswith("_results.db"):
                return os.path.join(dirpath, f)
    return None


def short(name):
    """ptv::swp::sweep_chunk_kernel<3, true, false, ...>(args) -> sweep_chunk_kernel<3,true,false,...>"""
    n = name.split("(ptv::")[0].split("(double")[0].split("(unsigned")[0]
    for pre in ("void ", "ptv::swp::", "ptv::(anonymous namespace)::", "ptv::"):
        n = n.replace(pre, "")
    return n.replace(", ", ",")


def load(outdir):
    per = collections.defaultdict(dict)    # kernel -> column -> value
    first = True
    for tag, counters in SETS.items():
        path = _db(outdir, tag)
        if not path:
            continue
        c = sqlite3.connect(path)
        if first:
            for name, n, dur, vg, ag, lds, scr, wg in c.execute(
                    "select name, count(*), avg(duration), max(vgpr_count), max(accum_vgpr_count), max(lds_size), max(scratch_size), "
                    "max(workgroup_x) from kernels group by name"):
                per[name].update(calls=n, avg_us=dur / 1e3, vgpr=vg, agpr=ag, lds=lds, scratch=scr, wg=wg)
            first = False
        for name, counter, total, n in c.execute("select kernel_name, counter_name, sum(value), count(*) from counters_collection "
                                                 "group by kernel_name, counter_name"):
            per[name][counter] = total / n
    calib = {"FETCH_SIZE": None, "WRITE_SIZE": None}
    for name, d in per.items():
        if "calib_copy_kernel" in name:
            for k in calib:
                if d.get(k):
                    calib[k] = CALIB_BYTES / (d[k] * 1024.0)
    return per, calib


def report(outdir):
    per, calib = load(outdir)
    cases = open(os.path.join(outdir, "cases.txt")).read().strip() if os.path.exists(os.path.join(outdir, "cases.txt")) else "?"
    sys.path.insert(0, ROOT)
    from proxtv_amd import build as _build
    print(f"# tools/kernel_counters.py: rocprofv3 --kernel-trace --pmc <set>, one pass per counter set ({', '.join(SETS)}), "
          f"python tools/profile_cases.py {cases} 2 ; build id {_build.build_id()}")
    print(f"# calibration on calib_copy_kernel (8 B/lane, 1 GiB): FETCH_SIZE x {calib['FETCH_SIZE'] or float('nan'):.3f}, WRITE_SIZE x "
          f"{calib['WRITE_SIZE'] or float('nan'):.3f} (counters in KiB; the corrected figures are below, in MB per launch)")
    print("# per launch: us = average duration; rd / wr = HBM-side bytes; VALU / SALU / LDS = wave-instructions issued (millions); "
          "waves; wcyc = wave-cycles (M); busy = SQ busy cycles (M); bank = LDS bank-conflict cycles (M); "
          "issue% = VALU x 4 cycles / (1024 SIMDs x duration x 2.4 GHz): how busy the vector pipes are ; wait% = SQ_WAIT_INST_ANY / wave-cycles")
    hdr = f"{'kernel':74s} {'calls':>5s} {'us':>8s} {'VGPR':>4s} {'LDS_B':>6s} {'scr':>4s} {'rd_MB':>7s} {'wr_MB':>7s} {'VALU':>6s} {'SALU':>6s} {'LDS':>5s} {'waves':>6s} {'wcyc':>6s} {'busy':>6s} {'bank':>5s} {'issu%':>5s} {'wait%':>5s}"
    print(hdr)
    rows = sorted(per.items(), key=lambda kv: -(kv[1].get("calls", 0) * kv[1].get("avg_us", 0)))
    total = sum(d.get("calls", 0) * d.get("avg_us", 0) for _, d in rows) or 1.0
    for name, d in rows:
        if d.get("calls", 0) * d.get("avg_us", 0) < 0.004 * total or "at::native" in name or name.startswith("__amd"):
            continue
        g = lambda k, s=1.0: (d[k] * s if d.get(k) is not None else float("nan"))
        rd = g("FETCH_SIZE", 1024.0 * (calib["FETCH_SIZE"] or 1.0) / 1e6)
        wr = g("WRITE_SIZE", 1024.0 * (calib["WRITE_SIZE"] or 1.0) / 1e6)
        wc = g("SQ_WAVE_CYCLES")
        valu_pct = 100.0 * g("SQ_INSTS_VALU") * 4 / (1024 * d.get("avg_us", 0) * 2400.0) if d.get("avg_us") else float("nan")
        wait_pct = 100.0 * g("SQ_WAIT_INST_ANY") / wc if wc == wc and wc > 0 else float("nan")
        print(f"{short(name)[:74]:74s} {d.get('calls', 0):5d} {d.get('avg_us', 0):8.2f} {d.get('vgpr', -1):4d} {d.get('lds', -1):6d} {d.get('scratch', -1):4d} "
              f"{rd:7.1f} {wr:7.1f} {g('SQ_INSTS_VALU', 1e-6):6.2f} {g('SQ_INSTS_SALU', 1e-6):6.2f} {g('SQ_INSTS_LDS', 1e-6):5.2f} "
              f"{g('SQ_WAVES'):6.0f} {g('SQ_WAVE_CYCLES', 1e-6):6.1f} {g('SQ_BUSY_CYCLES', 1e-6):6.2f} {g('SQ_LDS_BANK_CONFLICT', 1e-6):5.2f} {valu_pct:5.1f} {wait_pct:5.1f}")


def traffic(outdir):
    """The two headline kernels in the format bench.py reads (keyed to the build id of the kernel sources)."""
    per, calib = load(outdir)
    sys.path.insert(0, ROOT)
    from proxtv_amd import build as _build
    M = N = 4096
    out = {"source": "tools/kernel_counters.py: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE (one pass each), DR 4096^2 lambda 0.1",
           "build_id": _build.build_id(), "unit": "bytes per launch",
           "calibration": {k: {"kernel": "calib_copy_kernel (8 B/lane, 1 GiB)", "factor": v} for k, v in calib.items()}, "kernels": {}}
    want = {"row sweep (DR_ROW)": ("sweep_chunk_kernel<3, false", 32 * M * N), "column sweep (DR_COL)": ("sweep_along_kernel<1, false", 16 * M * N)}
    for label, (pat, algo) in want.items():
        hits = [(n, d) for n, d in per.items() if pat in n and d.get("FETCH_SIZE") and d.get("WRITE_SIZE")]
        if not hits:
            continue
        n, d = max(hits, key=lambda kv: kv[1].get("calls", 0))
        rd = d["FETCH_SIZE"] * 1024.0 * calib["FETCH_SIZE"]
        wr = d["WRITE_SIZE"] * 1024.0 * calib["WRITE_SIZE"]
        out["kernels"][label] = {"kernel": short(n), "hbm_read": rd, "hbm_write": wr, "hbm_total": rd + wr, "algorithmic": algo,
                                 "ratio_to_algorithmic": (rd + wr) / algo, "dispatches": d.get("calls")}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "collect":
        collect(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else CASES)
    elif sys.argv[1] == "report":
        report(sys.argv[2])
    else:
        traffic(sys.argv[2])
