#!/usr/bin/env python3
"""HBM traffic of the sweep kernels from rocprofv3 PMC counters, calibrated on a known copy.

Run on the GPU box (one --pmc pass per counter, no trace domains besides --kernel-trace):

    python tools/pmc_traffic.py collect gpurun_out/pmc      # runs rocprofv3 four times
    python tools/pmc_traffic.py report  gpurun_out/pmc > profiles/r02_pmc_traffic.json

Calibration (MI355X_MICROARCH.md, "HBM"): FETCH_SIZE / WRITE_SIZE come from the L2's fabric request counters and, on
gfx950, FETCH_SIZE under-reports wide coalesced reads; other widths are uncalibrated.  The sweep kernels move 8 bytes
per lane, so the factor is measured here with `calib_copy_kernel` (same width) on 1 GiB -- four times the Infinity
Cache -- whose true traffic is 1 GiB read + 1 GiB written.
"""
import json
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CALIB_N = 128 * 1024 * 1024          # doubles = 1 GiB


def calib_main():
    sys.path.insert(0, ROOT)
    import torch
    from proxtv_amd import _lib
    lib = _lib.require_device()
    src = torch.rand(CALIB_N, dtype=torch.float64, device="cuda")
    dst = torch.empty_like(src)
    torch.cuda.synchronize()
    for _ in range(3):
        lib.proxtv_calib_copy_dev(src.data_ptr(), dst.data_ptr(), CALIB_N, None)
    torch.cuda.synchronize()


def run(cmd):
    print("+", " ".join(cmd), flush=True)
    subprocess.run(cmd, check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


def collect(outdir):
    os.makedirs(outdir, exist_ok=True)
    env_cmd = ["rocprofv3", "--kernel-trace"]
    me = os.path.abspath(__file__)
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        run(env_cmd + ["--pmc", counter, "-d", os.path.join(outdir, f"calib_{counter}"), "-o", "c", "--",
                       sys.executable, me, "calib"])
        run(env_cmd + ["--pmc", counter, "-d", os.path.join(outdir, f"bench_{counter}"), "-o", "b", "--",
                       sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--no-c5"])


def averages(db, counter):
    c = sqlite3.connect(db)
    rows = c.execute("select kernel_name, sum(value), count(*) from counters_collection where counter_name = ? "
                     "group by kernel_name", (counter,)).fetchall()
    return {k: (v / n, n) for k, v, n in rows}


def report(outdir):
    M = N = 4096
    sys.path.insert(0, ROOT)
    from proxtv_amd import build as _build
    out = {"source": "rocprofv3 --kernel-trace --pmc <counter> (one pass per counter), python bench.py --steps 1 --warmup 1",
           "build_id": _build.build_id(),   # hash of the kernel sources: bench.py quotes these numbers for this build only
           "unit": "bytes per launch", "calibration": {}, "kernels": {}}
    factor = {}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        av = averages(os.path.join(outdir, f"calib_{counter}", "c_results.db"), counter)
        (name, (val, n)), = [(k, v) for k, v in av.items() if "calib_copy_kernel" in k]
        true_bytes = CALIB_N * 8
        reported = val * 1024.0          # the counters are in KiB
        factor[counter] = true_bytes / reported
        out["calibration"][counter] = {"kernel": "calib_copy_kernel (8 B/lane, 1 GiB)", "true_bytes": true_bytes,
                                       "reported_bytes": reported, "factor": factor[counter], "dispatches": n}
    # (DR_COL: the along-fibre kernel of dimension-0 sweeps, or the transposed tile when that is switched off)
    names = {"row sweep (DR_ROW)": "sweep_chunk_kernel<3,", "column sweep (DR_COL)": "sweep_along_kernel<1,"}
    algo = {"row sweep (DR_ROW)": {"read": 24 * M * N, "write": 8 * M * N},
            "column sweep (DR_COL)": {"read": 8 * M * N, "write": 8 * M * N}}
    per = {k: {} for k in names}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        av = averages(os.path.join(outdir, f"bench_{counter}", "b_results.db"), counter)
        for label, pat in names.items():
            hit = [(k, v) for k, v in av.items() if pat.replace(",", ", ") in k or pat in k]
            if hit:
                val, n = hit[0][1]
                per[label][counter] = {"reported_bytes": val * 1024.0, "corrected_bytes": val * 1024.0 * factor[counter],
                                       "dispatches": n}
    for label in names:
        if "FETCH_SIZE" in per[label] and "WRITE_SIZE" in per[label]:
            rd, wr = per[label]["FETCH_SIZE"]["corrected_bytes"], per[label]["WRITE_SIZE"]["corrected_bytes"]
            out["kernels"][label] = {"hbm_read": rd, "hbm_write": wr, "hbm_total": rd + wr,
                                     "algorithmic": algo[label]["read"] + algo[label]["write"],
                                     "ratio_to_algorithmic": (rd + wr) / (algo[label]["read"] + algo[label]["write"]),
                                     "raw": per[label]}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    if sys.argv[1] == "calib":
        calib_main()
    elif sys.argv[1] == "collect":
        collect(sys.argv[2])
    else:
        report(sys.argv[2])
