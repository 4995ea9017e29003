"""Per-workgroup phase timeline of the chunk kernels (option "trace"): where every workgroup of a sweep ran and when
its phases ended.  Answers: do the workgroups sharing a CU run in lockstep?  which phase is the long one?

    python tools/wg_trace.py [lambda [pinned mode]] > gpurun_out/wg_trace.txt
"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proxtv_amd import _lib, device

lam = float(sys.argv[1]) if len(sys.argv) > 1 else 0.1
lib = _lib.require_device()
if len(sys.argv) > 2:
    lib.proxtv_set_option(b"chunk_mode", int(sys.argv[2]))
x = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
y = device.colmajor_empty((4096, 4096))
device.tv1_2d(x, lam, out=y)


def report(title, rows=16):
    buf = np.zeros((32768, 8), dtype=np.uint64)
    n = lib.proxtv_debug_trace(buf.ctypes.data, 32768)
    buf = buf[:n]
    hw = buf[:, 0]
    hwid = (hw & np.uint64(0xffffffff)).astype(np.int64)
    xcc = (hw >> np.uint64(32)).astype(np.int64) & 0xf
    cu = (hwid >> 8) & 0xf; sh = (hwid >> 12) & 1; se = (hwid >> 13) & 7; tg = (hwid >> 16) & 0xf
    t = buf[:, 1:6].astype(np.int64)
    t = (t - t[:, 0].min()) * 0.01     # microseconds (100 MHz)
    print(f"## {title}: {n} workgroups; kernel span {t[:, 4].max():.1f} us")
    # how many records are in flight over the kernel's life (resident waves / workgroups), and per CU
    cuid0 = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    span = t[:, 4].max()
    grid = np.linspace(0.0, span, 41)
    alive = [(int(((t[:, 0] <= g) & (t[:, 4] > g)).sum())) for g in grid]
    print("# records in flight at 40 instants across the span:", " ".join(str(a) for a in alive))
    print(f"# mean in flight {np.mean((t[:, 4] - t[:, 0]).sum() / span):.0f} over {len(set(cuid0.tolist()))} CUs = "
          f"{(t[:, 4] - t[:, 0]).sum() / span / len(set(cuid0.tolist())):.2f} per CU; records per CU: min {np.bincount(cuid0).min()} max {np.bincount(cuid0)[np.bincount(cuid0) > 0].max()}")
    percu_end = np.array([t[cuid0 == c, 4].max() for c in sorted(set(cuid0.tolist()))])
    print(f"# last record of a CU ends at: min {percu_end.min():.1f} median {np.median(percu_end):.1f} max {percu_end.max():.1f} us; "
          f"first records start at {t[:, 0].min():.2f} .. {np.sort(t[:, 0])[min(n - 1, 4095)]:.2f} us (4096th)")
    if os.environ.get("WG_TRACE_DUMP"):
        np.savez_compressed(os.path.join(os.environ["WG_TRACE_DUMP"], title.split(",")[0].replace(" ", "_")[:40] + ".npz"), t=t, cuid=cuid0, xcc=xcc)
    print("#    wg xcc se sh cu tg | start staged walked rebuilt end (us)")
    cuid = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    order = np.lexsort((t[:, 0], cuid))
    for i in order[:rows]:
        print(f"{i:5d} {xcc[i]} {se[i]} {sh[i]} {cu[i]:2d} {tg[i]:2d} | " + " ".join(f"{v:7.2f}" for v in t[i]))
    ph = np.diff(t, axis=1)
    later = t[:, 0] > 1.0
    print("# distinct CUs:", len(set(cuid.tolist())), " tg values:", sorted(set(tg.tolist())))
    print("# mean phase lengths (us), all:          stage %.2f walk %.2f link+rebuild %.2f stream-out %.2f" % tuple(ph.mean(axis=0)))
    if later.any():
        print("# mean phase lengths (us), after round 1: stage %.2f walk %.2f link+rebuild %.2f stream-out %.2f" % tuple(ph[later].mean(axis=0)))
    print("# mean workgroup latency %.2f us; %d records (one per workgroup; the along-fibre kernel: one per wave)" % ((t[:, 4] - t[:, 0]).mean(), n))


lib.proxtv_set_option(b"trace", 1)
if os.environ.get("WG_TRACE_WEIGHTED") == "1":
    rng = np.random.default_rng(1)
    w1 = device.to_colmajor(torch.from_numpy(rng.uniform(0.05, 0.15, (4095, 4096))).cuda())
    w2 = device.to_colmajor(torch.from_numpy(rng.uniform(0.05, 0.15, (4096, 4095))).cuda())
    device.tv1w_2d(x, w1, w2, out=y)
    report("weighted row sweep, DRW_ROW_FINAL")
    device.tv1_fibres(x, 0.0, 1, weights=w2, out=y)
    report("weighted row sweep, OP_PROX")
    device.tv1_fibres(x, 0.0, 0, weights=w1, out=y)
    report("weighted column sweep (along the fibre), OP_PROX")
    sys.exit(0)
device.tv1_2d(x, lam, out=y)     # the last launch of a DR solve is the final row sweep (OP_DR_ROW_FINAL: two-operand input, epilogue fetches)
report("row sweep, DR_ROW_FINAL")
device.tv1_fibres(x, lam, 0, out=y)
report("column sweep (chunks along the fibre, one record per wave), OP_PROX")
device.tv1_fibres(x, lam, 1, out=y)
report("row sweep, OP_PROX")
lib.proxtv_set_option(b"trace", 0)
