#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json): 2-D TV-L1 Douglas-Rachford, 4096x4096 float64,
lambda = 0.1, 35 pinned iterations, on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" = one complete DR solve (DR2_TV semantics, through the C-ABI's device-pointer entry point) of one
synthetic 4096x4096 image that is already resident in HBM; the result stays in HBM.  With N > 1 (one process per
GPU, launched by torch.distributed.run) every rank solves its own independent image per step -- the path shards
over images with no collective in the data path ("scaling": "weak"); value = pixels all ranks processed / max-over-
ranks wall time.  Rank 0 prints ONE JSON line.

Extra objects on that line:
  roofline     -- the dominant sweep kernel: algorithmic HBM bytes per launch / average launch duration, measured
                  with hipEvents on the library's own stream during K further solves right after the timed ones (the
                  events cost ~0.8 ms per solve, so they stay out of the timed region; DESIGN.md "Measurement");
                  `by_kernel` = the same fraction for every sweep flavour (DR_COL, DR_ROW inside the solve; plain OP_PROX
                  sweeps in both directions measured on their own), `solve_frac` = the whole solve's algorithmic bytes
                  / ms_per_step / peak; `traffic` = PMC-measured HBM bytes per launch from profiles/, or null when that
                  file was measured on another build of the kernels (build id = hash of the kernel sources).
  step_ms      -- min / median / max of the K timed steps.
  c5           -- BASELINE config #5 per rank: 64 independent 2048x2048 images through proxtv_DR2_TV_batch_dev, solver
                  time and (N > 1) the RCCL gather to rank 0 timed separately.
  c3, c4, hard, lambda_1 -- the other BASELINE configurations and two neighbours of the headline, clocked by whoever runs this file
                  (rank 0, N = 1 only; outside the timed region): c3 = tv1w_2d weighted DR 4096^2; c4 = the 512x512x64 volume through
                  PD_TV (what tvgen runs) and Yang3; hard = SURVEY 8(d)'s back-tracking image (8 x 8 blocks + 0.2 N(0,1)) at 4096^2,
                  lambda = 0.5; lambda_1 = the headline image at lambda = 1.  Each: median ms of 3 solves, Mpixel/s, the solve-level
                  fraction of 8 TB/s from SURVEY 8(d)'s byte formulas, and an output check against the compiled reference's digest
                  in tests/golden/golden_large.npz.
  cpu_baseline -- the compiled reference (oracle/_ref, kind "reference") or, if it did not travel, this repo's
                  C restatement (kind "port"), timed on the host cores on a bounded sample (rank 0, N = 1 only): the headline
                  size at the best thread count of a probe, plus the 1024^2 probe at one thread and at all logical cores.
  end_to_end   -- the same solve through the drop-in host-pointer DR2_TV (transfers included); never `value`.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

M = N = 4096
LAM = 0.1
ITERS = 35
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); measured copy peak is ~6290 GB/s


def cpu_baseline():
    """Reference OpenMP CPU path on this box's host cores, bounded sample (one DR solve)."""
    from oracle import cpu
    cores = os.cpu_count() or 1
    if cpu.have_reference():
        lib, kind = cpu.reference(), "reference"
    else:
        lib, kind = cpu.oracle(), "port"
    # The reference's OpenMP scaling is poor (serial pointwise loops, strided gathers: SURVEY 6), and on a many-core
    # host "all cores" can be slower than a few.  Probe a few thread counts on a 1024^2 image (fractions of a second
    # each), then time the headline 4096^2 solve once with the best one -- bounded to roughly 10-30 s of CPU work.
    probe = np.asfortranarray(np.random.default_rng(1).standard_normal((1024, 1024)))
    cands = sorted({min(cores, c) for c in (8, 16, 32, 64, cores)})
    best_thr, best_t = cands[0], float("inf")
    for thr in cands:
        t0 = time.perf_counter()
        lib.dr2(probe, LAM, n_threads=thr)
        dt = time.perf_counter() - t0
        if dt < best_t:
            best_thr, best_t = thr, dt
    side = 4096 if best_t * 16 < 40 else 2048
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((side, side)))
    # median of three solves (a single ~5 s solve on a shared many-core host moved +-15 % from box to box)
    runs = []
    for _ in range(3):
        t0 = time.perf_counter()
        _, info, _ = lib.dr2(X, LAM, n_threads=best_thr)
        runs.append(time.perf_counter() - t0)
    dt = float(np.median(runs))
    # the same path at ONE thread and at ALL logical cores, on the bounded 1024^2 sample (a 4096^2 solve at one thread
    # alone takes ~40 s); Mpixel/s of that sample, so the three figures are comparable only through the stated sizes
    def sample_rate(thr):
        t0 = time.perf_counter()
        lib.dr2(probe, LAM, n_threads=thr)
        return 1024 * 1024 / (time.perf_counter() - t0) / 1e6
    one, everything = sample_rate(1), sample_rate(cores)
    return {"value": side * side / dt / 1e6, "unit": "Mpixel/s", "cores": best_thr, "kind": kind,
            "sample": f"one DR2_TV solve, {side}x{side} f64, lambda={LAM}, {int(info[0])} iterations, {dt:.2f} s with "
                      f"{best_thr} OpenMP threads (median of 3 solves: {', '.join(f'{r:.2f}' for r in runs)} s; thread count = best of {cands} "
                      f"probed on 1024x1024; host has {cores} logical cores)",
            "threads_1": {"value": one, "unit": "Mpixel/s", "cores": 1, "sample": "one DR2_TV solve, 1024x1024 f64, same lambda / iterations"},
            "threads_all": {"value": everything, "unit": "Mpixel/s", "cores": cores, "sample": "one DR2_TV solve, 1024x1024 f64, same lambda / iterations"},
            "best_on_sample": {"value": 1024 * 1024 / best_t / 1e6, "unit": "Mpixel/s", "cores": best_thr, "sample": "one DR2_TV solve, 1024x1024 f64"}}


def digest_check(y, g, key, tol=1e-6):
    """Compare a column-major device result with the fixture's digest of the compiled reference's output for the same input."""
    flat = y.permute(*reversed(range(y.dim()))).reshape(-1).cpu().numpy()   # storage order (the permuted view is contiguous)
    sub, ref = flat[::int(g["step"])], g[f"{key}/sub"]
    rel = float(np.max(np.abs(sub - ref)) / np.max(np.abs(ref)))
    rel_sum = float(abs(np.abs(flat).sum() - float(g[f"{key}/abs"])) / float(g[f"{key}/abs"]))
    return {"against": f"tests/golden/golden_large.npz {key} (compiled reference)", "rel_err": rel, "rel_err_abs_sum": rel_sum,
            "samples": int(sub.size), "tolerance": tol, "ok": bool(rel <= tol and rel_sum <= tol)}


def other_configs(device, torch, g):
    """BASELINE configs #3 / #4 and two neighbours of the headline: wall time of the device-resident solve (median of 3 after one
    warm-up; every solve returns synchronised), and the last result against the reference digest."""
    out = {}

    def dev(a):
        return device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())

    def clock(fn):
        fn()
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            res = fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)) * 1e3, res

    def entry(workload, ms, px, nbytes, y, info, key, want_iters):
        e = {"workload": workload, "ms": ms, "value": px / ms / 1e3, "unit": "Mpixel/s", "algorithmic_bytes": int(nbytes),
             "frac": nbytes / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "iterations": int(info[0])}
        try:
            e["output_check"] = digest_check(y, g, key)
            e["ok"] = bool(e["output_check"]["ok"] and int(info[0]) == want_iters)
        except KeyError as exc:
            e["output_check"] = {"error": f"no digest {exc} in the fixture"}
            e["ok"] = False
        return e

    px = M * N
    # config #3: weighted DR, per-pixel penalties ~ U(0.05, 0.15) -- inputs as tests/test_gpu_large.py::test_c3_weighted_dr_4096
    rng = np.random.default_rng(0)
    X = dev(rng.standard_normal((M, N)))
    W1, W2 = dev(rng.uniform(0.05, 0.15, (M - 1, N))), dev(rng.uniform(0.05, 0.15, (M, N - 1)))
    yo = device.colmajor_empty((M, N))
    ms, (y, info) = clock(lambda: device.tv1w_2d(X, W1, W2, out=yo))
    out["c3"] = entry("tv1w_2d (DR2L1W_TV, 35 iterations) 4096x4096 f64, per-pixel penalties U(0.05, 0.15)", ms, px,
                      8 * px * (8 * ITERS + 9), y, info, "c3/dr2w", ITERS)
    del W1, W2
    # the headline image at lambda = 1 (pieces of ~100 samples: the long-piece rungs)
    ms, (y, info) = clock(lambda: device.tv1_2d(X, 1.0, out=yo))
    out["lambda_1"] = entry("tv1_2d (DR2_TV, 35 iterations) on the headline image at lambda=1.0", ms, px, 8 * px * (6 * ITERS + 7),
                            y, info, "lam1/dr2", ITERS)
    # SURVEY 8(d)'s hard variant at the headline size: 8 x 8 random blocks + 0.2 N(0,1), lambda = 0.5
    r7 = np.random.default_rng(7)
    Xh = dev(np.kron(r7.standard_normal((8, 8)), np.ones((M // 8, N // 8))) + 0.2 * r7.standard_normal((M, N)))
    ms, (y, info) = clock(lambda: device.tv1_2d(Xh, 0.5, out=yo))
    out["hard"] = entry("tv1_2d (DR2_TV, 35 iterations) 4096x4096 f64: 8x8 random blocks of 512x512 + 0.2 N(0,1), lambda=0.5", ms, px,
                        8 * px * (6 * ITERS + 7), y, info, "hard4096/dr2", ITERS)
    del X, Xh, yo
    # config #4: 512 x 512 x 64 volume (float32 values up-cast, like the reference's Python surface), lambda = [0.1, 0.1, 0.05]
    V = dev(np.random.default_rng(0).standard_normal((512, 512, 64)).astype(np.float32).astype(np.float64))
    vo = device.colmajor_empty((512, 512, 64))
    nv = 512 * 512 * 64
    ms, (y, info) = clock(lambda: device.tvgen(V, [0.1, 0.1, 0.05], [1, 2, 3], out=vo))
    pd = entry("tvgen (PD_TV, 3 terms) 512x512x64 f64, lambda=[0.1, 0.1, 0.05]", ms, nv, 8 * nv * 17 * int(info[0]), y, info, "c4/pd", 35)
    ms, (y, info) = clock(lambda: device.tvgen(V, [0.1, 0.1, 0.1], [1, 2, 3], method="yang", out=vo))
    ya = entry("Yang3_TV (35 ADMM iterations) 512x512x64 f64, lambda=0.1", ms, nv, 8 * nv * 20 * (int(info[0]) - 1), y, info, "c4/yang3", 36)
    out["c4"] = {"pd_tv": pd, "yang3": ya, "ok": bool(pd["ok"] and ya["ok"])}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-others", action="store_true", help="skip the c3 / c4 / hard / lambda_1 objects")
    ap.add_argument("--no-c5", action="store_true", help="skip the config-#5 object (64 x 2048^2 per rank, ~13 GiB of HBM)")
    ap.add_argument("--no-gather", action="store_true", help="config #5 without the final gather to rank 0 (solver scaling and "
                    "gather time separable on a multi-GPU node)")
    ap.add_argument("--c5-images", type=int, default=64, help="images per rank of the config-#5 object (BASELINE: 64; the "
                    "shared-GPU plumbing test uses fewer)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs torch.distributed.run with --nproc-per-node {args.gpus}")
    # One rank per GPU over RCCL.  (PROXTV_BENCH_SHARED_GPU=1 is a plumbing dry run for boxes with a single GPU: every
    # rank uses device 0 and the two scalar collectives go over gloo -- never a measurement.)
    shared = os.environ.get("PROXTV_BENCH_SHARED_GPU") == "1"
    if shared and world > 1 and torch.cuda.device_count() >= world:
        raise SystemExit(f"PROXTV_BENCH_SHARED_GPU=1 with {torch.cuda.device_count()} devices visible for {world} ranks: the dry run is for "
                         "single-GPU boxes only -- unset it and every rank gets its own GPU over RCCL")
    torch.cuda.set_device(0 if shared else local)
    if world > 1:
        if shared:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))

    from proxtv_amd import _lib, device
    lib = _lib.require_device()

    # synthetic input of the BASELINE shape, a different image per rank, resident in HBM before timing starts
    x_host = np.random.default_rng(rank).standard_normal((M, N))
    xd = device.to_colmajor(torch.from_numpy(x_host).cuda())
    yd = device.colmajor_empty((M, N))

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        device.tv1_2d(xd, LAM, out=yd)

    # ---- the timed region: exactly K solves, nothing but the product path -------------------------------------------
    # (every solve returns synchronised, so the per-step stamps cost nothing)
    step_t = []
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ts = time.perf_counter()
        _, info = device.tv1_2d(xd, LAM, out=yd)
        step_t.append(time.perf_counter() - ts)
    barrier()
    dt = time.perf_counter() - t0
    assert int(info[0]) == ITERS, info

    # ---- did the timed solves do the work?  Rank 0's input is default_rng(0).standard_normal((4096, 4096)) -- exactly the `c2` fixture
    # of tests/golden/golden_large.npz, a digest (strided subsample + moments) of what the compiled reference returns for it -- so the
    # output of the LAST timed solve is compared with that digest here, outside the timed region.  (Data only: nothing under oracle/.)
    output_check = None
    if rank == 0:
        try:
            g = np.load(os.path.join(ROOT, "tests", "golden", "golden_large.npz"))
            flat = yd.permute(1, 0).reshape(-1).cpu().numpy()   # column-major storage order (the permuted view is contiguous)
            sub = flat[::int(g["step"])]
            ref = g["c2/dr2/sub"]
            rel = float(np.max(np.abs(sub - ref)) / np.max(np.abs(ref)))
            rel_sum = float(abs(np.abs(flat).sum() - float(g["c2/dr2/abs"])) / float(g["c2/dr2/abs"]))
            output_check = {"against": "tests/golden/golden_large.npz c2/dr2 (compiled reference, 4096x4096, lambda=0.1)",
                            "rel_err": rel, "rel_err_abs_sum": rel_sum, "samples": int(sub.size), "tolerance": 1e-6,
                            "ok": bool(rel <= 1e-6 and rel_sum <= 1e-6)}
        except (OSError, KeyError) as exc:
            output_check = {"error": str(exc)}

    # ---- end to end through the drop-in host-pointer entry point (what an unmodified caller of the reference gets): H2D + solve +
    # D2H from / to the caller's pageable numpy arrays.  Reported beside the device-resident rate, never as `value`.
    x_f = np.asfortranarray(x_host)
    y_f = np.zeros_like(x_f, order="F")
    info_h = np.zeros(3)
    e2e = []
    for k in range(4):
        ts = time.perf_counter()
        lib.DR2_TV(M, N, x_f.ctypes.data, LAM, LAM, 1.0, 1.0, y_f.ctypes.data, 1, 0, info_h.ctypes.data)
        if k: e2e.append(time.perf_counter() - ts)
    end_to_end_ms = float(np.median(e2e)) * 1e3
    del x_f, y_f

    # ---- the same K solves again with a hipEvent pair around every sweep (library option "profile") -------------------
    # The events go on the library's own stream; each record drains the queue, which costs ~6 us per event = ~0.8 ms per
    # solve, so the instrumented pass is kept out of the timed region and its own wall time is reported beside it.
    lib.proxtv_set_option(b"profile", 1)
    fam_ms = [0.0, 0.0, 0.0]
    fam_n = [0, 0, 0]
    barrier()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        device.tv1_2d(xd, LAM, out=yd)
        for f in range(3):
            fam_ms[f] += lib.proxtv_last_kernel_ms(f)
            fam_n[f] += lib.proxtv_last_kernel_launches(f)
    barrier()
    dt_events = time.perf_counter() - t1
    # plain 1-D prox sweeps (OP_PROX: what PD / Yang / batched tv1_1d launch) in both directions, the same way
    prox_ms = {}
    for dim, name in ((0, "prox_col"), (1, "prox_row")):
        tot, cnt = 0.0, 0
        for _ in range(max(2, args.steps)):
            device.tv1_fibres(xd, LAM, dim, out=yd)
            tot += lib.proxtv_last_kernel_ms(dim)
            cnt += lib.proxtv_last_kernel_launches(dim)
        prox_ms[name] = tot / max(cnt, 1)
    lib.proxtv_set_option(b"profile", 0)

    # ---- BASELINE config #5: 64 independent 2048^2 images per rank, then the gather --------------------------------------
    c5 = None
    if not args.no_c5:
        B5, S5 = args.c5_images, 2048
        g5 = torch.Generator(device="cuda").manual_seed(1000 + rank)
        x5 = torch.randn((B5, S5, S5), dtype=torch.float64, device="cuda", generator=g5).permute(2, 1, 0)   # column-major (M, N, B)
        y5 = device.colmajor_empty((S5, S5, B5))
        device.tv1_2d_batch(x5, LAM, out=y5)
        barrier()
        ts = time.perf_counter()
        device.tv1_2d_batch(x5, LAM, out=y5)
        barrier()
        t_solve = time.perf_counter() - ts
        t_gather = None
        hbm_peak_bytes = None
        if world > 1 and not args.no_gather:
            send = y5.permute(2, 1, 0)                      # (B, N, M) contiguous view of the same bytes
            if shared:
                send = send.cpu()                           # plumbing dry run: the same gather over gloo on host copies
            # rank 0 receives into ONE pre-allocated (world, B, N, M) tensor; the gather list is its rows (views, no copies)
            recv = torch.empty((world,) + tuple(send.shape), dtype=send.dtype, device=send.device) if rank == 0 else None
            bufs = list(recv.unbind(0)) if rank == 0 else None
            barrier()
            ts = time.perf_counter()
            dist.gather(send, gather_list=bufs, dst=0)
            barrier()
            t_gather = time.perf_counter() - ts
            # did every rank's block arrive intact?  (checksum of each rank's first image, sent separately)
            sums = [None] * world
            dist.all_gather_object(sums, float(send[0].double().sum()))
            gather_ok = None
            if rank == 0:
                gather_ok = all(abs(float(recv[r, 0].double().sum()) - sums[r]) <= 1e-9 * max(1.0, abs(sums[r])) for r in range(world))
            hbm_peak_bytes = int(torch.cuda.max_memory_allocated())   # torch's own allocations (operands + receive buffer), this rank
            del bufs, recv
        if world > 1:
            tt = torch.tensor([t_solve], dtype=torch.float64, device="cpu" if shared else "cuda")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            t_solve = float(tt.item())
        c5 = {"workload": f"{B5} independent {S5}x{S5} f64 images per rank, tv1_2d DR (35 iterations), lambda={LAM}, one batched solve",
              "images": B5 * world, "solve_ms": t_solve * 1e3, "value": world * B5 * S5 * S5 / t_solve / 1e6, "unit": "Mpixel/s",
              "gather_ms": None if t_gather is None else t_gather * 1e3,
              "gather_bytes": None if t_gather is None else (world - 1) * B5 * S5 * S5 * 8,
              "gather_checked": None if t_gather is None else gather_ok,
              "gather": ("skipped (--no-gather)" if (world > 1 and args.no_gather) else None if world == 1 else
                         "one dist.gather of every rank's (B, N, M) block into one pre-allocated (world, B, N, M) tensor on rank 0"),
              "rank0_torch_hbm_peak_bytes": hbm_peak_bytes,
              "ranks": world, "backend": (dist.get_backend() if world > 1 else None)}
        del x5, y5

    # ---- the other configurations, clocked here so that whoever runs this file clocks them (N = 1 only) -----------------------------
    others = None
    if world == 1 and not args.no_others:
        del xd, yd
        others = other_configs(device, torch, np.load(os.path.join(ROOT, "tests", "golden", "golden_large.npz")))

    # who took part: what the process group itself reports, and every rank's device (rank 0 prints them)
    ranks_info = [{"rank": 0, "device": torch.cuda.get_device_name(torch.cuda.current_device()), "index": torch.cuda.current_device()}]
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if shared else "cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        mine = {"rank": rank, "device": torch.cuda.get_device_name(torch.cuda.current_device()), "index": torch.cuda.current_device()}
        ranks_info = [None] * world
        dist.all_gather_object(ranks_info, mine)
        assert dist.get_world_size() == world

    if rank == 0:
        ms_per_step = dt / args.steps * 1e3
        value = world * M * N * args.steps / dt / 1e6
        # dominant kernel family and its algorithmic traffic per launch: column sweep R t, W s' = 16 B/px;
        # fused row sweep R s', R U, R t, W t = 32 B/px
        per_px = {0: 16, 1: 32}
        dom = 0 if fam_ms[0] >= fam_ms[1] else 1
        avg_ms = fam_ms[dom] / max(fam_n[dom], 1)
        achieved = per_px[dom] * M * N / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0

        def frac(bytes_per_px, ms):
            return (bytes_per_px * M * N / (ms * 1e-3) / 1e9) / HBM_PEAK_GBS if ms > 0 else None
        by_kernel = {
            "DR_COL": {"bytes_per_px": 16, "avg_launch_ms": fam_ms[0] / max(fam_n[0], 1), "frac": frac(16, fam_ms[0] / max(fam_n[0], 1))},
            "DR_ROW": {"bytes_per_px": 32, "avg_launch_ms": fam_ms[1] / max(fam_n[1], 1), "frac": frac(32, fam_ms[1] / max(fam_n[1], 1))},
            "PROX_COL": {"bytes_per_px": 16, "avg_launch_ms": prox_ms["prox_col"], "frac": frac(16, prox_ms["prox_col"])},
            "PROX_ROW": {"bytes_per_px": 16, "avg_launch_ms": prox_ms["prox_row"], "frac": frac(16, prox_ms["prox_row"])},
        }
        solve_bytes = 8 * M * N * (6 * ITERS + 7)          # 1736 B/pixel at 35 iterations
        # HBM bytes per launch from the PMC run of this same command (tools/pmc_traffic.py: FETCH_SIZE + WRITE_SIZE,
        # separate passes, calibrated on an 8-B/lane copy of known size).  bench.py cannot run under the profiler
        # itself, so the figure is read from profiles/ -- and only if it was measured on THIS build of the kernels.
        traffic = None
        traffic_source = None
        try:
            from proxtv_amd import build as _build
            import glob
            pmc = {}
            for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):   # newest round first
                with open(path) as f:
                    cand = json.load(f)
                if cand.get("build_id") == _build.build_id():
                    pmc = cand
                    break
            if pmc:
                label = ["column sweep (DR_COL)", "row sweep (DR_ROW)"][dom]
                traffic = pmc["kernels"][label]["hbm_total"]
                traffic_source = (f"{os.path.relpath(path, ROOT)}: rocprofv3 --pmc passes of this workload run by the builder on build "
                                  f"{pmc.get('build_id')} (= this build); not measured by this run -- bench.py cannot profile itself")
        except (OSError, KeyError, ValueError):
            pass
        line = {
            "metric": "Mpixel/s on 2D TV-L1 DR (4096x4096 f64, lambda=0.1); % HBM roofline",
            "value": value, "unit": "Mpixel/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f64", "data": "synthetic",
            "config": {"workload": "tv1_2d (DR2_TV, 35 iterations) on one 4096x4096 float64 N(0,1) image per GPU per step, "
                                   "lambda=0.1, input and output resident in HBM", "images_per_step": world,
                       "parallelism": f"independent images, {world} rank(s), no data-path collective",
                       "process_group": {"world_size": dist.get_world_size() if world > 1 else 1,
                                         "backend": dist.get_backend() if world > 1 else None, "ranks": ranks_info}},
            "roofline": {"bound": "hbm", "kernel": ["column sweep (DR_COL)", "row sweep (DR_ROW, fused reflections+combiner)"][dom],
                         "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": traffic, "traffic_source": traffic_source, "avg_launch_ms": avg_ms, "launches": fam_n[dom],
                         "measured": f"hipEvents around every sweep launch during {args.steps} further solves of the same input "
                                     f"({dt_events / args.steps * 1e3:.2f} ms per solve with the events in the stream; every event pair serialises its "
                                     f"launch, tail included, so family_ms_per_solve sums to slightly MORE than ms_per_step of the un-instrumented solves)",
                         "algorithmic_bytes_per_launch": per_px[dom] * M * N,
                         "family_ms_per_solve": {"col": fam_ms[0] / args.steps, "row": fam_ms[1] / args.steps,
                                                 "other": fam_ms[2] / args.steps},
                         "by_kernel": by_kernel,
                         "solve_frac": (solve_bytes / (ms_per_step * 1e-3) / 1e9) / HBM_PEAK_GBS,
                         "solve_algorithmic_bytes": solve_bytes},
            "output_check": output_check,
            "step_ms": {"min": min(step_t) * 1e3, "median": float(np.median(step_t)) * 1e3, "max": max(step_t) * 1e3},
            "end_to_end_ms": end_to_end_ms,
            "end_to_end": {"ms": end_to_end_ms, "value": M * N / end_to_end_ms / 1e3, "unit": "Mpixel/s",
                           "what": "DR2_TV through the host-pointer C-ABI: pageable numpy in, numpy out (2 x 134 MB over the link + the solve)"},
        }
        if c5 is not None:
            line["c5"] = c5
        if others:
            line.update(others)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)

    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
