"""What the mid-solve samples of the sweep operands say (policy_reprobe; option verbose prints the certain fraction whenever it changes) next
to what each rung costs: PD2 / Yang / DR on 4096^2 unit noise over lambda.
    python tools/operand_stats.py [--methods pd,yang,dr] [--lams 0.4,0.5,...]"""
import argparse, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from proxtv_amd import _lib, device
    lib = _lib.require_device()
    method, lam, mode = sys.argv[2], float(sys.argv[3]), int(sys.argv[4])
    X = device.to_colmajor(torch.from_numpy(np.random.default_rng(0).standard_normal((4096, 4096))).cuda())
    out = device.colmajor_empty((4096, 4096))
    lib.proxtv_set_option(b"chunk_mode", mode)
    run = lambda: device.tv1_2d(X, lam, method=method, out=out)
    run()
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    print(f"RESULT {method} lambda={lam} mode {mode}: {best * 1e3:.2f} ms fixups {lib.proxtv_last_fixups()}", flush=True)
    if mode < 0:
        lib.proxtv_set_option(b"verbose", 1)
        run()
    sys.exit(0)
ap = argparse.ArgumentParser()
ap.add_argument("--methods", default="pd,yang")
ap.add_argument("--lams", default="0.3,0.4,0.5,0.6,0.7,0.8")
args = ap.parse_args()
for method in args.methods.split(","):
    for lam in args.lams.split(","):
        for mode in (-1, 1, 3):
            r = subprocess.run([sys.executable, __file__, "--child", method, lam, str(mode)], capture_output=True, text=True)
            for line in (r.stdout + r.stderr).splitlines():
                if line.startswith("RESULT") or "certain fraction" in line:
                    print(line.replace("[proxtv_amd] policy: ", "    "), flush=True)
