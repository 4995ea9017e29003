"""Does the default policy take a good kernel on everything, or only on the headline's kind of data?  Every case runs under the default policy
and pinned to rungs 1 and 3 (each in its own process: a pinned rung on the wrong data can take seconds); a case whose default time is more than
1.25 x the best pinned time is flagged.
    python tools/robustness.py [--only substring] [--n 4096]"""
import argparse, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def images(n):
    import numpy as np
    r = np.random.default_rng(0)
    noise = r.standard_normal((n, n))
    k = np.fft.fftfreq(n)
    lowpass = np.exp(-(k[:, None] ** 2 + k[None, :] ** 2) * (n / 24.0) ** 2)
    smooth = np.real(np.fft.ifft2(np.fft.fft2(r.standard_normal((n, n))) * lowpass))
    smooth *= 1.0 / smooth.std()
    r7 = np.random.default_rng(7)
    blocks = np.kron(r7.standard_normal((8, 8)), np.ones((n // 8, n // 8)))
    spikes = np.zeros((n, n))
    idx = r.integers(0, n, (2000, 2))
    spikes[idx[:, 0], idx[:, 1]] = 5.0 * r.standard_normal(2000)
    half = noise.copy()
    half[:, n // 2:] = 0.25
    return {"noise": noise, "photo": smooth + 0.1 * r.standard_normal((n, n)), "blocks": blocks + 0.2 * r7.standard_normal((n, n)),
            "spikes": spikes + 0.01 * r.standard_normal((n, n)), "half": half}


CASES = [("dr", im, lam) for im, lams in (("noise", (0.1, 0.3, 0.5, 0.7, 1.0, 2.0, 5.0)), ("photo", (0.02, 0.1, 0.5, 2.0)), ("blocks", (0.1, 0.5, 2.0)),
                                          ("spikes", (0.05, 0.5)), ("half", (0.1, 1.0))) for lam in lams]
CASES += [("pd", "noise", lam) for lam in (0.1, 0.5, 1.0, 3.0)] + [("pd", "photo", lam) for lam in (0.1, 0.5)] + [("pd", "blocks", 0.5)]
CASES += [("yang", "noise", lam) for lam in (0.1, 1.0, 3.0, 10.0)] + [("yang", "photo", lam) for lam in (0.1, 1.0)] + [("yang", "blocks", 0.5)]
CASES += [("drw", "noise", sc) for sc in (0.1, 0.5, 1.0, 3.0)] + [("drw", "blocks", 0.5)]
CASES += [("pd3", "vol", lam) for lam in (0.1, 1.0)] + [("yang3", "vol", lam) for lam in (0.1, 1.0, 3.0)]
CASES += [("batch", "noise", lam) for lam in (0.1, 1.0)] + [("1d", "signal", lam) for lam in (0.1, 1.0, 10.0)]


def child(mode, n, only):
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    from proxtv_amd import _lib, device
    lib = _lib.require_device()
    lib.proxtv_set_option(b"chunk_mode", mode)
    dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    ims = None
    for method, im, lam in CASES:
        name = f"{method}/{im}/{lam}"
        if only and only not in name:
            continue
        if im in ("noise", "photo", "blocks", "spikes", "half") and ims is None:
            ims = {k: dev(v) for k, v in images(n).items()}
        r = np.random.default_rng(3)
        if method in ("dr", "pd", "yang"):
            X, out = ims[im], device.colmajor_empty((n, n))
            run = lambda: device.tv1_2d(X, lam, method=method, out=out)
        elif method == "drw":
            X, out = ims[im], device.colmajor_empty((n, n))
            W1, W2 = dev(r.uniform(0.5, 1.5, (n - 1, n)) * lam), dev(r.uniform(0.5, 1.5, (n, n - 1)) * lam)
            run = lambda: device.tv1w_2d(X, W1, W2, out=out)
        elif method in ("pd3", "yang3"):
            V, out = dev(np.random.default_rng(0).standard_normal((512, 512, 64))), device.colmajor_empty((512, 512, 64))
            run = lambda: device.tvgen(V, [lam, lam, lam / 2], [1, 2, 3], method="yang" if method == "yang3" else None, out=out)
        elif method == "batch":
            m = n // 2
            Bx = torch.empty_strided((m, m, 8), (1, m, m * m), dtype=torch.float64, device="cuda")
            for b in range(8):
                Bx[:, :, b] = ims[im][(b % 2) * m:(b % 2 + 1) * m, (b // 2 % 2) * m:(b // 2 % 2 + 1) * m]
            bout = torch.empty_strided(Bx.shape, Bx.stride(), dtype=torch.float64, device="cuda")
            run = lambda: device.tv1_2d_batch(Bx, lam, out=bout)
        else:
            sig = torch.from_numpy(np.cumsum(r.standard_normal(1 << 22)) * 0.05 + r.standard_normal(1 << 22)).cuda()
            sout = torch.empty_like(sig)
            run = lambda: device.tv1_fibres(sig.reshape(-1, 1), lam, 0, out=sout.reshape(-1, 1))
        try:
            run()
            best = 1e9
            for _ in range(2):
                torch.cuda.synchronize(); t0 = time.perf_counter(); run(); torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)
                if best > 1.0:
                    break
            print(json.dumps({"case": name, "mode": mode, "ms": best * 1e3, "ran": lib.proxtv_chunk_mode(), "fixups": lib.proxtv_last_fixups()}), flush=True)
        except Exception as e:   # noqa: BLE001 -- a case a surface does not take is a line in the table, not the end of it
            print(json.dumps({"case": name, "mode": mode, "error": str(e)[:120]}), flush=True)


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--child", type=int, default=None)
    ap.add_argument("--only", default="")
    ap.add_argument("--n", type=int, default=4096)
    args = ap.parse_args()
    if args.child is not None:
        child(args.child, args.n, args.only)
        sys.exit(0)
    table = {}
    for mode in (-1, 1, 3):
        r = subprocess.run([sys.executable, __file__, "--child", str(mode), "--n", str(args.n), "--only", args.only], capture_output=True, text=True)
        for line in r.stdout.splitlines():
            if line.startswith("{"):
                d = json.loads(line)
                table.setdefault(d["case"], {})[mode] = d
        if r.returncode:
            print(r.stderr[-2000:])
    print(f"{'case':28s} {'default ms':>11s} {'ran':>4s} {'fixups':>8s} {'rung 1 ms':>11s} {'rung 3 ms':>11s}  default / best pinned")
    for case, row in table.items():
        ms = lambda m: row.get(m, {}).get("ms")
        pinned = [v for v in (ms(1), ms(3)) if v]
        d = ms(-1)
        ratio = d / min(pinned) if d and pinned else float("nan")
        fmt = lambda v: f"{v:11.2f}" if v else f"{'--':>11s}"
        print(f"{case:28s} {fmt(d)} {row.get(-1, {}).get('ran', ''):>4} {row.get(-1, {}).get('fixups', ''):>8} {fmt(ms(1))} {fmt(ms(3))}  {ratio:5.2f}"
              + ("   <-- slower than a pinned rung by a quarter" if ratio > 1.25 else ""))
