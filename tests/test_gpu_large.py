"""BASELINE.json configurations at full size on the GPU, checked against fingerprints of the compiled reference's
outputs (tests/golden/golden_large.npz: strided subsample + moments, made by oracle/gen_golden.py --large) and
through size-independent properties of the prox.  Inputs are re-created from the same seeds and verified against
the fixture's input fingerprint first."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


def _check_digest(y, g, key, tol=1e-6):
    flat = np.asarray(y).ravel(order="F")
    step = int(g["step"]) if not key.startswith("hard") else 1031
    assert_close(flat[::step], g[f"{key}/sub"], tol=tol, what=f"{key} subsample")
    scale = float(g[f"{key}/abs"])
    assert abs(flat.sum() - float(g[f"{key}/sum"])) <= tol * scale, key
    assert abs(np.abs(flat).sum() - scale) <= tol * scale, key
    assert abs((flat * flat).sum() - float(g[f"{key}/sq"])) <= tol * float(g[f"{key}/sq"]), key


def _tv_aniso(y):
    return sum(np.abs(np.diff(y, axis=a)).sum() for a in range(y.ndim))


def test_c2_dr_4096(ptv, glarge):
    """Config #2 (north-star): tv1_2d DR on one 4096x4096 float64 image, lambda = 0.1, 35 pinned iterations."""
    X = np.asfortranarray(np.random.default_rng(0).standard_normal((4096, 4096)))
    np.testing.assert_array_equal(X.ravel(order="F")[::int(glarge["step"])], glarge["c2/X/sub"])
    y = ptv.tv1_2d(X, 0.1)
    _check_digest(y, glarge, "c2/dr2")
    assert glarge["c2/dr2_info"][0] == 35
    # properties: mean preserved by every prox sweep; TV decreases; the prox moves each pixel by at most 4*lambda
    assert abs(y.mean() - X.mean()) < 1e-9
    assert _tv_aniso(y) < _tv_aniso(X)
    assert np.max(np.abs(y - X)) <= 4 * 0.1 + 1e-6


def test_c3_weighted_dr_4096(ptv, glarge):
    """Config #3: tv1w_2d on 4096x4096 with per-pixel weights ~ U(0.05, 0.15)."""
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.standard_normal((4096, 4096)))
    W1 = np.asfortranarray(rng.uniform(0.05, 0.15, (4095, 4096)))
    W2 = np.asfortranarray(rng.uniform(0.05, 0.15, (4096, 4095)))
    np.testing.assert_array_equal(W2.ravel(order="F")[::int(glarge["step"])], glarge["c3/W2/sub"])
    y = ptv.tv1w_2d(X, W1, W2)
    _check_digest(y, glarge, "c3/dr2w")
    assert abs(y.mean() - X.mean()) < 1e-9


def test_c3_weighted_dr_4096_global_memory_walks(ptv, clib, glarge):
    """The same solve through the geometries that walk global memory (chunks with 256-sample zones, one sequential walk
    per fibre).  Exactly-sized device buffers: an index one past the last fibre's weights is a GPU memory fault, not a
    silently wrong read -- the regression this test pins (the pipelined walker fetched r[n - 1] of the last fibre)."""
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.standard_normal((4096, 4096)))
    W1 = np.asfortranarray(rng.uniform(0.05, 0.15, (4095, 4096)))
    W2 = np.asfortranarray(rng.uniform(0.05, 0.15, (4096, 4095)))
    before = clib.proxtv_set_option(b"chunk_mode", 3)
    try:
        for mode in (3, 5):
            clib.proxtv_set_option(b"chunk_mode", mode)
            _check_digest(ptv.tv1w_2d(X, W1, W2), glarge, "c3/dr2w")
    finally:
        clib.proxtv_set_option(b"chunk_mode", before)


def test_c4_volume(ptv, clib, glarge):
    """Config #4: 512x512x64 volume (float32 up-cast like the reference surface does), lambda = [0.1, 0.1, 0.05].
    tvgen runs PD_TV (35 iterations, RC_ITERS) -- plus scalar-lambda Yang3_TV, the C entry point BASELINE names."""
    V32 = np.random.default_rng(0).standard_normal((512, 512, 64)).astype(np.float32)
    y = ptv.tvgen(V32, [0.1, 0.1, 0.05], [1, 2, 3], [1, 1, 1])
    assert y.dtype == np.float64 and y.flags.f_contiguous
    _check_digest(y, glarge, "c4/pd")
    want = glarge["c4/pd_info"]
    assert want[0] == 35 and want[2] == 1
    V = np.asfortranarray(V32.astype(np.float64))
    out, info = np.zeros(V.shape, order="F"), np.zeros(3)
    rc = clib.Yang3_TV(512, 512, 64, V.ctypes.data, 0.1, out.ctypes.data, 0, info.ctypes.data)
    assert rc == 1 and info[0] == glarge["c4/yang3_info"][0] == 36
    _check_digest(out, glarge, "c4/yang3")


def test_c5_batch_items(ptv, glarge):
    """Config #5 (per-GPU shard shape): independent 2048x2048 images solved as one batch == reference DR2_TV per image."""
    xs = np.stack([np.random.default_rng(k).standard_normal((2048, 2048)) for k in range(3)])
    ys = ptv.tv1_2d_batch(xs, 0.1)
    for k in range(3):
        _check_digest(np.asfortranarray(ys[k]), glarge, f"c5/{k}/dr2")


def test_hard_blocks_image(ptv, glarge):
    """Back-tracking-heavy variant (8x8 random blocks + noise, lambda = 0.5): the exactness-under-rewinds case."""
    rng = np.random.default_rng(7)
    Xh = np.asfortranarray(np.kron(rng.standard_normal((8, 8)), np.ones((128, 128))) + 0.2 * rng.standard_normal((1024, 1024)))
    np.testing.assert_array_equal(Xh.ravel(order="F")[::1031], glarge["hard/X/sub"])
    _check_digest(ptv.tv1_2d(Xh, 0.5), glarge, "hard/dr2")


def test_sweep_kernel_properties_full_size(ptv):
    """Properties of the per-sweep kernel at 4096 fibres x 4096 samples, both orientations (no oracle needed):
    per-fibre mean preserved, lambda = 0 is the identity, huge lambda gives the fibre mean, idempotence."""
    torch = pytest.importorskip("torch")
    from proxtv_amd import device
    g = torch.Generator(device="cpu").manual_seed(1)
    X = torch.randn((4096, 4096), generator=g, dtype=torch.float64)
    xd = device.to_colmajor(X.cuda())
    for dim in (0, 1):
        y = device.tv1_fibres(xd, 0.1, dim)
        assert torch.allclose(y.mean(dim=dim), xd.mean(dim=dim), atol=1e-12)
        assert torch.equal(device.tv1_fibres(xd, 0.0, dim), xd)
        big = device.tv1_fibres(xd, 1e6, dim)
        assert torch.allclose(big, xd.mean(dim=dim, keepdim=True).expand_as(xd), atol=1e-9)
        # the fibre TV never grows and the dual certificate holds: |cumsum(x - y)| <= lambda along the fibre
        u = torch.cumsum(xd - y, dim=dim)
        assert float(u.abs().max()) <= 0.1 * (1 + 1e-9)


def test_c5_full_shard_one_batched_call(glarge):
    """Config #5 at the per-GPU shard size: B = 64 independent 2048x2048 images in ONE batched solve.  The three images the
    fixture holds reference fingerprints for sit at scattered batch positions; ten sampled images are also solved alone
    through the single-image entry point and must match the batched result bit for bit (batch-invariant kernels and
    reductions, and a geometry choice that is a function of the input's statistics)."""
    import torch
    from proxtv_amd import device
    B, S = 64, 2048
    place = {0: 5, 1: 40, 2: 63}                      # fixture image k -> batch position
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.randn((B, S, S), dtype=torch.float64, device="cuda", generator=g)        # (B, N, M): image b column-major
    for k, pos in place.items():
        img = np.random.default_rng(k).standard_normal((S, S))
        x[pos] = torch.from_numpy(np.ascontiguousarray(img.T)).cuda()
    xb = x.permute(2, 1, 0)                            # column-major (M, N, B) view of the same bytes
    yb, info = device.tv1_2d_batch(xb, 0.1)
    assert int(info[0]) == 35
    for k, pos in place.items():
        _check_digest(np.asfortranarray(yb[:, :, pos].cpu().numpy()), glarge, f"c5/{k}/dr2")
    for pos in (0, 5, 9, 17, 26, 31, 40, 48, 55, 63):
        one, _ = device.tv1_2d(device.to_colmajor(xb[:, :, pos].contiguous()), 0.1)
        assert torch.equal(one, yb[:, :, pos]), f"image {pos}: batched and single-image results differ"
    # size-independent properties over the whole shard: per-image mean preserved, each pixel moved by at most 4 lambda
    assert float((yb.mean(dim=(0, 1)) - xb.mean(dim=(0, 1))).abs().max()) < 1e-9
    assert float((yb - xb).abs().max()) <= 0.4 + 1e-6


def test_bench_two_ranks_on_one_gpu_dry_run():
    """bench.py's N > 1 path end to end on a one-GPU box: torch.distributed.run, two ranks sharing device 0
    (PROXTV_BENCH_SHARED_GPU=1: gloo instead of RCCL -- plumbing, not a measurement).  Exercises the rank environment, the
    barriers, the max-over-ranks all-reduce, the config-#5 object with its gather (checked on arrival) and the JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PROXTV_BENCH_SHARED_GPU="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--c5-images", "4"]
    out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["unit"] == "Mpixel/s"
    assert d["config"]["images_per_step"] == 2
    pg = d["config"]["process_group"]
    assert pg["world_size"] == 2 and pg["backend"] == "gloo" and [r["rank"] for r in pg["ranks"]] == [0, 1]
    assert d["value"] == pytest.approx(2 * 4096 * 4096 * 2 / (d["ms_per_step"] * 2 * 1e-3) / 1e6, rel=1e-6)
    assert "cpu_baseline" not in d                      # rank 0 at N = 1 only
    c5 = d["c5"]
    assert c5["ranks"] == 2 and c5["images"] == 8 and c5["gather_checked"] is True and c5["gather_ms"] > 0
    assert c5["gather_bytes"] == 4 * 2048 * 2048 * 8
    assert d["roofline"]["frac"] > 0
    assert d["output_check"]["ok"] is True and d["output_check"]["rel_err"] <= 1e-6   # rank 0's image IS the c2 fixture


def test_bench_two_ranks_full_shard_and_no_gather():
    """The same dry run at the FULL per-rank shard of BASELINE config #5 (64 images of 2048 x 2048 per rank): rank 0 receives every
    rank's block in one pre-allocated (world, B, N, M) tensor, the block check passes, and the high-water mark of torch's allocations
    on rank 0 is on record; then once more with --no-gather, which must leave the solver figures and drop the gather's."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PROXTV_BENCH_SHARED_GPU="1")

    def run(extra):
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1"] + extra
        out = subprocess.run(cmd, env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, out.stderr[-3000:]
        lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
        assert len(lines) == 1, out.stdout[-2000:]
        return json.loads(lines[0])["c5"]

    c5 = run(["--c5-images", "64"])
    assert c5["ranks"] == 2 and c5["images"] == 128 and c5["gather_checked"] is True and c5["gather_ms"] > 0
    assert c5["gather_bytes"] == 64 * 2048 * 2048 * 8
    # (shared-GPU dry run: the blocks travel as host copies over gloo, so torch's DEVICE high-water mark holds x5 and y5 only;
    #  on a real node the (world, B, N, M) receive tensor adds world x 2 GiB on rank 0)
    assert c5["rank0_torch_hbm_peak_bytes"] >= 2 * 64 * 2048 * 2048 * 8
    c5n = run(["--c5-images", "8", "--no-gather"])
    assert c5n["ranks"] == 2 and c5n["images"] == 16 and c5n["gather_ms"] is None and c5n["gather_bytes"] is None
    assert "skipped" in c5n["gather"] and c5n["solve_ms"] > 0


def test_full_size_solves_under_the_certifier(ptv, clib, glarge):
    """Configs #2 and #3 and a full-size PD2 with option certify on: every one of their sweeps -- 72 per DR solve, the late iterates with
    their near-ties included -- is checked against the optimality conditions of the prox, fibre by fibre (4096 fibres of 4096 samples a
    sweep).  No fibre fails, and the results are the reference's."""
    rng = np.random.default_rng(0)
    X = np.asfortranarray(rng.standard_normal((4096, 4096)))
    W1 = np.asfortranarray(rng.uniform(0.05, 0.15, (4095, 4096)))
    W2 = np.asfortranarray(rng.uniform(0.05, 0.15, (4096, 4095)))
    c = {k: clib.proxtv_debug_counter(k) for k in (b"certify_sweeps", b"certify_failures", b"certify_skipped")}
    before = clib.proxtv_set_option(b"certify", 1)
    try:
        _check_digest(ptv.tv1_2d(X, 0.1), glarge, "c2/dr2")
        _check_digest(ptv.tv1w_2d(X, W1, W2), glarge, "c3/dr2w")
        y = ptv.tv1_2d(X, 0.1, method="pd")
        _check_digest(ptv.tv1_2d(X, 1.0), glarge, "lam1/dr2")          # (the pinning rung)
    finally:
        clib.proxtv_set_option(b"certify", before)
    assert abs(y.mean() - X.mean()) < 1e-9
    assert clib.proxtv_debug_counter(b"certify_failures") == c[b"certify_failures"]
    assert clib.proxtv_debug_counter(b"certify_sweeps") - c[b"certify_sweeps"] >= 3 * 71
    assert clib.proxtv_debug_counter(b"certify_skipped") == c[b"certify_skipped"]
