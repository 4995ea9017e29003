"""The speculative-chunk machinery under stress: data whose walks do NOT meet inside the warm-up zone, so links stay
unproven, windows overflow and the repair kernel / geometry policy must keep the result exact.  Every case is
checked against the CPU oracle and across the pinned geometry modes (0: LDS window, 16-sample zones; 1: the same with
second-chance rounds inside a block; 2: LDS window, 64-sample zones; 3: the pinning solver -- or, where it does not apply,
global-memory chunks with 256-sample zones; 4: global-memory chunks, 1024-sample zones; 5: one sequential walk per fibre)."""
import numpy as np
import pytest

from conftest import assert_close

pytestmark = pytest.mark.gpu


@pytest.fixture()
def modes(clib):
    """Pin a geometry mode for the duration of a test; what was set before (PROXTV_CHUNK_MODE pins the whole run) returns."""
    before = clib.proxtv_set_option(b"chunk_mode", -1)
    clib.proxtv_set_option(b"chunk_mode", before)

    def set_mode(m):
        clib.proxtv_set_option(b"chunk_mode", m)
    yield set_mode
    clib.proxtv_set_option(b"chunk_mode", before)


def _signals(rng, n):
    yield "randn", rng.standard_normal(n)
    yield "blocks", np.repeat(rng.standard_normal(n // 50 + 1), 50)[:n] + 0.2 * rng.standard_normal(n)
    yield "walk", np.cumsum(rng.standard_normal(n)) * 0.3
    yield "ramp+noise", np.linspace(-5, 5, n) + 0.05 * rng.standard_normal(n)
    yield "constant", np.full(n, 1.25)
    yield "sparse spikes", (rng.random(n) < 0.01) * 10.0 * rng.standard_normal(n)


def test_long_fibres_all_modes_vs_oracle(ptv, clib, oracle, modes):
    """Single long fibres (many chunks per fibre) over lambdas from 'every sample bends' to 'one flat piece'."""
    rng = np.random.default_rng(41)
    for name, x in _signals(rng, 5000):
        for lam in (0.05, 0.5, 3.0, 40.0):
            want = oracle.tv1_hybrid(x, lam)
            for m in (0, 1, 2, 3, 4, 5):
                modes(m)
                got = ptv.tv1_1d(x, lam)
                assert_close(got, want, tol=1e-11, what=f"{name} lam={lam} mode={m}")


def test_repairs_happen_and_are_exact(ptv, clib, oracle, modes):
    """lambda = 1 on unit noise: walks need ~10-70 samples to meet, so 16-sample zones leave many links unproven."""
    rng = np.random.default_rng(42)
    X = rng.standard_normal((700, 900))
    want = oracle.dr2(X, 1.0)[0]
    modes(0)
    got0 = ptv.tv1_2d(X, 1.0)
    fix0 = clib.proxtv_last_fixups()
    assert fix0 > 0, "expected unproven links with 16-sample zones at lambda = 1"
    assert_close(got0, want, tol=1e-11, what="mode 0")
    modes(1)                                           # second chances inside the blocks: fewer fibres left to repair
    assert_close(ptv.tv1_2d(X, 1.0), want, tol=1e-11, what="mode 1")
    assert clib.proxtv_last_fixups() <= fix0
    modes(2)
    assert_close(ptv.tv1_2d(X, 1.0), want, tol=1e-11, what="mode 2")
    assert clib.proxtv_last_fixups() < fix0            # longer zones prove (almost) every link
    modes(3)
    assert_close(ptv.tv1_2d(X, 1.0), want, tol=1e-11, what="mode 3")
    assert clib.proxtv_last_fixups() < fix0            # 256-sample zones from global memory
    modes(4)                                           # fibres shorter than 1024: falls through to the sequential walk
    assert_close(ptv.tv1_2d(X, 1.0), want, tol=1e-11, what="mode 4")
    modes(5)
    assert_close(ptv.tv1_2d(X, 1.0), want, tol=1e-11, what="mode 5")
    assert clib.proxtv_last_fixups() == 0


def test_global_chunks_long_pieces(ptv, clib, oracle, modes):
    """Large lambda on long fibres: pieces of hundreds of samples.  The global-memory chunk modes must agree with the
    oracle whether their zones are long enough (few repairs) or not (every chunk repaired)."""
    rng = np.random.default_rng(46)
    X = rng.standard_normal((2200, 130))
    for lam in (3.0, 8.0):
        want = oracle.dr2(X, lam)[0]
        for m in (1, 3, 4, 5, -1):
            modes(m)
            assert_close(ptv.tv1_2d(X, lam), want, tol=1e-10, what=f"lam={lam} mode {m}")
    torch = pytest.importorskip("torch")
    from proxtv_amd import device
    A = np.cumsum(rng.standard_normal((3000, 70)), axis=0) * 0.05 + rng.standard_normal((3000, 70))
    for lam in (2.0, 6.0):
        want = np.apply_along_axis(lambda f: oracle.tv1_hybrid(f, lam), 0, A)
        for m in (3, 4):
            modes(m)
            ad = device.to_colmajor(torch.from_numpy(A).cuda())
            assert_close(device.tv1_fibres(ad, lam, 0).cpu().numpy(), want, tol=1e-11, what=f"dim0 lam={lam} mode={m}")
            bd = device.to_colmajor(torch.from_numpy(np.ascontiguousarray(A.T)).cuda())
            assert_close(device.tv1_fibres(bd, lam, 1).cpu().numpy(), want.T, tol=1e-11, what=f"dim1 lam={lam} mode={m}")


def test_lively_stretches_followed_by_flat_ones(clib, oracle, modes):
    """Fibres whose lively stretches (unit noise) alternate with flat ones longer than a workgroup's span, through the
    speculative rungs pinned, against the oracle fibre by fibre.  In a flat stretch every link across workgroups is in
    doubt (no bend in the warm-up zone to start from), and the repair kernel jumps from the chunk that took a repair walk
    over straight to the next such link: the bend its next walk starts from is the record of the chunk before that link,
    which the chunk kernel proved -- never one of the records further back, which a repair walk may have made stale
    (the host model of the stage: tests/test_repair_model_host.py)."""
    torch = pytest.importorskip("torch")
    from proxtv_amd import device
    rng = np.random.default_rng(4242)
    m, n = 2048, 192
    A = np.empty((m, n))
    for j in range(n):
        k, lively = 0, bool(rng.integers(0, 2))
        while k < m:
            span = int(rng.integers(30, 330)) if lively else int(rng.integers(100, 700))
            A[k:k + span, j] = rng.normal() * 2 + (rng.standard_normal(min(span, m - k)) if lively else 0.0)
            k += span
            lively = not lively
    ad = device.to_colmajor(torch.from_numpy(A).cuda())
    bd = device.to_colmajor(torch.from_numpy(np.ascontiguousarray(A.T)).cuda())
    before = clib.proxtv_set_option(b"tile", 1)
    try:
        for lam in (0.3, 0.7, 1.0, 1.6):
            want = np.apply_along_axis(lambda f: oracle.tv1_hybrid(f, lam), 0, A)
            for m_, tile in ((0, 1), (1, 1), (1, 0), (2, 1)):
                modes(m_)
                clib.proxtv_set_option(b"tile", tile)
                assert_close(device.tv1_fibres(ad, lam, 0).cpu().numpy(), want, tol=1e-10, what=f"dim0 lam={lam} rung {m_}")
                assert_close(device.tv1_fibres(bd, lam, 1).cpu().numpy(), want.T, tol=1e-10, what=f"dim1 lam={lam} rung {m_} tile {tile}")
    finally:
        clib.proxtv_set_option(b"tile", before)


@pytest.mark.parametrize("deterministic", [1, 0])
def test_policy_escalates_and_recovers(ptv, clib, oracle, modes, deterministic):
    """The policy, seeded from the input's statistics (deterministic = 1: the default) or hill-climbing on measured times
    (0): easy data runs the 16-sample-zone LDS geometry (mode 0, or its robust instantiation 1 when the timing of a trial
    on so small an image is a toss-up); moderate lambda moves up the ladder; an easy problem afterwards comes back down.
    Results are exact throughout."""
    modes(-1)
    det = clib.proxtv_set_option(b"deterministic", deterministic)
    try:
        _escalates_and_recovers(ptv, clib, oracle)
    finally:
        clib.proxtv_set_option(b"deterministic", det)


def _escalates_and_recovers(ptv, clib, oracle):
    rng = np.random.default_rng(43)
    X = rng.standard_normal((600, 640))
    assert_close(ptv.tv1_2d(X, 0.1), oracle.dr2(X, 0.1)[0], tol=1e-11)
    assert_close(ptv.tv1_2d(X, 0.1), oracle.dr2(X, 0.1)[0], tol=1e-11)
    assert clib.proxtv_chunk_mode() <= 1 and clib.proxtv_last_fixups() == 0
    assert_close(ptv.tv1_2d(X, 1.0), oracle.dr2(X, 1.0)[0], tol=1e-11)
    assert clib.proxtv_chunk_mode() >= 1
    assert_close(ptv.tv1_2d(X, 30.0), oracle.dr2(X, 30.0)[0], tol=1e-11)       # one piece per fibre: hopeless for chunks
    for _ in range(3):
        assert_close(ptv.tv1_2d(X, 0.1), oracle.dr2(X, 0.1)[0], tol=1e-11)
    assert clib.proxtv_chunk_mode() <= 1


def test_weighted_and_nd_under_repair(ptv, clib, oracle, modes):
    rng = np.random.default_rng(44)
    X = rng.standard_normal((520, 300))
    W1, W2 = rng.uniform(0.5, 1.5, (519, 300)), rng.uniform(0.5, 1.5, (520, 299))
    want = oracle.dr2w(X, W1, W2)[0]
    for m in (0, 1, 2, 3, 5, -1):
        modes(m)
        assert_close(ptv.tv1w_2d(X, W1, W2), want, tol=1e-11, what=f"weighted mode {m}")
    V = rng.standard_normal((300, 280, 6))
    wantv = oracle.pd(V, [0.8, 0.9, 0.2], [1, 2, 3])[0]
    for m in (0, 1, 2, 3, 5, -1):
        modes(m)
        assert_close(ptv.tvgen(V, [0.8, 0.9, 0.2], [1, 2, 3], [1, 1, 1]), wantv, tol=1e-10, what=f"pd mode {m}")


def test_fibre_lengths_around_chunk_and_block_edges(ptv, clib, oracle, modes):
    """Lengths from the chunked path's lower bound (96) up, not multiples of the chunk (16) / block (128) sizes, for both
    sweep orientations."""
    rng = np.random.default_rng(45)
    torch = pytest.importorskip("torch")
    from proxtv_amd import device
    for n in (96, 97, 111, 112, 127, 128, 129, 200, 255, 256, 257, 271, 272, 383, 384, 385, 511, 1000):
        A = rng.standard_normal((n, 70))
        for lam, m in ((0.1, 0), (0.7, 0), (0.7, 1), (0.7, 2), (0.7, 3)):
            modes(m)
            ad = device.to_colmajor(torch.from_numpy(A).cuda())
            got = device.tv1_fibres(ad, lam, 0).cpu().numpy()                       # contiguous fibres of length n
            want = np.apply_along_axis(lambda f: oracle.tv1_hybrid(f, lam), 0, A)
            assert_close(got, want, tol=1e-11, what=f"dim0 n={n} lam={lam} mode={m}")
            bd = device.to_colmajor(torch.from_numpy(np.ascontiguousarray(A.T)).cuda())
            got = device.tv1_fibres(bd, lam, 1).cpu().numpy()                       # strided fibres of length n
            assert_close(got, want.T, tol=1e-11, what=f"dim1 n={n} lam={lam} mode={m}")


def _needs_xlink(clib):
    """The jobs repair hangs on the sweep's `dirty` word, which exists only with the in-kernel link check (option xlink, default 1)."""
    was = clib.proxtv_set_option(b"xlink", 1)
    clib.proxtv_set_option(b"xlink", was)
    if was == 0:
        pytest.skip("option xlink = 0 (PROXTV_XLINK): no dirty word, the jobs repair is never launched")



def test_jobs_repair_equals_the_sequential_repair(clib, oracle, modes):
    """Option repair_jobs: failed links across workgroups repaired one lane per failure (sweep_repair_jobs_kernel) before the
    sequential repair kernel takes what is left.  Rung 1 at lambda 0.65-0.9 on unit noise is where such links fail in numbers.
    Every case is solved with the option off (0), gated by the sampled statistic (1: the default) and always on (2): the three
    results must be the same to the last bit -- a job parks exactly the values the sequential walk would write -- and exact."""
    import torch
    from proxtv_amd import device
    _needs_xlink(clib)
    rng = np.random.default_rng(5)
    dev = lambda a: device.to_colmajor(torch.from_numpy(np.ascontiguousarray(a)).cuda())
    before = clib.proxtv_set_option(b"repair_jobs", 1)
    tile_before = clib.proxtv_set_option(b"tile", 1)
    launched = 0
    try:
        modes(1)
        cases = []
        for n, m in ((1024, 1024), (777, 1500)):
            X = rng.standard_normal((n, m))
            Xd = dev(X)
            for lam in (0.4, 0.7, 0.9):
                cases.append((f"DR {n}x{m} lam {lam}", lambda Xd=Xd, lam=lam: device.tv1_2d(Xd, lam)[0], (X, lam)))
            for d in (0, 1):
                cases.append((f"one sweep dim {d} {n}x{m}", lambda Xd=Xd, d=d: device.tv1_fibres(Xd, 0.7, d), None))
        X = rng.standard_normal((1536, 640))
        Xd = dev(X)
        W1, W2 = dev(rng.uniform(0.35, 1.05, (1535, 640))), dev(rng.uniform(0.35, 1.05, (1536, 639)))
        for tile in (0, 1):
            cases.append((f"DR 1536x640 lam 0.7 tile={tile}", lambda tile=tile: (clib.proxtv_set_option(b"tile", tile), device.tv1_2d(Xd, 0.7)[0])[1], None))
            cases.append((f"weighted DR tile={tile}", lambda tile=tile: (clib.proxtv_set_option(b"tile", tile), device.tv1w_2d(Xd, W1, W2)[0])[1], None))
        V = dev(rng.standard_normal((256, 320, 24)))
        cases.append(("tvgen PD 3-D lam 0.7", lambda: (clib.proxtv_set_option(b"tile", 1), device.tvgen(V, [0.7, 0.7, 0.7], [1, 2, 3])[0])[1], None))
        Xpd = dev(rng.standard_normal((1024, 1024)))
        cases.append(("PD2 1024^2 lam 0.7", lambda: device.tv1_2d(Xpd, 0.7, method="pd")[0], None))
        for label, run, ref in cases:
            outs, fixes = [], []
            for jobs in (0, 1, 2):
                clib.proxtv_set_option(b"repair_jobs", jobs)
                c0 = clib.proxtv_debug_counter(b"repair_jobs_launches")
                outs.append(run().clone())
                fixes.append(clib.proxtv_last_fixups())
                n_launched = clib.proxtv_debug_counter(b"repair_jobs_launches") - c0
                if jobs == 0:
                    assert n_launched == 0, label
                if jobs == 2:
                    launched += n_launched
            assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2]), (label, fixes)
            assert fixes[0] == fixes[2], (label, fixes)   # the same fibres were repaired, whoever repaired them
            if ref is not None:
                assert_close(outs[2].cpu().numpy(), oracle.dr2(ref[0], ref[1])[0], tol=1e-11, what=label)
    finally:
        clib.proxtv_set_option(b"repair_jobs", before)
        clib.proxtv_set_option(b"tile", tile_before)
    assert launched > 0   # (a run in which the jobs kernel never had a launch cannot pass for a green one)


def test_jobs_repair_is_gated_by_the_sampled_statistic(clib):
    """Default policy (repair_jobs = 1): the jobs kernel is launched only where the sampled certain fraction says links across
    workgroups fail in numbers (lambda >= 0.65 on unit noise); the headline regime never pays for it."""
    import torch
    from proxtv_amd import device
    _needs_xlink(clib)
    X = device.to_colmajor(torch.from_numpy(np.random.default_rng(7).standard_normal((2048, 2048))).cuda())
    assert clib.proxtv_set_option(b"repair_jobs", 1) in (0, 1, 2)
    if clib.proxtv_set_option(b"chunk_mode", -1) != -1:
        pytest.skip("a pinned rung (PROXTV_CHUNK_MODE) overrides the seeded policy")
    c0 = clib.proxtv_debug_counter(b"repair_jobs_launches")
    device.tv1_2d(X, 0.1)
    assert clib.proxtv_debug_counter(b"repair_jobs_launches") == c0
    device.tv1_2d(X, 0.7)
    assert clib.proxtv_debug_counter(b"repair_jobs_launches") > c0


def test_hand_over_at_a_knot_with_zero_jump(ptv, clib, oracle, modes):
    """tests/golden/degenerate_knot_fibre.npz: late in a Dykstra loop the operand reproduces the previous result on whole stretches and the
    string has knots whose jump is zero up to rounding -- a bend to the repair walk, none to the chunk's own walk, which round differently.
    The repair walk hands over to a chunk that began at the same bend; the chunk's rows behind it must be what the chunk's walk found
    (round 5, fuzz seed 111 case 1260: rows 94, 95 were off by 0.028 in pinned mode 0).  As columns (along-fibre kernel) and as rows
    (tiles), alone and among noise, every pinned rung."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "degenerate_knot_fibre.npz"))
    y, lam, want = g["y"], float(g["lam"]), g["expected"]
    rng = np.random.default_rng(47)
    cols = np.asfortranarray(np.repeat(y[:, None], 96, axis=1))
    cols[:, 1::3] += 1e-3 * rng.standard_normal((y.size, 32))          # (neighbours that are not the same fibre)
    want_cols = np.apply_along_axis(lambda f: oracle.tv1_hybrid(np.ascontiguousarray(f), lam), 0, cols)
    assert np.max(np.abs(want_cols[:, 0] - want)) <= 1e-14
    for m in (0, 1, 2, 3, 4, 5, -1):
        modes(m)
        assert_close(ptv.tv1_1d(y, lam), want, tol=1e-12, what=f"the fibre alone, mode {m}")
        assert_close(ptv.tvgen(cols, [lam], [1], [1]), want_cols, tol=1e-12, what=f"as columns, mode {m}")
        rows = np.asfortranarray(cols.T)
        assert_close(ptv.tvgen(rows, [lam], [2], [1]), want_cols.T, tol=1e-12, what=f"as rows, mode {m}")
        for jobs in (0, 2):
            before = clib.proxtv_set_option(b"repair_jobs", jobs)
            try:
                assert_close(ptv.tvgen(cols, [lam], [1], [1]), want_cols, tol=1e-12, what=f"as columns, mode {m}, repair_jobs {jobs}")
            finally:
                clib.proxtv_set_option(b"repair_jobs", before)
