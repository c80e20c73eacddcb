"""The natural-order narrow pass (round 5: k_narrow, include/moira_pb.h mpb_path_info) against the oracle and the reference's
vectors.  Whichever pass computes a read -- the narrow pass, or the sorted pipeline on the sub-batch it hands back -- the
result must be the reference's bit for bit (moira/bernoullimodule.c:152-166,219-251), so every test here forces the pass
(MPB_FLAG_NARROW_ROWS) on batches it would never choose, and compares with the same batch through MPB_FLAG_NO_NARROW."""
import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from moira_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def run_device(eng, q, fixed_len, **kw):
    """A host matrix through mpb_filter_device (resident batch) -> (ee, ns, pass, counts, path)."""
    n, stride = q.shape
    d_q, d_ee, d_ns, d_pass = eng.alloc(max(1, n * stride)), eng.alloc(max(1, n * 8)), eng.alloc(max(1, n * 4)), eng.alloc(max(1, n))
    try:
        d_q.upload(np.ascontiguousarray(q))
        # results of an earlier call must never be mistaken for this one's
        d_ee.upload(np.full(n, -7.0)); d_ns.upload(np.full(n, -7, np.int32)); d_pass.upload(np.full(n, 9, np.uint8))
        c = eng.filter_device(d_q, n, stride, fixed_len=fixed_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(**kw))
        path = eng.last_path()
        return d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n), c, path
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()


@pytest.mark.parametrize("rows", [2, 3, 4])
@pytest.mark.parametrize("profile", [0, 1])
def test_forced_narrow_is_the_oracle(eng, oracle, rows, profile):
    """BASELINE's synthetic model (median 5 rows, a quarter of the reads with an N: most reads are handed back) and the clean
    profile (nearly every read finished by the pass itself)."""
    n, stride, L = 70_001, 320, 300                      # not a multiple of 64: the last row block is partial
    q, _ = oracle.synth_fill(n, stride, fixed_len=L, seed=11, profile=profile)
    ee, ns, ps, need = oracle.filter_batch(q, fixed_len=L, threads=8)
    e1, n1, p1, c, path = run_device(eng, q, L, narrow_rows=rows)
    assert path["narrow_rows"] == rows
    assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps)
    assert (c.n_reads, c.n_pass, c.n_fail) == (n, int(ps.sum()), n - int(ps.sum()))
    # exactly the reads the pass cannot finish are handed back: more rows than it holds (or a lower-case 'n': none here)
    assert path["n_fallback"] == int((need > rows).sum())
    if profile == 1:
        assert path["n_fallback"] == 0 and (need == 2).all() and ns.max() > 0


@pytest.mark.parametrize("L,stride", [(1, 16), (3, 16), (15, 16), (16, 16), (17, 32), (63, 64), (64, 64), (65, 80), (100, 112),
                                      (127, 128), (128, 128), (129, 144), (250, 256), (299, 304), (300, 304), (301, 304),
                                      (300, 320), (320, 320), (600, 608), (1000, 1008), (1023, 1024), (1500, 1536),
                                      # k_narrow_rs (strides that are a multiple of 64): one read per lane (stride % 128 == 0) and
                                      # two (stride % 128 == 64); reads that end on a half panel, one chunk into one, and far before
                                      # the row does (whole halves and panels of padding are skipped)
                                      (10, 320), (64, 320), (130, 320), (257, 320), (40, 256), (65, 128), (100, 192), (191, 192),
                                      (383, 384), (200, 384), (440, 448), (2000, 2048), (30, 1024)])
def test_lengths_and_strides(eng, oracle, L, stride):
    """Every tail shape of the panel walk: lengths that end inside a dword, a chunk, a panel; strides that are not a multiple of
    the 64-byte panel (k_narrow: the last panel's DMA is clamped into the row) and strides that are (k_narrow_rs: a lane walks
    one or two rows as one stream of whole 128-byte lines); garbage past the read's end."""
    rng = np.random.default_rng(L * 1000 + stride)
    n = 1000 + (L % 7)
    q = rng.integers(20, 41, (n, stride), dtype=np.uint8)
    q[rng.random(n) < 0.3, :] = rng.integers(2, 41, stride, dtype=np.uint8)      # a third of the reads are bad ones
    q[:, L:] = rng.integers(0, 256, (n, stride - L), dtype=np.uint8)             # padding: anything
    hit = rng.random((n, L)) < 0.002
    q[:, :L][hit] = rng.choice(np.array([0, 255], np.uint8), int(hit.sum()))
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=L, threads=8)
    for rows in (2, 4):
        e1, n1, p1, c, path = run_device(eng, q, L, narrow_rows=rows)
        assert path["narrow_rows"] == rows
        assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps), (L, stride, rows)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 256, 257, 4097])
def test_small_and_partial_row_blocks(eng, oracle, n):
    q, _ = oracle.synth_fill(n, 320, fixed_len=300, seed=5, profile=1)
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=4)
    e1, n1, p1, c, path = run_device(eng, q, 300, narrow_rows=2)
    assert path["narrow_rows"] == 2
    assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps) and c.n_pass == int(ps.sum())


@pytest.mark.parametrize("kw", [dict(ambigs="treat_as_errors"), dict(ambigs="ignore"), dict(ambigs="disallow"),
                                dict(ambigs="treat_as_errors", round_=True), dict(ambigs="ignore", maxerrors=0.4),
                                dict(alpha=0.05, uncert=0.002), dict(alpha=0.3), dict(alpha=1e-4), dict(alpha=0.9)])
def test_modes(eng, oracle, kw):
    """--ambigs / --round / --maxerrors / alpha: the pass' epilogue is the tile classes' epilogue.  alpha = 0.9 and 0.3 put the
    crossing on the FIRST row for clean reads (the reference's undefined case: ee = 0 by the Python twin's definition)."""
    n, L = 20_000, 300
    q, _ = oracle.synth_fill(n, 320, fixed_len=L, seed=21, profile=1)
    q[::7, 5] = 0
    q[::11, 17] = 255
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=L, threads=8, **kw)
    for rows in (2, 3):
        e1, n1, p1, c, path = run_device(eng, q, L, narrow_rows=rows, **kw)
        assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps), (kw, rows)
        assert c.n_pass == int(ps.sum())


def test_reference_vectors_through_the_narrow_pass(eng):
    """The reference's own results (tests/golden/*.npz: bernoullimodule.c, and its Python twin where C is undefined), every
    read of every set of up to 1023 bases, grouped by length into fixed-length batches and forced through the pass."""
    done = 0
    for name in G.NPZ_SETS:
        if name == "long_reads":
            continue
        s = G.load_set(name)
        q, lens, exp = s["q"], s["lens"], G.expected_value(s)
        for L in np.unique(lens):
            if L < 1:
                continue
            idx = np.nonzero(lens == L)[0]
            for rows in (2, 4):
                e1, n1, p1, c, path = run_device(eng, q[idx], int(L), narrow_rows=rows, alpha=float(s["alpha"]), ambigs="ignore")
                assert path["narrow_rows"] == rows
                assert same(e1, exp[idx]), (name, int(L), rows)
                assert np.array_equal(n1, s["ns_ref"][idx])
            done += len(idx)
    assert done > 10_000


def test_the_choice(eng, oracle):
    """A clean batch takes the pass by itself (from a sample of <= 0.1 % of its reads), BASELINE's model does not; the decision
    is reused for the next batch of the same shape; MPB_FLAG_NO_NARROW wins over everything; results never depend on it."""
    nmax, stride, L = 400_000, 320, 300
    d_q, d_ee, d_ns, d_pass = eng.alloc(nmax * stride), eng.alloc(nmax * 8), eng.alloc(nmax * 4), eng.alloc(nmax)
    try:
        for profile, want_rows, n in ((1, 2, nmax), (0, 0, nmax - 10_000)):      # another shape: the first decision is not reused
            eng.synth_fill(d_q, n, stride, fixed_len=L, seed=31, profile=profile)
            hq = d_q.download(np.uint8, n * stride).reshape(n, stride)
            ee, ns, ps, _ = oracle.filter_batch(hq, fixed_len=L, threads=16)
            c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
            p1 = eng.last_path()
            assert p1["sampled"] and sum(p1["sample_hist"]) <= max(256, n // 1000) and p1["narrow_rows"] == want_rows, p1
            assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_ns.download(np.int32, n), ns)
            assert np.array_equal(d_pass.download(np.uint8, n), ps) and c.n_pass == int(ps.sum())
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
            p2 = eng.last_path()
            assert not p2["sampled"] and p2["narrow_rows"] == want_rows
            assert same(d_ee.download(np.float64, n), ee)
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(no_narrow=True))
            assert eng.last_path()["narrow_rows"] == 0
            assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_pass.download(np.uint8, n), ps)
        # a small batch never takes the pass by itself
        eng.filter_device(d_q, 100_000, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        assert eng.last_path()["narrow_rows"] == 0
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()


def test_a_stale_choice_costs_time_not_results(eng, oracle):
    """The decision is reused while the shape stays: fill the same buffer with BASELINE's model after a clean batch chose the
    pass.  Nearly every read is handed back -- results are the oracle's, and the next call looks again."""
    n, stride, L = 300_000, 320, 300
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    try:
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=41, profile=1)
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        assert eng.last_path()["narrow_rows"] == 2
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=41, profile=0)
        hq = d_q.download(np.uint8, n * stride).reshape(n, stride)
        ee, ns, ps, _ = oracle.filter_batch(hq, fixed_len=L, threads=16)
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        p = eng.last_path()
        assert p["narrow_rows"] == 2 and not p["sampled"] and p["n_fallback"] > n // 2
        assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_ns.download(np.int32, n), ns)
        assert np.array_equal(d_pass.download(np.uint8, n), ps)
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        p = eng.last_path()
        assert p["sampled"] and p["narrow_rows"] == 0
        assert same(d_ee.download(np.float64, n), ee)
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()


def test_high_quality_full_size(eng, oracle):
    """bench.py's extras.high_quality_300 at its size: 10 M x 300 bp of the clean profile, resident, the pass chosen by the
    library; every read compared with the oracle."""
    from test_gpu_parity import compare_every_read
    n, stride, L, seed = 10_000_000, 320, 300, 2
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    try:
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed, profile=1)
        c1 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        path = eng.last_path()
        assert path["narrow_rows"] == 2 and path["n_fallback"] == 0, path
        ee1, ns1, ps1 = d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n)
        assert c1.n_pass == int(ps1.sum()) and not np.isnan(ee1).any()
        assert compare_every_read(eng, oracle, d_q, n, stride, ee1, ns1, ps1, fixed_len=L, label="high_quality_300 (narrow pass)") == n
        c2 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(no_narrow=True))
        assert same(d_ee.download(np.float64, n), ee1) and c2.n_pass == c1.n_pass
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()


def test_the_handed_back_reads_need_no_memory_of_their_own(eng, oracle):
    """ADVICE r5 (a stale choice + a tight HBM budget made the call fail with MPB_E_NOMEM: the reads a narrow pass handed back were
    gathered into a dense block of their own).  Round 6: they run through the sorted pipeline where they lie (the prepass and the
    scatter walk the list), so the call needs no block at all -- here the test takes nearly all of the device's free memory first,
    forces the pass on BASELINE's model (most reads handed back) and gets the oracle's results.  And when a stale choice meets a batch
    of bad reads (more than a quarter handed back, pass not forced) the whole batch takes the sorted pipeline (n_fallback == n)."""
    import ctypes as C
    hip = C.CDLL("libamdhip64.so")
    n, stride, L = 1_500_000, 320, 300
    q, _ = oracle.synth_fill(n, stride, fixed_len=L, seed=77, profile=0)            # BASELINE's model: most reads are handed back
    ee, ns, ps, need = oracle.filter_batch(q, fixed_len=L, threads=16)
    bufs = [eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
    d_q, d_ee, d_ns, d_pass = bufs
    hog = None
    try:
        d_q.upload(q)
        prm = eng.params(narrow_rows=2)
        # once with room: the workspaces of the pass and of the sorted pipeline exist from here on
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm)
        free, total = C.c_size_t(0), C.c_size_t(0)
        assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
        handed = int((need > 2).sum())
        assert handed * (stride + 17) > 200 << 20                                  # what a dense copy of them would take: hundreds of MB
        hog = eng.alloc(max(1, free.value - (64 << 20)))                           # leave about 64 MB
        d_ee.upload(np.full(n, -7.0))
        c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm)
        path = eng.last_path()
        assert path["narrow_rows"] == 2 and path["n_fallback"] == handed
        assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_ns.download(np.int32, n), ns)
        assert np.array_equal(d_pass.download(np.uint8, n), ps) and c.n_pass == int(ps.sum())
    finally:
        for b in bufs + ([hog] if hog is not None else []):
            b.free()
