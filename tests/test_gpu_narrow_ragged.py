"""The narrow pass on RAGGED batches (round 6: k_rag_sort, k_rag_scan, k_narrow_rg; include/moira_pb.h mpb_path_info): one padded
matrix + int32 len[], as the reference's paired mode produces them (moira/moira.py:789-801 -> make_contig, :1376-1558).
Whichever pass computes a read -- the narrow pass in its length-sorted order, or the sorted pipeline on the sub-batch it hands
back -- the result must be the reference's bit for bit (moira/bernoullimodule.c:152-166,219-251; the per-read limit
len(sequence) * uncert of moira/moira.py:949-950).  Every test forces the pass (MPB_FLAG_NARROW_ROWS) on the batch AS IT IS: no
regrouping by length on the host."""
import os

import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    from moira_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def run_ragged(eng, q, lens, want_counts=True, **kw):
    """A host matrix + lengths through mpb_filter_device (resident batch) -> (ee, ns, pass, counts, path)."""
    n, stride = q.shape
    bufs = [eng.alloc(max(1, n * stride)), eng.alloc(max(1, n * 8)), eng.alloc(max(1, n * 4)), eng.alloc(max(1, n)), eng.alloc(max(1, n * 4))]
    d_q, d_ee, d_ns, d_pass, d_len = bufs
    try:
        d_q.upload(np.ascontiguousarray(q))
        d_len.upload(np.ascontiguousarray(lens, dtype=np.int32))
        # results of an earlier call must never be mistaken for this one's
        d_ee.upload(np.full(n, -7.0)); d_ns.upload(np.full(n, -7, np.int32)); d_pass.upload(np.full(n, 9, np.uint8))
        c = eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(**kw),
                              want_counts=want_counts)
        path = eng.last_path()
        return d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n), c, path
    finally:
        for b in bufs:
            b.free()


@pytest.mark.parametrize("rows", [2, 3, 4])
@pytest.mark.parametrize("profile,stride", [(0, 608), (1, 640), (1, 608)])
def test_forced_ragged_narrow_is_the_oracle(eng, oracle, rows, profile, stride):
    """BASELINE configs[4] (lengths U{50..600}) in BASELINE's quality model (most reads handed back) and in the clean profile
    (nearly every read finished by the pass); a stride that is a multiple of 128 (rows are whole lines) and one that is not."""
    n = 70_001                                           # not a multiple of 64 or 4096: partial last group and window
    q, lens = oracle.synth_fill(n, stride, min_len=50, max_len=600, seed=13, profile=profile)
    ee, ns, ps, need = oracle.filter_batch(q, lens=lens, threads=8)
    e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows)
    assert path["narrow_rows"] == rows
    assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps)
    assert (c.n_reads, c.n_pass, c.n_fail) == (n, int(ps.sum()), n - int(ps.sum()))
    # exactly the reads the pass cannot finish are handed back: more rows than it holds (no lower-case 'n' here)
    assert path["n_fallback"] == int((need > rows).sum())
    if profile == 1:                                     # (clean reads of more than ~400 bases need a third row)
        assert (need <= 3).all() and (rows == 2 or path["n_fallback"] == 0)


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 255, 4095, 4096, 4097, 8193, 12_345])
def test_small_batches_partial_groups_and_windows(eng, oracle, n):
    q, lens = oracle.synth_fill(n, 640, min_len=50, max_len=600, seed=5 + n, profile=1)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=4)
    e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=2)
    assert path["narrow_rows"] == 2
    assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps) and c.n_pass == int(ps.sum())


@pytest.mark.parametrize("stride,lo,hi", [(16, 0, 16), (48, 1, 48), (128, 0, 128), (304, 100, 300), (320, 290, 310), (512, 241, 502),
                                          (1008, 900, 1008), (1024, 1, 1023), (2048, 30, 2040), (4096, 3000, 4096), (4096, 0, 4096)])
def test_lengths_and_strides(eng, oracle, stride, lo, hi):
    """Lengths that end inside a dword, a chunk, a half, a panel; empty reads and reads that fill their row; rows that are no
    multiple of a line; coarse sort keys (rows of more than 1008 bytes); garbage past each read's end; a third of the reads bad."""
    rng = np.random.default_rng(stride * 7 + lo)
    n = 9_000 + stride % 11
    q = rng.integers(25, 41, (n, stride), dtype=np.uint8)
    bad = rng.random(n) < 0.3
    q[bad, :] = rng.integers(2, 41, (int(bad.sum()), stride), dtype=np.uint8)
    lens = rng.integers(lo, hi + 1, n).astype(np.int32)
    lens[:3] = (lo, hi, (lo + hi) // 2)
    col = np.arange(stride)[None, :]
    pad = col >= lens[:, None]
    q[pad] = rng.integers(0, 256, int(pad.sum()), dtype=np.uint8)            # padding: anything
    hit = (rng.random((n, stride)) < 0.002) & ~pad
    q[hit] = rng.choice(np.array([0, 255], np.uint8), int(hit.sum()))
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8)
    for rows in (2, 4):
        e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows)
        assert path["narrow_rows"] == rows
        assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps), (stride, rows)


@pytest.mark.parametrize("kw", [dict(ambigs="treat_as_errors"), dict(ambigs="ignore"), dict(ambigs="disallow"),
                                dict(ambigs="treat_as_errors", round_=True), dict(ambigs="ignore", maxerrors=0.4),
                                dict(alpha=0.05, uncert=0.002), dict(alpha=0.3), dict(alpha=1e-4), dict(alpha=0.9)])
def test_modes(eng, oracle, kw):
    """--ambigs / --round / --maxerrors / alpha and the per-read limit len * uncert (moira.py:949-950) in the pass' epilogue."""
    n = 20_000
    q, lens = oracle.synth_fill(n, 640, min_len=50, max_len=600, seed=21, profile=1)
    q[::7, 5] = 0
    q[::11, 17] = 255
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8, **kw)
    for rows in (2, 3):
        e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows, **kw)
        assert path["narrow_rows"] == rows
        assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps), (kw, rows)
        assert c.n_pass == int(ps.sum())


def test_reference_vectors_as_they_are(eng):
    """The reference's own results (tests/golden/*.npz: bernoullimodule.c, and its Python twin where C is undefined): every set as
    ONE ragged batch, forced through the pass with 2, 3 and 4 rows -- no regrouping by length."""
    done = 0
    for name in G.NPZ_SETS:
        s = G.load_set(name)
        q, lens, exp = s["q"], s["lens"], G.expected_value(s)
        for rows in (2, 3, 4):
            e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows, alpha=float(s["alpha"]), ambigs="ignore")
            assert path["narrow_rows"] == (rows if q.shape[1] <= 4096 else 0), (name, q.shape)
            assert same(e1, exp), (name, rows)
            assert np.array_equal(n1, s["ns_ref"])
        done += len(lens)
    assert done > 10_000


@pytest.mark.parametrize("rows,split", [(3, 4), (3, 12), (3, 20), (3, 30), (3, 38), (3, 255), (4, 10), (4, 25)])
@pytest.mark.parametrize("profile", [0, 1])
def test_mixed_rows(eng, oracle, rows, split, profile):
    """Mixed rows (k_narrow_rg<R, R - 1>): groups whose longest read has at most `split` chunks run with a row less.  Whatever the
    split -- below every read, in the middle of the length range, above every read -- the results are the oracle's: a read of a
    short group that needs the row it did not get is handed back like any other."""
    n = 40_000 + split
    q, lens = oracle.synth_fill(n, 640, min_len=50, max_len=600, seed=17, profile=profile)
    ee, ns, ps, need = oracle.filter_batch(q, lens=lens, threads=8)
    e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows, narrow_split=split)
    assert path["narrow_rows"] == rows and path["narrow_split"] == split
    assert same(e1, ee) and np.array_equal(n1, ns) and np.array_equal(p1, ps) and c.n_pass == int(ps.sum())
    # at least the reads that need more than `rows` rows are handed back, at most those that need more than rows - 1
    assert int((need > rows).sum()) <= path["n_fallback"] <= int((need > rows - 1).sum())
    if split == 255:
        assert path["n_fallback"] == int((need > rows - 1).sum())       # every group is a short one


def test_the_library_mixes_rows_by_itself_on_clean_ragged_reads(eng, oracle):
    """Clean reads of 50..600 bases need a third row from about 400 bases on: the library takes three rows, and two for the groups
    safely below the shortest sampled read that needs the third (mpb_path_info.narrow_split); nothing is handed back."""
    n, stride = 600_000, 640
    bufs = [eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n), eng.alloc(n * 4)]
    d_q, d_ee, d_ns, d_pass, d_len = bufs
    try:
        eng.synth_fill(d_q, n, stride, min_len=50, max_len=600, d_len=d_len, seed=33, profile=1)
        hq, hl = d_q.download(np.uint8, n * stride).reshape(n, stride), d_len.download(np.int32, n)
        ee, ns, ps, need = oracle.filter_batch(hq, lens=hl, threads=16)
        c = eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        path = eng.last_path()
        shortest3 = int(((hl[need >= 3] + 15) // 16).min())
        assert path["narrow_rows"] == 3 and 15 <= path["narrow_split"] < shortest3 and path["n_fallback"] <= n // 1000, (path, shortest3)
        assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_pass.download(np.uint8, n), ps) and c.n_pass == int(ps.sum())
    finally:
        for b in bufs:
            b.free()


def paired_golden_contigs(eng):
    recs = []
    for kind in ("good", "bad"):
        recs += G.read_fasta_qual(os.path.join(ROOT, "tests", "golden", "reference_test_results", "paired.qc." + kind))
    q, lens = eng.pack([r[2] for r in recs], [r[3] for r in recs], stride=512)
    return recs, q, lens


@pytest.mark.parametrize("rows", [2, 3, 4])
def test_the_references_paired_contigs(eng, oracle, rows):
    """The 400 representative contigs of the reference's paired golden run (241-502 bp; moira/test/test_results/paired.qc.*), tiled
    in a seeded random order to 100,000 rows: the ragged, high-quality batches the pass was built for."""
    recs, cq, cl = paired_golden_contigs(eng)
    rng = np.random.default_rng(3)
    idx = rng.permutation(np.arange(100_000) % len(recs))
    q, lens = cq[idx], cl[idx]
    ee, ns, ps, _ = oracle.filter_batch(cq, lens=cl, threads=8)
    e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=rows)
    assert path["narrow_rows"] == rows
    assert same(e1, ee[idx]) and np.array_equal(n1, ns[idx]) and np.array_equal(p1, ps[idx])


def test_lengths_outside_the_row_are_reported_not_obeyed(eng, oracle):
    """A negative length, or one beyond the row: the read gets ee = NaN, pass = 0 (by the sorted pipeline, which the pass hands
    it to) and the call that fetches counts fails -- exactly as without the pass."""
    n, stride = 10_000, 320
    q, lens = oracle.synth_fill(n, stride, min_len=50, max_len=300, seed=9, profile=1)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=4)
    lens = lens.copy()
    lens[[5, 4097, 9_999]] = (-1, 321, 1 << 30)
    with pytest.raises(ValueError, match="3 read length"):
        run_ragged(eng, q, lens, narrow_rows=2)
    e1, n1, p1, _, path = run_ragged(eng, q, lens, want_counts=False, narrow_rows=2)
    assert path["narrow_rows"] == 2 and path["n_fallback"] == 3
    ok = np.ones(n, bool)
    ok[[5, 4097, 9_999]] = False
    assert same(e1[ok], ee[ok]) and np.array_equal(p1[ok], ps[ok])
    assert np.isnan(e1[~ok]).all() and (p1[~ok] == 0).all()
    # the counter that call raised is reported by the next call that fetches counts (include/moira_pb.h), whatever pass it takes;
    # after that a batch with good lengths is clean again
    lens[[5, 4097, 9_999]] = 100
    ee2, ns2, ps2, _ = oracle.filter_batch(q, lens=lens, threads=4)
    with pytest.raises(ValueError, match="3 read length"):
        run_ragged(eng, q, lens, narrow_rows=2)
    e2, n2, p2, c, _ = run_ragged(eng, q, lens, narrow_rows=2)
    assert same(e2, ee2) and np.array_equal(p2, ps2) and c.n_pass == int(ps2.sum())


def test_rows_wider_than_the_pass_takes_go_through_the_sorted_pipeline(eng, oracle):
    n, stride = 3_000, 4112
    rng = np.random.default_rng(1)
    q = rng.integers(30, 41, (n, stride), dtype=np.uint8)
    lens = rng.integers(1, 4100, n).astype(np.int32)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8)
    e1, n1, p1, c, path = run_ragged(eng, q, lens, narrow_rows=2)
    assert path["narrow_rows"] == 0
    assert same(e1, ee) and np.array_equal(p1, ps)


def test_the_choice_on_ragged_batches(eng, oracle):
    """A clean ragged batch takes the pass by itself (from a sample of <= 0.1 % of its reads), BASELINE's model does not; the
    decision is reused; a fixed-length batch of the same shape is another batch; MPB_FLAG_NO_NARROW wins; results never depend."""
    nmax, stride = 400_000, 640
    d_q, d_ee, d_ns, d_pass, d_len = (eng.alloc(nmax * stride), eng.alloc(nmax * 8), eng.alloc(nmax * 4), eng.alloc(nmax),
                                      eng.alloc(nmax * 4))
    try:
        for profile, want_rows, n in ((1, 3, nmax), (0, 0, nmax - 10_000)):
            eng.synth_fill(d_q, n, stride, min_len=50, max_len=600, d_len=d_len, seed=31, profile=profile)
            hq = d_q.download(np.uint8, n * stride).reshape(n, stride)
            hl = d_len.download(np.int32, n)
            ee, ns, ps, _ = oracle.filter_batch(hq, lens=hl, threads=16)
            c = eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
            p1 = eng.last_path()
            assert p1["sampled"] and p1["narrow_rows"] == want_rows, p1
            assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_ns.download(np.int32, n), ns)
            assert np.array_equal(d_pass.download(np.uint8, n), ps) and c.n_pass == int(ps.sum())
            eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
            p2 = eng.last_path()
            assert not p2["sampled"] and p2["narrow_rows"] == want_rows
            assert same(d_ee.download(np.float64, n), ee)
            eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(no_narrow=True))
            assert eng.last_path()["narrow_rows"] == 0
            assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_pass.download(np.uint8, n), ps)
        # the same buffer read as a fixed-length batch: sampled again (the cached choice belongs to the ragged batch)
        eng.filter_device(d_q, nmax - 10_000, stride, fixed_len=600, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
        assert eng.last_path()["sampled"]
    finally:
        for b in (d_q, d_ee, d_ns, d_pass, d_len):
            b.free()


def test_high_quality_ragged_full_size(eng, oracle):
    """bench.py's extras.high_quality_ragged at its size: 5 M reads of U{50..600} bases of the clean profile in a stride-640
    matrix, resident, the pass chosen by the library; every read compared with the oracle."""
    from test_gpu_parity import compare_every_read
    n, stride, seed = 5_000_000, 640, 6
    d_q, d_ee, d_ns, d_pass, d_len = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n), eng.alloc(n * 4)
    try:
        eng.synth_fill(d_q, n, stride, min_len=50, max_len=600, d_len=d_len, seed=seed, profile=1)
        c1 = eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        path = eng.last_path()
        assert path["narrow_rows"] in (2, 3), path
        ee1, ns1, ps1 = d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n)
        assert compare_every_read(eng, oracle, d_q, n, stride, ee1, ns1, ps1, lens=d_len.download(np.int32, n),
                                  label="high_quality_ragged (narrow pass, R = %d)" % path["narrow_rows"]) == n
        assert c1.n_pass == int(ps1.sum())
    finally:
        for b in (d_q, d_ee, d_ns, d_pass, d_len):
            b.free()
