"""Parity of the HIP path (through the C ABI) against the oracle and the reference's fixtures.
Integer outputs and decisions: identical.  ee: bit-exact in the default mode (stronger than the
1e-9 relative tolerance BASELINE.json's north_star allows); within 1e-9 relative in FAST_FMA mode."""
import os
import sys

import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu

REL_TOL = 1e-9      # north_star: "within 1e-9 relative"


@pytest.fixture(scope="module", params=["batched", "small"])
def eng(request):
    """Every parity test runs twice: through the sorted, tiled pipeline whatever the batch size, and with
    mpb_filter_host free to send batches of <= 4096 reads through the one-read-per-wave launch."""
    from moira_amd.engine import Engine
    e = Engine(0)
    e.batched_only = request.param == "batched"
    yield e
    e.close()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def test_kats_through_dropin_module(eng):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "moira_amd", "dropin"))
    import bernoulli
    kat = G.load_kat()
    k = kat["kat1"]
    assert bernoulli.calculate_errors_PB(k["seq"], k["quals"], k["alpha"]) == (6.446879136706666, 0)
    assert bernoulli.calculate_errors(k["seq"], k["quals"], k["alpha"]) == (6.446879136706666, 0)
    for name, want in (("kat2_forward_truncate200", 0.9685179556745876),
                       ("kat3_paired_truncate200", 0.9643903629780557)):
        k = kat[name]
        ee, ns = bernoulli.calculate_errors_PB(k["seq"], k["quals"], k["alpha"])
        assert ee + ns == want
    s = G.load_set("synth250")
    ee, ns = bernoulli.calculate_errors_PB_batch(s["q"], lens=s["lens"], alpha=float(s["alpha"]))
    assert same(ee, G.expected_value(s)) and np.array_equal(ns, s["ns_ref"])


def test_dropin_error_behaviour(eng):
    f = eng.calculate_errors_PB
    with pytest.raises(ValueError, match="Alpha must be between 0 and 1"):
        f("ACGT", [30] * 4, 0.0)
    with pytest.raises(ValueError, match="Alpha must be between 0 and 1"):
        f("ACGT", [30] * 4, 1.0)
    with pytest.raises(ValueError, match="same length"):
        f("ACGT", [30] * 3, 0.005)
    with pytest.raises(TypeError):
        f("ACGT", (30, 30, 30, 30), 0.005)          # "O!" wants a list
    with pytest.raises(TypeError):
        f("ACGT", [30, 30.5, 30, 30], 0.005)
    with pytest.raises(ValueError):
        f("ACGT", [30, -2, 30, 30], 0.005)
    assert f("ACGT", [0] * 4, 0.005) == (3.987440567842452, 0)      # Q0 -> Q1
    assert f("", [], 0.005) == (0.0, 0)
    assert f("NNnN", [20] * 4, 0.005) == (0.0, 4)


def test_device_lut_and_box_libm_match_the_committed_fixture(eng):
    """SURVEY §8c last bullet: what this GPU box's libm produced and the context uploaded, against the hex-float
    table from the build container (tests/golden/kat.json).  A libm drift between the two machines fails this test by
    name instead of the vector sets below."""
    import math
    lut, probes = G.lut_fixture()
    a, b = eng.device_lut()
    for q, (p, am, bm) in lut.items():
        assert a[q] == am and b[q] == bm, q
    assert (a[0], b[0], a[255], b[255]) == (1.0, 0.0, 1.0, 0.0)
    for lam, j, e, pw in probes:                       # the Poisson tail's libm calls run on this box's host
        assert math.exp(-lam) == e, lam
        try:
            got = math.pow(lam, j)
        except OverflowError:
            got = math.inf
        assert got == pw, (lam, j)


@pytest.mark.parametrize("name", G.NPZ_SETS)
def test_reference_vectors_bit_exact(eng, name):
    s = G.load_set(name)
    r = eng.filter(s["q"], lens=s["lens"], alpha=float(s["alpha"]), ambigs="ignore")
    exp = G.expected_value(s)
    assert same(r.ee, exp), int((r.ee != exp).sum())
    assert np.array_equal(r.ns, s["ns_ref"])


@pytest.mark.parametrize("which", ["forward", "paired"])
def test_reference_golden_files_decisions(eng, golden_dir, which):
    base = os.path.join(golden_dir, "reference_test_results", which + ".qc.")
    for kind, want in (("good", True), ("bad", False)):
        recs = G.read_fasta_qual(base + kind)
        q, lens = eng.pack([r[2] for r in recs], [r[3] for r in recs])
        r = eng.filter(q, lens=lens, alpha=0.005, uncert=0.01, ambigs="treat_as_errors")
        assert np.all(r.passed == want), (which, kind, int((r.passed != want).sum()))


@pytest.mark.parametrize("kw", [dict(ambigs="treat_as_errors"), dict(ambigs="ignore"),
                                dict(ambigs="disallow"), dict(ambigs="treat_as_errors", round_=True),
                                dict(ambigs="ignore", maxerrors=2.5), dict(alpha=0.05, uncert=0.02),
                                dict(alpha=0.3), dict(alpha=1e-4)])
def test_modes_match_oracle(eng, oracle, kw):
    s = G.load_set("rand_mixed")
    q, lens = s["q"], s["lens"]
    r = eng.filter(q, lens=lens, **kw)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8, **kw)
    assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    assert r.n_pass == int(ps.sum())


def test_fixed_length_ignores_padding_garbage(eng, oracle):
    q, lens = oracle.synth_fill(3000, 320, fixed_len=300, seed=11)
    dirty = q.copy()
    dirty[:, 300:] = np.random.default_rng(1).integers(0, 256, (3000, 20), dtype=np.uint8)
    a = eng.filter(q, fixed_len=300)
    b = eng.filter(dirty, fixed_len=300)
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8)
    assert same(a.ee, ee) and same(b.ee, ee) and np.array_equal(b.ns, ns)
    assert np.array_equal(a.passed, ps.astype(bool))


def test_overflow_pass_is_exercised_and_exact(eng, oracle):
    q, lens = oracle.synth_fill(20000, 320, fixed_len=300, seed=3)
    ee, ns, ps, rows = oracle.filter_batch(q, fixed_len=300, threads=8)
    r = eng.filter(q, fixed_len=300, test_underpredict=True)
    assert r.n_overflow > 1000                       # the second pass really ran
    assert same(r.ee, ee) and np.array_equal(r.passed, ps.astype(bool))
    r2 = eng.filter(q, fixed_len=300)
    assert r2.n_overflow <= 20                       # and normally (almost) never does
    assert same(r2.ee, ee)


def test_algorithmic_cell_count_is_the_oracles(eng, oracle):
    """MPB_FLAG_COUNT_CELLS (bench.py's fp64_valu.frac_algorithmic): the device-side sum of sum_k min(k + 1, J) over the
    reads equals the same sum taken from the oracle's rows J; with and without the overflow pass; results unchanged."""
    def want(rows, lens, ns):
        J, Lp = rows.astype(np.int64), (lens - ns).astype(np.int64)
        return int(np.where(J <= Lp, J * (J + 1) // 2 + (Lp - J) * J, Lp * (Lp + 1) // 2).sum())
    for kw, gen in (({"fixed_len": 300}, dict(n=60000, stride=320, fixed_len=300, seed=2)),
                    ({}, dict(n=50000, stride=608, min_len=50, max_len=600, seed=5))):
        q, lens = oracle.synth_fill(**gen)
        ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
        n, stride = q.shape
        d_q, d_len = eng.alloc(q.nbytes).upload(q), eng.alloc(n * 4).upload(lens)
        d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        for under in (False, True):
            prm = eng.params(count_cells=True, test_underpredict=under)
            c = eng.filter_device(d_q, n, stride, d_len=None if kw else d_len, fixed_len=kw.get("fixed_len", 0),
                                  d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm)
            assert eng.algorithmic_cells() == want(rows, lens, ns)
            assert same(d_ee.download(np.float64, n), ee)
            assert (c.n_overflow > 100) == under
        eng.filter_device(d_q, n, stride, d_len=None if kw else d_len, fixed_len=kw.get("fixed_len", 0),
                          d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params())
        for b in (d_q, d_len, d_ee, d_ns, d_pass):
            b.free()


def test_wide_classes(eng, oracle):
    """Reads needing hundreds of DP rows (G > 1 classes, DPP row hand-over, staged CDF)."""
    rng = np.random.default_rng(5)
    rows_q, lens = [], []
    for L, lo, hi in ((600, 1, 4), (1000, 1, 3), (1023, 1, 2), (500, 2, 8), (350, 1, 12), (800, 3, 20),
                      (1023, 1, 40), (97, 1, 3), (64, 1, 2), (33, 1, 2), (1023, 30, 41)):
        for _ in range(6):
            rows_q.append(rng.integers(lo, hi, L).astype(np.uint8))
            lens.append(L)
    q = np.zeros((len(lens), 1024), np.uint8)
    for i, r in enumerate(rows_q):
        q[i, :len(r)] = r
    q[3, 17] = 0
    q[5, 100] = 255
    lens = np.array(lens, np.int32)
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    assert rows.max() > 600 and rows.min() < 10
    r = eng.filter(q, lens=lens)
    assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    if eng.batched_only:                 # the class table belongs to the pipeline; the small path has none
        hist = eng.class_histogram()
        assert sum(hist.values()) == len(lens) and sum(v for k, v in hist.items() if k >= 512) > 0


def test_empty_and_degenerate_batches(eng, oracle):
    r = eng.filter(np.zeros((0, 16), np.uint8), fixed_len=10)
    assert len(r.ee) == 0 and r.n_pass == 0
    q = np.zeros((5, 16), np.uint8)
    q[1, :3] = 40
    q[2, :16] = 2
    lens = np.array([0, 3, 16, 5, 1], np.int32)
    r = eng.filter(q, lens=lens)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens)
    assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    with pytest.raises(ValueError):
        eng.filter(q, lens=lens, alpha=1.5)
    with pytest.raises(ValueError):
        eng.filter(np.zeros((2, 24), np.uint8), fixed_len=10)          # stride not a multiple of 16


def test_ragged_config5_sample(eng, oracle):
    q, lens = oracle.synth_fill(30000, 608, min_len=50, max_len=600, seed=5)
    r = eng.filter(q, lens=lens)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8)
    assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))


def test_fast_fma_mode_within_tolerance(eng, oracle):
    q, lens = oracle.synth_fill(20000, 320, fixed_len=300, seed=4)
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8)
    r = eng.filter(q, fixed_len=300, fast_fma=True)
    rel = np.abs(r.ee - ee) / np.maximum(np.abs(ee), 1e-300)
    assert rel.max() <= REL_TOL, rel.max()
    assert np.array_equal(r.passed, ps.astype(bool))
    with pytest.raises(ValueError, match="alpha >= 1e-5"):         # the fma error grows like 1/alpha: refused below
        eng.filter(q[:100], fixed_len=300, fast_fma=True, alpha=1e-6)


def test_fast_fma_mode_keeps_decisions_exact_at_the_threshold(eng, oracle):
    """Thresholds placed exactly ON reads' expected errors (ee <= maxerrors is then decided by the last
    bit), and --round with thresholds on integers: the fma mode must still give the exact decisions."""
    q, lens = oracle.synth_fill(6000, 320, fixed_len=300, seed=8)
    ee0, _, _, _ = oracle.filter_batch(q, fixed_len=300, threads=8, ambigs="ignore")
    picks = [float(x) for x in np.unique(ee0[(ee0 > 0.5) & (ee0 < 60)])[::97][:12]]
    assert len(picks) >= 8
    for me in picks:
        for kw in (dict(maxerrors=me, ambigs="ignore"), dict(maxerrors=me, ambigs="ignore", round_=True)):
            ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8, **kw)
            r = eng.filter(q, fixed_len=300, fast_fma=True, **kw)
            assert np.array_equal(r.passed, ps.astype(bool)), (me, kw)
            rel = np.abs(r.ee - ee) / np.maximum(np.abs(ee), 1e-300)
            assert rel[ee > 0].max() <= REL_TOL
    # uncert thresholds: L * uncert hit exactly by construction
    for k in (10, 200, 999):
        u = float(ee0[k] / 300.0)
        if 0 < u <= 1:
            ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8, uncert=u, ambigs="ignore")
            r = eng.filter(q, fixed_len=300, fast_fma=True, uncert=u, ambigs="ignore")
            assert np.array_equal(r.passed, ps.astype(bool)), u


def test_device_synth_matches_host_and_device_resident_filter(eng, oracle):
    n, stride, L = 50000, 320, 300
    d_q = eng.alloc(n * stride)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2, first_read=1000)
    host_q, _ = oracle.synth_fill(n, stride, fixed_len=L, seed=2, first_read=1000)
    assert np.array_equal(d_q.download(np.uint8, n * stride).reshape(n, stride), host_q)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    ee, ns, ps, _ = oracle.filter_batch(host_q, fixed_len=L, threads=8)
    assert same(d_ee.download(np.float64, n), ee)
    assert np.array_equal(d_ns.download(np.int32, n), ns)
    assert np.array_equal(d_pass.download(np.uint8, n), ps)
    assert (c.n_reads, c.n_pass, c.n_fail) == (n, int(ps.sum()), n - int(ps.sum()))
    # ragged fill
    d_len = eng.alloc(n * 4)
    d_q2 = eng.alloc(n * 608)
    eng.synth_fill(d_q2, n, 608, min_len=50, max_len=600, d_len=d_len, seed=5)
    hq, hl = oracle.synth_fill(n, 608, min_len=50, max_len=600, seed=5)
    assert np.array_equal(d_len.download(np.int32, n), hl)
    assert np.array_equal(d_q2.download(np.uint8, n * 608).reshape(n, 608), hq)
    for b in (d_q, d_ee, d_ns, d_pass, d_len, d_q2):
        b.free()


def compare_every_read(eng, oracle, d_q, n, stride, ee1, ns1, ps1, fixed_len=None, lens=None, step=4_000_000, label="", row0=0):
    """Bit-for-bit comparison of ALL n reads of a resident batch with the oracle (no sampling).  The
    oracle reads the very bytes the GPU filtered: the matrix comes back from HBM a few million rows at a time (that the
    device generator writes what the host generator writes is checked on its own, by test_device_generator_* and by the
    host-regenerated windows of the callers).  Prints the compared-read count."""
    import time
    threads = oracle.lib().pbo_max_threads()
    t0, done = time.time(), 0
    for start in range(row0, row0 + n, step):              # rows row0 .. row0 + n - 1 of the resident batch (default: all of it)
        m = min(step, row0 + n - start)
        hq = d_q.download(np.uint8, m * stride, offset=start * stride).reshape(m, stride)
        if lens is None:
            ee, ns, ps, _ = oracle.filter_batch(hq, fixed_len=fixed_len, threads=threads)
        else:
            ee, ns, ps, _ = oracle.filter_batch(hq, lens=lens[start:start + m], threads=threads)
        sl = slice(start, start + m)
        assert same(ee1[sl], ee), (label, start)
        assert np.array_equal(ns1[sl], ns) and np.array_equal(ps1[sl], ps), (label, start)
        done += m
    from conftest import note_parity
    note_parity("[every-read parity] %s: %d of %d reads compared bit for bit with the oracle (%d threads, %.1f s)"
                % (label, done, n, threads, time.time() - t0))
    return done


def test_config2_full_size(eng, oracle):
    """BASELINE config 2: 10M x 300 bp resident in HBM.  Size-independent properties (determinism, pass count == sum
    of flags == threshold test on ee, no NaN) AND a bit-for-bit comparison of every one of the 10 M reads with the
    oracle (2 s of oracle on the 16 cores a GPU box grants), plus host-regenerated windows (the counter-based
    generator on the host writes what the device wrote)."""
    n, stride, L, seed = 10_000_000, 320, 300, 2
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed)
    c1 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    ee1 = d_ee.download(np.float64, n)
    ps1 = d_pass.download(np.uint8, n)
    c2 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    assert same(d_ee.download(np.float64, n), ee1) and (c1.n_pass, c1.n_overflow) == (c2.n_pass, c2.n_overflow)
    assert c1.n_pass == int(ps1.sum()) and c1.n_reads == n
    assert not np.isnan(ee1).any()
    assert np.array_equal(ps1.astype(bool), ee1 <= L * 0.01)
    ns1 = d_ns.download(np.int32, n)
    assert compare_every_read(eng, oracle, d_q, n, stride, ee1, ns1, ps1, fixed_len=L, label="config 2") == n
    rng = np.random.default_rng(99)
    for start in rng.integers(0, n - 64, 100):
        hq, _ = oracle.synth_fill(64, stride, fixed_len=L, seed=seed, first_read=int(start))
        assert np.array_equal(d_q.download(np.uint8, 64 * stride, offset=int(start) * stride).reshape(64, stride), hq)
    for b in (d_q, d_ee, d_ns, d_pass):
        b.free()


def test_ragged_config5_full_size(eng, oracle):
    """BASELINE config 5 at bench.py's size: 5 M reads of 50-600 bases in one stride-608 matrix (the batch of
    extras.ragged_config5), every read compared with the oracle."""
    n, stride, seed = 5_000_000, 608, 5
    d_q, d_len = eng.alloc(n * stride), eng.alloc(n * 4)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, min_len=50, max_len=600, d_len=d_len, seed=seed)
    c = eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    ee1, ns1, ps1 = d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n)
    lens = d_len.download(np.int32, n)
    assert lens.min() == 50 and lens.max() == 600 and c.n_pass == int(ps1.sum())
    assert np.array_equal(ps1.astype(bool), ee1 <= lens * 0.01)
    assert compare_every_read(eng, oracle, d_q, n, stride, ee1, ns1, ps1, lens=lens, step=2_000_000, label="config 5") == n
    hq, hl = oracle.synth_fill(4096, stride, min_len=50, max_len=600, seed=seed, first_read=n - 4096)
    assert np.array_equal(lens[n - 4096:], hl)
    assert np.array_equal(d_q.download(np.uint8, 4096 * stride, offset=(n - 4096) * stride).reshape(4096, stride), hq)
    for b in (d_q, d_len, d_ee, d_ns, d_pass):
        b.free()


def test_length_bucketed_layout_config5(eng, oracle):
    """BASELINE config 5: mixed 50-600 bp, length-bucketed (one padded matrix per 64-bp bucket)."""
    from moira_amd.buckets import filter_matrix_bucketed, filter_bucketed, bucket_of
    q, lens = oracle.synth_fill(40000, 608, min_len=50, max_len=600, seed=5)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8)
    e2, n2, p2 = filter_matrix_bucketed(eng, q, lens)
    assert same(e2, ee) and np.array_equal(n2, ns) and np.array_equal(p2, ps.astype(bool))
    assert set(np.unique(bucket_of(lens))) == set(range(64, 641, 64))
    # list-of-reads entry (what a parser hands over)
    sub = range(0, 600)
    seqs = ["".join("N" if v == 0 else "A" for v in q[i, :lens[i]]) for i in sub]
    quals = [[int(v) if v else 20 for v in q[i, :lens[i]]] for i in sub]
    e3, n3, p3 = filter_bucketed(eng, seqs, quals)
    assert same(e3, ee[:600]) and np.array_equal(n3, ns[:600]) and np.array_equal(p3, ps[:600].astype(bool))


def test_sharded_entry_single_rank(eng, oracle):
    from moira_amd.shard import filter_sharded, engine_filter_fn
    q, lens = oracle.synth_fill(5000, 256, fixed_len=250, seed=1)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=8)
    e, n_, p, totals = filter_sharded(q, lens, engine_filter_fn(eng), dist=None)
    assert same(e, ee) and np.array_equal(n_, ns) and np.array_equal(p, ps.astype(bool))
    assert totals == (int(ps.sum()), 5000 - int(ps.sum()))


def test_on_device_ascii_decode(eng, oracle):
    """SURVEY f-4: FASTQ quality bytes + base letters in HBM -> packed matrix == host packer."""
    rng = np.random.default_rng(12)
    n, stride = 3000, 160
    lens = rng.integers(0, 151, n).astype(np.int32)
    seq = rng.choice(np.frombuffer(b"ACGTNn", np.uint8), (n, stride), p=[.24, .24, .24, .24, .03, .01])
    qual = (rng.integers(0, 42, (n, stride)) + 33).astype(np.uint8)
    want = np.zeros((n, stride), np.uint8)
    for i in range(n):
        s = seq[i, :lens[i]].tobytes().decode()
        want[i] = oracle.pack_read(s, [int(v) - 33 for v in qual[i, :lens[i]]], stride)
    d_seq, d_qual, d_out = eng.alloc(n * stride).upload(seq), eng.alloc(n * stride).upload(qual), eng.alloc(n * stride)
    d_len, d_err = eng.alloc(n * 4).upload(lens), eng.alloc(4).upload(np.zeros(1, np.int32))
    eng.decode_ascii_device(d_seq, d_qual, n, stride, d_out, d_len=d_len, fastq_offset=33, d_err=d_err)
    assert np.array_equal(d_out.download(np.uint8, n * stride).reshape(n, stride), want)
    assert d_err.download(np.int32, 1)[0] == 0
    # and the decoded matrix feeds the filter directly, all on device
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.filter_device(d_out, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    ee, ns, ps, _ = oracle.filter_batch(want, lens=lens, threads=4)
    assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_pass.download(np.uint8, n), ps)
    q2, l2 = eng.pack_batch_ascii([seq[i, :lens[i]].tobytes().decode() for i in range(n)],
                                  [qual[i, :lens[i]].tobytes().decode("latin-1") for i in range(n)], stride=stride)
    assert np.array_equal(q2, want) and np.array_equal(l2, lens)
    for b in (d_seq, d_qual, d_out, d_len, d_err, d_ee, d_ns, d_pass):
        b.free()


def test_host_entry_chunks_large_batches(eng, oracle):
    """mpb_filter_host cuts a batch into chunks of <= 128 MiB of qualities that travel through three
    pinned/device slots (H2D of chunk k+1 | kernels of chunk k | D2H of chunk k-1): 1.2 M rows of stride
    1024 are ten chunks, with ragged lengths, from pageable and from pinned memory; every read is compared."""
    n, stride = 1_200_000, 1024
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 61, n).astype(np.int32)
    base, _ = oracle.synth_fill(n, 64, fixed_len=60, seed=21)
    base[np.arange(64)[None, :] >= lens[:, None]] = 0
    ee, ns, ps, _ = oracle.filter_batch(base, lens=lens, threads=oracle.lib().pbo_max_threads())
    pin = eng.host_alloc((n, stride), np.uint8)
    for kind in ("pageable", "pinned"):
        q = np.zeros((n, stride), np.uint8) if kind == "pageable" else pin
        q[:, 64:] = 0
        q[:, :64] = base
        r = eng.filter(q, lens=lens)
        assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool)), kind
        assert r.n_pass == int(ps.sum())
        del q
    eng.host_free(pin)


def test_host_entry_overlapped_pipeline_ragged_config5(eng, oracle):
    """The same pipeline on BASELINE config 5's shape (ragged 50-600 bp, stride 608): 1 M reads = five
    chunks, last chunk shorter than the others; overflow totals are summed over the chunks."""
    n = 1_000_003
    q, lens = oracle.synth_fill(n, 608, min_len=50, max_len=600, seed=5)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=oracle.lib().pbo_max_threads())
    r = eng.filter(q, lens=lens)
    assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    assert r.n_pass == int(ps.sum())
    r2 = eng.filter(q, lens=lens, test_underpredict=True)          # forces the second pass in every chunk
    assert same(r2.ee, ee) and r2.n_overflow > 10000


def test_lengths_are_validated_never_clamped(eng, oracle):
    """a length that does not fit its row used to be clamped silently (a 1024-base read in a stride-1024
    batch was scored on 1023 bases).  The reference scores every base it is given, so: the host entry fails before
    anything is computed; the device entry -- whose lengths live in HBM -- gives such a read ee = NaN, pass = 0
    (also when the caller never fetches the counts, so an asynchronous caller can never consume a
    result computed on a clamped length) and fails the call that does fetch them, once."""
    import ctypes as C
    from moira_amd import _lib as L
    q = np.full((3, 1008), 30, np.uint8)
    q[2, :] = 12
    e_ok, _, p_ok, _ = oracle.filter_batch(q, lens=np.array([100, 1008, 1008], np.int32))
    for bad in (1009, 20000, -1):
        lens = np.array([100, bad, 1008], np.int32)
        with pytest.raises(ValueError):
            eng.filter(q, lens=lens)
        ee, ns, ps = np.empty(3), np.empty(3, np.int32), np.empty(3, np.uint8)
        prm = eng.params()
        rc = eng.lib.mpb_filter_host(eng.ctx, q.ctypes.data, 3, 1008, lens.ctypes.data, 0, C.byref(prm),
                                     ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, None)
        assert rc == L.E_INVALID
        d_q, d_len = eng.alloc(q.nbytes).upload(q), eng.alloc(12).upload(lens)
        d_ee, d_ns, d_pass = eng.alloc(24), eng.alloc(12), eng.alloc(3)
        # asynchronous form (counts = NULL): MPB_OK, and the bad row carries NaN / 0 -- the others their results
        eng.filter_device(d_q, 3, 1008, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
        eng.synchronize()
        got_e, got_p = d_ee.download(np.float64, 3), d_pass.download(np.uint8, 3)
        assert np.isnan(got_e[1]) and got_p[1] == 0
        assert got_e[0] == e_ok[0] and got_e[2] == e_ok[2] and got_p[0] == p_ok[0] and got_p[2] == p_ok[2]
        # a host-entry call in between is not failed by the count that call left behind
        r = eng.filter(q, lens=np.array([100, 1008, 1008], np.int32))
        assert same(r.ee, e_ok)
        # synchronous form: the call reports it
        with pytest.raises(ValueError, match="outside"):
            eng.filter_device(d_q, 3, 1008, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        ok = np.array([100, 1008, 1008], np.int32)
        d_len.upload(ok)
        c = eng.filter_device(d_q, 3, 1008, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)   # the error does not stick
        assert c.n_reads == 3 and same(d_ee.download(np.float64, 3), e_ok)
        for b in (d_q, d_len, d_ee, d_ns, d_pass):
            b.free()
    with pytest.raises(ValueError):
        eng.filter(q, fixed_len=1009)


@pytest.mark.parametrize("kw", [dict(), dict(alpha=0.05, uncert=0.02), dict(alpha=0.3), dict(maxerrors=6.0, ambigs="ignore"),
                                dict(round_=True), dict(ambigs="disallow")])
def test_decision_only_mode_never_changes_a_decision(eng, oracle, kw):
    """MPB_FLAG_DECISION_ONLY: reads proven to fail are reported (pass=0, ee=+inf) without their DP;
    every flag must equal the full computation's, every computed ee must stay bit-exact."""
    q, lens = oracle.synth_fill(60000, 320, fixed_len=300, seed=6)
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8, **kw)
    r = eng.filter(q, fixed_len=300, decision_only=True, **kw)
    assert np.array_equal(r.passed, ps.astype(bool)) and np.array_equal(r.ns, ns)
    skipped = np.isinf(r.ee)
    assert not np.isnan(r.ee).any()
    assert skipped.sum() > 1000                       # the shortcut really fires on this workload
    assert not ps[skipped].any()                      # only failing reads are ever skipped
    assert same(r.ee[~skipped], ee[~skipped])
    s = G.load_set("rand_mixed")
    r2 = eng.filter(s["q"], lens=s["lens"], decision_only=True, **kw)
    e2, n2, p2, _ = oracle.filter_batch(s["q"], lens=s["lens"], threads=8, **kw)
    sk = np.isinf(r2.ee)
    assert np.array_equal(r2.passed, p2.astype(bool)) and same(r2.ee[~sk], e2[~sk]) and not p2[sk].any()


def test_two_contexts_from_two_threads(oracle):
    """One context per host thread on the same GPU: no shared mutable state between contexts."""
    import threading
    from moira_amd.engine import Engine
    qs = [oracle.synth_fill(30000, 320, fixed_len=300, seed=s)[0] for s in (31, 32)]
    want = [oracle.filter_batch(q, fixed_len=300, threads=4) for q in qs]
    got = [None, None]
    errs = []

    def work(k):
        try:
            for _ in range(6):           # fresh contexts: the first filter call of each also sets up its workspace
                with Engine(0) as e:
                    for _ in range(3):
                        got[k] = e.filter(qs[k], fixed_len=300)
                assert same(got[k].ee, want[k][0])
        except Exception as ex:      # pragma: no cover
            errs.append(ex)
    ts = [threading.Thread(target=work, args=(k,)) for k in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert not errs
    for k in (0, 1):
        assert same(got[k].ee, want[k][0]) and np.array_equal(got[k].passed, want[k][2].astype(bool))
