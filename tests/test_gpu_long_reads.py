"""Reads longer than 1023 bases through the Poisson-binomial HIP path.

The reference's Python twin scores any length (moira/moira.py:1561-1634; `--error_calc poisson_binomial_py`,
moira.py:820-821); its C extension overruns the stack near 1000 bases.  Two things had to go: the row-length limit
of the prepass / one-read-per-wave kernels (a long, good read needs few DP rows and runs in the ordinary tile
classes), and the 1024-row limit of the DP (a long, bad read needs more rows than one wave holds: k_wide, up to
16 waves x 1024 rows).  Everything is compared bit for bit with the oracle's two-term recurrence, which is pinned
to the reference (tests/test_oracle_golden.py, incl. the long-read vectors of tests/golden/long_reads.npz)."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module", params=["batched", "small"])
def eng(request):
    from moira_amd.engine import Engine
    e = Engine(0)
    e.batched_only = request.param == "batched"
    yield e
    e.close()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _batch(specs, stride, seed):
    """specs: (length, q_lo, q_hi_exclusive, copies) -> (q, lens)"""
    rng = np.random.default_rng(seed)
    rows, lens = [], []
    for L, lo, hi, copies in specs:
        for _ in range(copies):
            rows.append(rng.integers(lo, hi, L).astype(np.uint8))
            lens.append(L)
    q = np.zeros((len(lens), stride), np.uint8)
    for i, r in enumerate(rows):
        q[i, :len(r)] = r
    return q, np.array(lens, np.int32)


LONG_SPECS = [
    # good long reads: few rows, ordinary tile classes on long rows
    (1024, 30, 41, 4), (1025, 30, 41, 4), (1500, 25, 41, 4), (2500, 20, 41, 3), (4096, 30, 41, 3),
    # middling: hundreds of rows, the wide tile classes (G = 16..64) on long rows
    (1500, 8, 20, 3), (2500, 10, 25, 3), (4096, 12, 30, 2), (4096, 7, 12, 2),
    # bad long reads: more than 1024 rows -> k_wide (2, 3, 4 ... waves)
    (1500, 1, 3, 3), (2500, 1, 4, 3), (2500, 2, 6, 2), (4096, 1, 3, 2), (4096, 3, 6, 2),
    # short reads in the same batch
    (300, 2, 41, 6), (97, 1, 3, 3), (1023, 1, 2, 2), (50, 40, 41, 2),
]


def test_long_reads_match_oracle(eng, oracle):
    q, lens = _batch(LONG_SPECS, 4096, 7)
    q[2, 100] = 0; q[2, 1000] = 255           # ambiguous bases beyond column 960 (second prepass panel)
    q[30, 2000:2040] = 0                      # a run of N in a wide read
    q[33, 17] = 255
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    assert rows.max() > 2500 and (rows > 1024).sum() >= 8 and rows.min() <= 3
    r = eng.filter(q, lens=lens)
    bad = np.nonzero(~((r.ee == ee) | (np.isnan(r.ee) & np.isnan(ee))))[0]
    assert len(bad) == 0, [(int(i), int(lens[i]), int(rows[i]), r.ee[i], ee[i]) for i in bad[:8]]
    assert np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    for kw in (dict(ambigs="ignore", alpha=0.05), dict(ambigs="disallow", round_=True), dict(maxerrors=900.0, alpha=1e-4)):
        e2, n2, p2, _ = oracle.filter_batch(q, lens=lens, threads=8, **kw)
        r2 = eng.filter(q, lens=lens, **kw)
        assert same(r2.ee, e2) and np.array_equal(r2.ns, n2) and np.array_equal(r2.passed, p2.astype(bool)), kw


def test_len_equal_to_stride_1024(eng, oracle):
    """A read of exactly 1024 bases in a stride-1024 batch used to fail the call (rounds 1-2)."""
    q, lens = _batch([(1024, 30, 41, 3), (1024, 1, 3, 2), (1024, 10, 20, 2), (1000, 2, 41, 3)], 1024, 11)
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    r = eng.filter(q, lens=lens)
    assert same(r.ee, ee) and np.array_equal(r.passed, ps.astype(bool))
    qf = q[:7]
    ee, ns, ps, rows = oracle.filter_batch(qf, fixed_len=1024, threads=8)
    r = eng.filter(qf, fixed_len=1024)
    assert same(r.ee, ee) and np.array_equal(r.passed, ps.astype(bool))


def test_longest_supported_read(eng, oracle):
    """16383 bases: good (a handful of rows), terrible (about 10,000 rows: 11 waves of k_wide) and all-Q1
    (about 13,100 rows)."""
    q, lens = _batch([(16383, 30, 41, 2), (16383, 2, 3, 1), (16383, 1, 2, 1), (16383, 5, 30, 1), (9000, 1, 5, 2)], 16384, 13)
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    assert rows.max() > 12000
    r = eng.filter(q, lens=lens)
    assert same(r.ee, ee), (r.ee, ee, rows)
    assert np.array_equal(r.passed, ps.astype(bool))
    assert not np.isnan(r.ee).any()


def test_wide_reads_underpredicted_take_the_final_pass(eng, oracle):
    """MPB_FLAG_TEST_UNDERPREDICT halves every row budget: wide reads then miss theirs in the main pass and are
    re-run by k_wide with len + 1 rows, short reads of the batch too; results must not change."""
    q, lens = _batch([(2500, 1, 4, 4), (4096, 2, 5, 3), (1500, 1, 3, 3), (300, 5, 41, 20), (2000, 20, 41, 4)], 4096, 17)
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    r = eng.filter(q, lens=lens, test_underpredict=True, batched_only=True)
    assert r.n_overflow >= 10
    assert same(r.ee, ee) and np.array_equal(r.passed, ps.astype(bool))


def test_fixed_length_long_batch(eng, oracle):
    """Fixed-length batches of long reads (no length array): 6000 reads of 2560 bases, mixed quality, so that the
    prepass' panel loop, the tile classes on long rows and k_wide all see a real batch."""
    n, L = 6000, 2560
    rng = np.random.default_rng(19)
    q = np.zeros((n, L), np.uint8)
    lo = rng.integers(1, 35, n)
    for i in range(n):
        q[i] = rng.integers(lo[i], lo[i] + 6, L)
    q[rng.integers(0, n, 200), rng.integers(0, L, 200)] = 0
    ee, ns, ps, rows = oracle.filter_batch(q, fixed_len=L, threads=8)
    assert (rows > 1024).sum() > 100 and (rows < 20).sum() > 1000
    r = eng.filter(q, fixed_len=L)
    bad = np.nonzero(r.ee != ee)[0]
    assert len(bad) == 0, [(int(i), int(rows[i]), r.ee[i], ee[i]) for i in bad[:8]]
    assert np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
    assert r.n_pass == int(ps.sum())


def test_per_read_entry_long(eng, oracle):
    """bernoulli.calculate_errors_PB on long reads (what an unchanged moira.py would call per read)."""
    rng = np.random.default_rng(23)
    for L, lo, hi in ((1024, 30, 41), (1500, 20, 41), (3000, 1, 4), (5000, 10, 41)):
        quals = [int(x) for x in rng.integers(lo, hi, L)]
        seq = "".join(rng.choice(list("ACGTN"), L, p=[0.2495, 0.2495, 0.2495, 0.2495, 0.002]))
        want = oracle.ee_rowwise(seq, quals, 0.005)
        assert eng.calculate_errors_PB(seq, quals, 0.005) == (want[0], want[1])


def test_reads_beyond_16383_bases(eng, oracle):
    """Round 4: the length limit is 65535 bases (rounds 1-3: 16383, because the fallback had to cover len + 1 rows).  What a
    read may NEED is still 16384 DP rows: good and middling long reads (a handful to a few thousand rows: tile classes on
    long rows, k_wide) are bit-exact against the oracle, with ambiguous bases and in a batch with short reads; a read that
    needs more rows than 16 waves hold has no result (NaN, rejected) -- never a wrong one."""
    q, lens = _batch([(20000, 30, 41, 2), (40000, 25, 41, 1), (65535, 33, 41, 1), (30000, 9, 14, 1), (50000, 12, 20, 1),
                      (65535, 10, 12, 1), (300, 2, 41, 3), (1500, 1, 3, 1), (24000, 1, 2, 1), (65535, 2, 3, 1)], 65536, 29)
    q[0, 19000] = 0; q[0, 150] = 255; q[3, 29999] = 0; q[5, 40000:40100] = 0
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=8)
    need = rows > 16384
    assert need.sum() == 2 and rows[~need].max() > 6000 and (rows < 40).sum() >= 5     # Q1 x 24000 and Q2 x 65535 need too much
    r = eng.filter(q, lens=lens, batched_only=True)
    assert same(r.ee[~need], ee[~need]) and np.array_equal(r.ns, ns)
    assert np.array_equal(r.passed[~need], ps[~need].astype(bool))
    assert np.isnan(r.ee[need]).all() and not r.passed[need].any()
    assert not np.isnan(r.ee[~need]).any()
    # the per-read entry takes them too (through the pipeline: the one-read-per-wave kernel stops at 16384-byte rows)
    seq = "".join("N" if v == 0 else "n" if v == 255 else "A" for v in q[0, :lens[0]])
    quals = [int(v) if v not in (0, 255) else 30 for v in q[0, :lens[0]]]
    assert eng.calculate_errors_PB(seq, quals, 0.005) == oracle.ee_rowwise(seq, quals, 0.005)[:2]


def test_reads_with_more_than_32767_lower_case_n(eng, oracle):
    """the ambiguity counters are 'N' | 'n' << 16 in one word; unpacked with a signed shift, 32768 or more
    lower-case n made the count negative and the result silently wrong.  A 40,000-base read of 'n' only, one that mixes 35,000
    'n' with 20,000 'N' and scored bases, and a 65,535-base read of 'n' only."""
    q = np.zeros((4, 65536), np.uint8)
    lens = np.array([40000, 60000, 65535, 300], np.int32)
    q[0, :40000] = 255
    rng = np.random.default_rng(5)
    q[1, :60000] = rng.integers(25, 41, 60000)
    pos = rng.permutation(60000)
    q[1, pos[:35000]] = 255
    q[1, pos[35000:55000]] = 0
    q[2, :65535] = 255
    q[3, :300] = 30
    ee, ns, ps, rows = oracle.filter_batch(q, lens=lens, threads=4)
    assert list(ns) == [40000, 55000, 65535, 0]
    for kw in (dict(ambigs="treat_as_errors"), dict(ambigs="ignore"), dict(ambigs="disallow")):
        e2, n2, p2, _ = oracle.filter_batch(q, lens=lens, threads=4, **kw)
        r = eng.filter(q, lens=lens, batched_only=True, **kw)
        assert same(r.ee, e2) and np.array_equal(r.ns, n2) and np.array_equal(r.passed, p2.astype(bool)), kw
    r = eng.filter(q[:, :16384], lens=np.minimum(lens, 16384))          # and through the one-read-per-wave kernel
    e3, n3, p3, _ = oracle.filter_batch(q[:, :16384], lens=np.minimum(lens, 16384), threads=4)
    assert same(r.ee, e3) and np.array_equal(r.ns, n3)


def test_too_long_is_refused_not_truncated(eng):
    from moira_amd import _lib as L
    prm = eng.params()
    q = np.full((2, 65600), 30, np.uint8)
    ee, ns, ps = np.zeros(2), np.zeros(2, np.int32), np.zeros(2, np.uint8)
    rc = eng.lib.mpb_filter_host(eng.ctx, q.ctypes.data, 2, 65600 // 16 * 16, None, 65590, C.byref(prm),
                                 ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, None)
    assert rc == L.E_INVALID
    with pytest.raises(ValueError, match="65535"):
        eng.calculate_errors_PB("A" * 65536, [30] * 65536, 0.005)
