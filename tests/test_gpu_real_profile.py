"""bench.py's extras.real_profile: the reference's own reads -- moira/test/test1.fastq and the contigs of its paired
golden run -- tiled in a seeded random order to resident batches.  Every row of the tiled batch must carry the oracle's result
for the unique read it is a copy of."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_tiled_reference_reads_equal_the_oracle_on_the_unique_rows(oracle):
    import bench
    from moira_amd.engine import Engine
    with Engine(0) as eng:
        batches = bench.real_profile_batches(eng, rows_fixed=1_000_003, rows_ragged=500_001)
        assert [len(b[5]) for b in batches] == [1000, 400]
        for label, q, lens, fixed_len, idx, uq, ul in batches:
            n, stride = q.shape
            # copies of one read are not adjacent (a seeded permutation of the tiling)
            assert (idx[1:] == idx[:-1]).mean() < 0.01 and len(np.unique(idx)) == len(uq)
            if lens is None:
                ee, ns, ps, rows = oracle.filter_batch(uq, fixed_len=fixed_len, threads=8)
            else:
                ee, ns, ps, rows = oracle.filter_batch(uq, lens=ul, threads=8)
            bufs = [eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)] + ([eng.alloc(n * 4)] if lens is not None else [])
            try:
                bufs[0].upload(q)
                if lens is not None:
                    bufs[4].upload(lens)
                for kw in (dict(), dict(no_narrow=True)) + ((dict(narrow_rows=3),) if lens is None else ()):
                    c = eng.filter_device(bufs[0], n, stride, d_len=bufs[4] if lens is not None else None, fixed_len=fixed_len,
                                          d_ee=bufs[1], d_ns=bufs[2], d_pass=bufs[3], params=eng.params(**kw))
                    assert np.array_equal(bufs[1].download(np.float64, n), ee[idx], equal_nan=True), (label, kw)
                    assert np.array_equal(bufs[2].download(np.int32, n), ns[idx]) and np.array_equal(bufs[3].download(np.uint8, n), ps[idx])
                    assert c.n_pass == int(ps[idx].sum())
            finally:
                for b in bufs:
                    b.free()
            print("%s: %d rows, rows needed: median %d, max %d, %.1f %% above 64" %
                  (label, n, int(np.median(rows)), int(rows.max()), 100.0 * (rows > 64).mean()))
