#!/usr/bin/env python3
"""Round 3: reads longer than 1023 bases, resident in HBM, HIP-event timed per kernel.

  a  good long reads (full-length 16S shape: 1,500 bases, Q25-40): few DP rows, ordinary tile classes on long rows
  b  the synthetic quality model at 50-2,000 bases (the bench line's extras.long_reads_ragged_50_2000)
  c  wide reads only: 4,096 bases at Q1-4 (about 2,900 rows: 3 waves of k_wide per read); cells/s against the
     FP64-VALU issue peak, which is the roof of the tile classes as well
  d  the longest read the path takes: 16,383 bases at Q2 (about 10,500 rows, 11 waves)

    python tools/long_read_rate.py > profiles/r03_long_reads.txt
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

FP64_PEAK = 39.3e12     # v_mul/add_f64 lane-ops per second, nominal (bench.py)


def timed(eng, label, fn, n, bases, reps=5, cells=None):
    fn(); eng.synchronize()
    eng.timing(True); eng.timing_reset()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    eng.synchronize()
    dt = (time.perf_counter() - t) / reps
    kt = {k: round(v[0] / reps, 3) for k, v in eng.kernel_times().items() if v[0]}
    eng.timing(False)
    extra = ""
    if cells:
        extra = "  %.3g DP cells x 3 ops / wide time = %.1f %% of the nominal FP64-VALU peak" % (
            cells, 100 * cells * 3 / (kt.get("wide", 0) * 1e-3 or 1e9) / FP64_PEAK)
    print("%s: %.3f ms per pass = %.3e reads/s = %.3e bases/s  kernels(ms) %s%s"
          % (label, dt * 1e3, n / dt, bases / dt, kt, extra), flush=True)


def resident(eng, q, lens):
    n, stride = q.shape
    bufs = [eng.alloc(q.nbytes).upload(q), eng.alloc(n * 4).upload(lens.astype(np.int32)),
            eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
    return bufs


def main():
    rng = np.random.default_rng(3)
    with Engine(0) as eng:
        prm = eng.params(no_narrow=True)      # (the row-budget histograms below describe the sorted pipeline)
        # a
        n, L = 200_000, 1500
        q = rng.integers(25, 41, (n, 1536), dtype=np.uint8)
        lens = np.full(n, L, np.int32)
        b = resident(eng, q, lens)
        timed(eng, "a  200 k x 1,500 bases, Q25-40 (tile classes, long rows)",
              lambda: eng.filter_device(b[0], n, 1536, d_len=b[1], d_ee=b[2], d_ns=b[3], d_pass=b[4], params=prm, want_counts=False),
              n, n * L)
        print("   row budgets:", {k: v for k, v in eng.class_histogram().items() if v})
        for x in b:
            x.free()
        # b
        n, stride = 1_000_000, 2048
        d_q, d_len = eng.alloc(n * stride), eng.alloc(n * 4)
        d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        eng.synth_fill(d_q, n, stride, fixed_len=0, min_len=50, max_len=2000, d_len=d_len, seed=7)
        ln = d_len.download(np.int32, n)
        timed(eng, "b  1 M synthetic reads, lengths U{50..2000}",
              lambda: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False),
              n, int(ln.sum()))
        h = eng.class_histogram()
        print("   row budgets:", {k: v for k, v in h.items() if v}, " reads in k_wide:", n - sum(h.values()))
        for x in (d_q, d_len, d_ee, d_ns, d_pass):
            x.free()
        # c
        n, L = 4096, 4096
        q = rng.integers(1, 4, (n, L), dtype=np.uint8)
        lens = np.full(n, L, np.int32)
        b = resident(eng, q, lens)
        c = eng.filter_device(b[0], n, L, d_len=b[1], d_ee=b[2], d_ns=b[3], d_pass=b[4], params=prm)
        ee = b[2].download(np.float64, n)
        waves = np.ceil((ee + 40) / 1024)          # row budget ~ ee + a few sigma
        cells = float((waves * 1024 * L).sum())
        timed(eng, "c  4,096 x 4,096 bases, Q1-3 (every read in k_wide; ee %.0f..%.0f)" % (ee.min(), ee.max()),
              lambda: eng.filter_device(b[0], n, L, d_len=b[1], d_ee=b[2], d_ns=b[3], d_pass=b[4], params=prm, want_counts=False),
              n, n * L, cells=cells)
        print("   overflow re-runs:", c.n_overflow)
        for x in b:
            x.free()
        # d
        n, L = 512, 16383
        q = np.full((n, 16384), 2, np.uint8)
        lens = np.full(n, L, np.int32)
        b = resident(eng, q, lens)
        eng.filter_device(b[0], n, 16384, d_len=b[1], d_ee=b[2], d_ns=b[3], d_pass=b[4], params=prm)
        ee = b[2].download(np.float64, n)
        cells = float((np.ceil((ee + 200) / 1024) * 1024 * L).sum())
        timed(eng, "d  512 x 16,383 bases, Q2 (ee %.0f: 11 waves per read)" % ee[0],
              lambda: eng.filter_device(b[0], n, 16384, d_len=b[1], d_ee=b[2], d_ns=b[3], d_pass=b[4], params=prm, want_counts=False),
              n, n * L, reps=3, cells=cells)
        for x in b:
            x.free()


if __name__ == "__main__":
    main()
