#!/usr/bin/env python3
"""k_dp's cost per DP cell against the read length (what is left of configs[4]'s per-base cost after the row-budget
growth and the masked lane-steps): fixed-length batches of BASELINE's quality model at L = 75 .. 600, the same number of bases each;
per batch the k_dp time (HIP events), the budget cells (class histogram x L) and ps per cell.  A fixed cost per read (tile set-up,
epilogue with its FP64 division, the hand-over of finished lanes) weighs more the shorter the read.
    python tools/dp_cost_by_length.py [bases = 1.5e9]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

bases = float(sys.argv[1]) if len(sys.argv) > 1 else 1.5e9
with Engine(0) as eng:
    print("%6s %9s %7s %9s %10s %12s %11s" % ("L", "reads", "stride", "dp_ms", "cells", "ps_per_cell", "ps_per_base"))
    for L in (75, 100, 150, 200, 300, 400, 600):
        n = int(bases / L)
        stride = (L + 63) // 64 * 64
        d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
        prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", no_narrow=True)
        run = lambda: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        for _ in range(20):
            run()
        eng.synchronize()
        hist = eng.class_histogram()
        eng.timing(True); eng.timing_reset()
        for _ in range(10):
            run()
        dp = eng.kernel_times()["dp"]
        eng.timing(False)
        ms = dp[0] / dp[1]
        cells = sum(c * k for c, k in hist.items()) * L
        print("%6d %9d %7d %9.3f %10.3e %12.4f %11.3f" % (L, n, stride, ms, cells, ms * 1e9 / cells, ms * 1e9 / (n * L)), flush=True)
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()
