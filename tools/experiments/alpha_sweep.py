#!/usr/bin/env python3
"""Overflow rate and step time of the resident config-2 batch over alpha (the row-budget prediction is a
Cornish-Fisher quantile: how far does it hold?)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

n, L, stride = 4_000_000, 300, 320
with Engine(0) as eng:
    d_q = eng.alloc(n * stride)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
    for alpha in (1e-7, 1e-5, 1e-3, 0.005, 0.05, 0.3, 0.7, 0.95, 0.999):
        prm = eng.params(alpha=alpha)
        c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm)
        eng.synchronize()
        t = time.perf_counter()
        for _ in range(3):
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        eng.synchronize()
        dt = (time.perf_counter() - t) / 3
        print("alpha %-8g overflow %8d of %d (%.4f %%)  %.3f ms/step  pass %d" % (alpha, c.n_overflow, n, 100.0 * c.n_overflow / n, dt * 1e3, c.n_pass), flush=True)
