#!/usr/bin/env python3
"""Overflow rate of the row-budget prediction on quality profiles far from the smooth synthetic model:
excellent reads with a few terrible bases, bimodal reads, short reads, all-low reads."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

rng = np.random.default_rng(5)
n, L = 400_000, 300
cases = {}
for k in (1, 2, 3, 5, 8, 15):
    q = np.full((n, 320), 40, np.uint8)
    for _ in range(k):
        q[np.arange(n), rng.integers(0, L, n)] = rng.choice([1, 2, 3, 5], n)
    cases["Q40 + %d terrible bases" % k] = q
q = np.where(rng.random((n, 320)) < 0.1, rng.integers(1, 6, (n, 320)), 38).astype(np.uint8)
cases["10 % bases Q1-5, rest Q38"] = q
q = np.where(rng.random((n, 1)) < 0.5, rng.integers(30, 41, (n, 320)), rng.integers(2, 12, (n, 320))).astype(np.uint8)
cases["half the reads Q30-40, half Q2-11"] = q
cases["uniform Q1-40"] = rng.integers(1, 41, (n, 320)).astype(np.uint8)
cases["Q20 flat"] = np.full((n, 320), 20, np.uint8)
with Engine(0) as eng:
    eng.batched_only = True
    for name, q in cases.items():
        for alpha in (0.005, 0.05):
            r = eng.filter(q, fixed_len=L, alpha=alpha)
            t = time.perf_counter()
            eng.filter(q, fixed_len=L, alpha=alpha)
            dt = time.perf_counter() - t
            print("%-36s alpha %-5g overflow %7d (%.3f %%)  %.1f ms host-path  ee mean %.2f" %
                  (name, alpha, r.n_overflow, 100.0 * r.n_overflow / n, dt * 1e3, float(np.nanmean(r.ee))), flush=True)
