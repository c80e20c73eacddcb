#!/usr/bin/env python3
"""Latency of the per-read drop-in entry (bernoulli.calculate_errors_PB) and of small batches."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "moira_amd", "dropin"))
import numpy as np  # noqa: E402
import bernoulli  # noqa: E402
from moira_amd.engine import default_engine  # noqa: E402

rng = np.random.default_rng(1)
seq = "".join(rng.choice(list("ACGT"), 300))
quals = [int(x) for x in np.clip(38 - (np.arange(300) / 300) ** 3 * 20 - rng.integers(0, 6, 300), 2, 40)]
for _ in range(20):
    bernoulli.calculate_errors_PB(seq, quals, 0.005)
t = time.perf_counter()
n = 2000
for _ in range(n):
    bernoulli.calculate_errors_PB(seq, quals, 0.005)
dt = time.perf_counter() - t
print("calculate_errors_PB: %.1f us per call = %.0f reads/s" % (dt / n * 1e6, n / dt))
eng = default_engine()
for m in (1, 64, 1024, 2048, 4096, 8192, 16384, 32768):
    q = np.tile(np.array(quals, np.uint8), (m, 1))
    q = np.pad(q, ((0, 0), (0, 20)))
    for _ in range(5):
        eng.filter(q, fixed_len=300)
    t = time.perf_counter()
    k = 200
    for _ in range(k):
        eng.filter(q, fixed_len=300)
    dt = (time.perf_counter() - t) / k
    print("Engine.filter batch of %5d: %.1f us per call = %.3e reads/s" % (m, dt * 1e6, m / dt))
