#!/usr/bin/env python3
"""Run ON THE GPU BOX under `rocprofv3 --kernel-trace --stats`: N single-read calls of one length through Engine.filter
(k_small): the kernel's duration against the read's length = launch-independent cost + cost per base.
  python3 tools/experiments/small_probe.py LENGTH [CALLS] [ROWS: bad|good]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import default_engine  # noqa: E402

L = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
kind = sys.argv[3] if len(sys.argv) > 3 else "good"
rng = np.random.default_rng(1)
quals = np.clip(38 - (np.arange(L) / L) ** 3 * 20 - rng.integers(0, 6, L), 2, 40).astype(np.uint8)
if kind == "bad":
    quals = rng.integers(8, 21, L).astype(np.uint8)
q = np.pad(quals[None, :], ((0, 0), (0, (-L) % 16 + 16)))
eng = default_engine()
for _ in range(20):
    eng.filter(q, fixed_len=L)
t = time.perf_counter()
for _ in range(n):
    eng.filter(q, fixed_len=L)
dt = (time.perf_counter() - t) / n
print("L = %d (%s): %.1f us per call" % (L, kind, dt * 1e6))
