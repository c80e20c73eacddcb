#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (mpb_filter_host): packed reads in host memory in,
ee/Ns/pass in host memory out.  Reported in DESIGN.md beside (never instead of) bench.py's
HBM-resident `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import pb_oracle as O  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4_000_000
q, lens = O.synth_fill(n, 320, fixed_len=300, seed=2)
with Engine(0) as eng:
    eng.filter(q[:100000], fixed_len=300)
    for rep in range(3):
        t = time.perf_counter()
        r = eng.filter(q, fixed_len=300)
        dt = time.perf_counter() - t
        print("host path: %d reads in %.1f ms = %.3e reads/s (%.2f GB/s of qscores over PCIe), pass=%d"
              % (n, dt * 1e3, n / dt, n * 320 / dt / 1e9, r.n_pass), flush=True)
