#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry (mpb_filter_host): packed reads in host memory in,
ee/Ns/pass in host memory out, through the three-slot pipeline (H2D | kernels | D2H overlapped).
Two sources: an ordinary (pageable) numpy matrix, staged into pinned memory by the library, and a pinned
matrix from mpb_host_alloc, DMA-ed where it lies.  Reported in DESIGN.md and bench.py's `extras.host_fed`
beside (never instead of) the HBM-resident `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
stride, L = 320, 300
with Engine(0) as eng:
    d = eng.alloc(n * stride)
    eng.synth_fill(d, n, stride, fixed_len=L, seed=2)
    host = d.download(np.uint8, n * stride).reshape(n, stride)
    d.free()
    pin = eng.host_alloc((n, stride), np.uint8)
    pin[:] = host
    eng.filter(host[:200000], fixed_len=L)
    for name, q in (("pageable", host), ("pinned", pin)):
        for rep in range(4):
            t = time.perf_counter()
            r = eng.filter(q, fixed_len=L)
            dt = time.perf_counter() - t
            print("host path (%s source): %d reads in %.1f ms = %.3e reads/s (%.2f GB/s of qscores over PCIe), pass=%d"
                  % (name, n, dt * 1e3, n / dt, n * stride / dt / 1e9, r.n_pass), flush=True)
    eng.host_free(pin)
