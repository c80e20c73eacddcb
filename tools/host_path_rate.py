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
    out = (np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.uint8))     # reused result arrays, pages already present
    for name, q in (("pageable", host), ("pinned", pin)):
        for rep in range(4):
            t = time.perf_counter()
            r = eng.filter(q, fixed_len=L, out=out)
            dt = time.perf_counter() - t
            print("host path (%s source): %d reads in %.1f ms = %.3e reads/s (%.2f GB/s of qscores over PCIe), pass=%d"
                  % (name, n, dt * 1e3, n / dt, n * stride / dt / 1e9, r.n_pass), flush=True)
    eng.host_free(pin)

# where the time goes: the C call alone with outputs that already have their pages (no first-touch faults), from pinned memory
import ctypes as C  # noqa: E402
from moira_amd import _lib as L  # noqa: E402
with Engine(0) as eng:
    pin = eng.host_alloc((n, stride), np.uint8)
    pin[:] = host
    ee, ns, ps = np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.uint8)
    pee, pns, pps = eng.host_alloc(n, np.float64), eng.host_alloc(n, np.int32), eng.host_alloc(n, np.uint8)
    prm = eng.params()
    cnt = L.FilterCounts()
    for name, (a, b, c) in (("touched pageable outputs", (ee, ns, ps)), ("pinned outputs", (pee, pns, pps))):
        for rep in range(3):
            t = time.perf_counter()
            L.check(eng.lib.mpb_filter_host(eng.ctx, pin.ctypes.data, n, stride, None, L_ := 300, C.byref(prm),
                                            a.ctypes.data, b.ctypes.data, c.ctypes.data, C.byref(cnt)))
            dt = time.perf_counter() - t
            print("C call only, pinned source, %s: %.1f ms = %.2f GB/s, pass=%d" % (name, dt * 1e3, n * stride / dt / 1e9, cnt.n_pass), flush=True)
