#!/usr/bin/env python3
"""--error_calc poisson (SURVEY f-3) on a resident batch: the device part (k_lambda: per-read in-order sum of
error probabilities, bit-identical to the reference's sequential sum) timed with HIP events, and the host tail
(mpb_poisson_finish_host: libm exp / pow per CDF term, split over the granted CPUs) timed on the wall clock."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402
from moira_amd import _lib as L  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 10_000_000
stride, Lr = 320, 300
with Engine(0) as eng:
    d_q, d_lam, d_ns = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4)
    eng.synth_fill(d_q, n, stride, fixed_len=Lr, seed=2)
    # the synthetic model has no lower-case n (byte 255), which the Poisson entry rejects
    for _ in range(2):
        L.check(eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, None, Lr, d_lam.ptr, d_ns.ptr))
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        L.check(eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, None, Lr, d_lam.ptr, d_ns.ptr))
    ms, cnt = eng.kernel_times()["lambda"]
    eng.timing(False)
    ms /= cnt
    print("k_lambda: %.3f ms per %d reads = %.3e reads/s = %.2f TB/s of qualities (%.1f%% of the 8 TB/s roof)"
          % (ms, n, n / ms * 1e3, n * stride / ms / 1e9, 100 * n * (Lr + 12) / (ms * 1e-3) / 8e12))
    if "--device-only" in sys.argv:
        sys.exit(0)
    lam, ns = d_lam.download(np.float64, n), d_ns.download(np.int32, n)
    ee, ps = np.empty(n), np.empty(n, np.uint8)
    prm = eng.params()
    for rep in range(2):
        t = time.perf_counter()
        L.check(eng.lib.mpb_poisson_finish_host(lam.ctypes.data, ns.ctypes.data, None, Lr, n, C.byref(prm), ee.ctypes.data, ps.ctypes.data))
        dt = time.perf_counter() - t
        print("host tail: %.1f ms per %d reads = %.3e reads/s (pass %d)" % (dt * 1e3, n, n / dt, int(ps.sum())), flush=True)
