#!/usr/bin/env python3
"""BASELINE config 5: ragged reads, lengths ~ U{50..600}, resident in HBM.  One wide matrix (stride 608)
against the length-bucketed layout (buckets of ceil(L/64)*64, one launch sequence per bucket).
Kernel-resident rates, HIP-event timed per kernel."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 5_000_000
stride, lo, hi = 608, 50, 600
with Engine(0) as eng:
    d_q, d_len = eng.alloc(n * stride), eng.alloc(n * 4)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=0, min_len=lo, max_len=hi, d_len=d_len, seed=5)
    prm = eng.params()

    def run(label, fn, reps=5):
        fn(); eng.synchronize()
        eng.timing(True); eng.timing_reset()
        t = time.perf_counter()
        for _ in range(reps):
            fn()
        eng.synchronize()
        dt = (time.perf_counter() - t) / reps
        kt = {k: round(v[0] / reps, 3) for k, v in eng.kernel_times().items() if v[0]}
        eng.timing(False)
        print("%s: %.3f ms per pass = %.3e reads/s  %s" % (label, dt * 1e3, n / dt, kt), flush=True)

    run("one matrix, stride 608", lambda: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns,
                                                            d_pass=d_pass, params=prm, want_counts=False))
    # bucketed: re-pack on the host once (not timed), keep every bucket resident
    lens = d_len.download(np.int32, n)
    q = d_q.download(np.uint8, n * stride).reshape(n, stride)
    b = np.maximum((lens + 63) // 64 * 64, 64)
    buckets = []
    for s in np.unique(b):
        idx = np.nonzero(b == s)[0]
        sub = np.ascontiguousarray(q[idx, :int(s)])
        m = len(idx)
        bq, bl = eng.alloc(m * int(s)).upload(sub), eng.alloc(m * 4).upload(lens[idx])
        buckets.append((int(s), m, bq, bl, eng.alloc(m * 8), eng.alloc(m * 4), eng.alloc(m)))

    def bucketed():
        for s, m, bq, bl, e, ns_, p in buckets:
            eng.filter_device(bq, m, s, d_len=bl, d_ee=e, d_ns=ns_, d_pass=p, params=prm, want_counts=False)
    if "--no-buckets" not in sys.argv:
        run("length-bucketed, %d buckets" % len(buckets), bucketed)
