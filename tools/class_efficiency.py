#!/usr/bin/env python3
"""Per-class efficiency of k_dp on REAL rows of the config-2 workload: the synthetic batch is classed once,
rows of one row-budget class at a time are gathered into a batch of their own (tiled up to a fixed size), and
k_dp is timed on it (HIP events).  FP64 issue efficiency = cells x 3 / time / nominal peak.  Shows which classes
sit below the FP64 roof (LUT-bound narrow bodies, DPP hand-over in wide ones, partial tiles)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

PEAK = 39.3e12
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 2_000_000
stride, L = 320, 300
with Engine(0) as eng:
    d_q = eng.alloc(n0 * stride)
    d_ee, d_ns, d_pass = eng.alloc(n0 * 8), eng.alloc(n0 * 4), eng.alloc(n0)
    eng.synth_fill(d_q, n0, stride, fixed_len=L, seed=2)
    eng.filter_device(d_q, n0, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    caps = eng.read_budgets(n0)
    host = d_q.download(np.uint8, n0 * stride).reshape(n0, stride)
    d_b = eng.alloc(target * stride)
    print("%5s %9s %9s %8s %8s %7s" % ("cap", "reads", "batch", "dp_ms", "Gcell/s", "fp64%"))
    tot_t = tot_c = 0.0
    for cap in sorted(set(caps.tolist())):
        idx = np.nonzero(caps == cap)[0]
        if len(idx) < 200:
            continue
        reps = max(1, target // len(idx))
        sel = np.tile(idx, reps)[:target]
        m = len(sel)
        d_b.upload(host[sel])
        eng.filter_device(d_b, m, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        hist = eng.class_histogram()
        assert hist.get(cap, 0) == m, (cap, hist)
        eng.timing(True); eng.timing_reset()
        for _ in range(5):
            eng.filter_device(d_b, m, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
        t = eng.kernel_times()["dp"]
        eng.timing(False)
        ms = t[0] / t[1]
        cells = cap * L * m
        share = len(idx) / n0
        tot_t += ms / m * len(idx)
        tot_c += cap * L * len(idx)
        print("%5d %9d %9d %8.3f %8.1f %6.1f%%   (%.1f%% of reads, %.1f%% of cells)"
              % (cap, len(idx), m, ms, cells / ms / 1e6, 100 * cells * 3 / (ms * 1e-3) / PEAK, 100 * share,
                 100 * cap * len(idx) / (caps.astype(np.int64).sum())), flush=True)
    print("sum of per-class times for the %d-read batch: %.3f ms -> %.1f%% fp64" % (n0, tot_t, 100 * tot_c * 3 / (tot_t * 1e-3) / PEAK))
