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
from moira_amd import _lib as L_  # noqa: E402

PEAK = 39.3e12
n0 = int(sys.argv[1]) if len(sys.argv) > 1 else 6_000_000
target = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000_000      # reads per single-class batch for G = 1 (divided by G)
UNIQ = 1_000_000                                                      # distinct rows uploaded per class, replicated on the device
stride, L = 320, 300
with Engine(0) as eng:
    d_q = eng.alloc(n0 * stride)
    nmax = max(n0, target)
    d_ee, d_ns, d_pass = eng.alloc(nmax * 8), eng.alloc(nmax * 4), eng.alloc(nmax)
    eng.synth_fill(d_q, n0, stride, fixed_len=L, seed=2)
    eng.filter_device(d_q, n0, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(no_narrow=True))
    caps = eng.read_budgets(n0)
    host = d_q.download(np.uint8, n0 * stride).reshape(n0, stride)
    d_b = eng.alloc(target * stride)
    print("%5s %9s %9s %8s %8s %7s" % ("cap", "reads", "batch", "dp_ms", "Gcell/s", "fp64%"))
    tot_t = tot_c = 0.0
    for cap in sorted(set(caps.tolist())):
        idx = np.nonzero(caps == cap)[0]
        if len(idx) < 200:
            continue
        G = 1 if cap <= 16 else 2 if cap <= 32 else 4 if cap <= 64 else 8 if cap <= 128 else 16
        uniq = np.ascontiguousarray(host[idx[:UNIQ]])
        m = 0
        while m + len(uniq) <= target // G:          # replicate the class's rows until the batch fills the chip many times over
            L_.check(eng.lib.mpb_memcpy_h2d(eng.ctx, d_b.ptr + m * stride, uniq.ctypes.data, uniq.nbytes))
            m += len(uniq)
        if m == 0:
            continue
        eng.filter_device(d_b, m, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=eng.params(no_narrow=True))
        hist = eng.class_histogram()
        assert hist.get(cap, 0) == m, (cap, hist)
        eng.timing(True); eng.timing_reset()
        for _ in range(4):
            eng.filter_device(d_b, m, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False, params=eng.params(no_narrow=True))
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
