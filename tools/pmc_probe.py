#!/usr/bin/env python3
"""The workload of bench.py's live counter passes: started by bench.py as
    rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir> -- python3 tools/pmc_probe.py READS LENGTH SEED STRIDE
in a fresh process of its own.  A few launches of (a) the headline batch through the library's default path (k_dp), (b) the clean
profile of the same shape (k_narrow_rs / k_narrow) and (c) a clean ragged batch of half as many reads, U{50..600}, stride 640
(k_narrow_rg).  No torch, no oracle: the library through ctypes only."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
L = int(sys.argv[2]) if len(sys.argv) > 2 else 300
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 2
stride = int(sys.argv[4]) if len(sys.argv) > 4 else (L + 63) // 64 * 64
launches = int(os.environ.get("PMC_PROBE_LAUNCHES", "3"))
for profile in (0, 1):                       # a context each: the choice of pass is remembered per batch SHAPE, and these two share it
    with Engine(0) as eng:
        d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors")
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed, profile=profile)
        for _ in range(launches):
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        eng.synchronize()
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()
with Engine(0) as eng:
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors")
    nr, sr = max(n // 2, 1), 640
    if nr * sr <= n * stride:
        d_len = eng.alloc(nr * 4)
        eng.synth_fill(d_q, nr, sr, min_len=50, max_len=600, d_len=d_len, seed=6, profile=1)
        for _ in range(launches):
            eng.filter_device(d_q, nr, sr, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        eng.synchronize()
    print("pmc_probe: %d reads x %d (stride %d, seed %d), %d launches per batch" % (n, L, stride, seed, launches))
