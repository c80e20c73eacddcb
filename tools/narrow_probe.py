#!/usr/bin/env python3
"""One line: the clean 10 M x 300 batch through the narrow pass with R rows (default 2): ms per step, k_narrow ms, handed back.
For tools/experiments/variants.sh (VARIANT_CMD="python tools/narrow_probe.py 2")."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
stride, L, steps = int(os.environ.get("PROBE_STRIDE", "320")), 300, 30      # PROBE_STRIDE=304: rows that are no multiple of 64 bytes
with Engine(0) as eng:
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2, profile=1)
    prm = eng.params(narrow_rows=R)
    for _ in range(40):                      # settle the clock
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
    eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
    kt = eng.kernel_times()
    print("R=%d step %.3f ms  narrow %.3f ms  rest %.3f ms  handed back %d" % (
        R, ms, kt["narrow"][0] / 5, sum(v[0] for k, v in kt.items() if k != "narrow") / 5, eng.last_path()["n_fallback"]))
