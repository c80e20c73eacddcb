#!/usr/bin/env python3
"""One line: a ragged resident batch (lengths U{50..600}, stride 640; clean profile by default) through the narrow pass with R rows
(0: the library's choice): ms per step, the narrow span, handed back.  For tools/narrow_pmc.sh (PROBE=tools/ragged_probe.py).
    python tools/ragged_probe.py [R=3] [reads=5000000] [profile=1]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

R = int(sys.argv[1]) if len(sys.argv) > 1 else 3
n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
profile = int(sys.argv[3]) if len(sys.argv) > 3 else 1
stride, steps = int(os.environ.get("PROBE_STRIDE", "640")), 30
with Engine(0) as eng:
    d_q, d_len, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, min_len=50, max_len=min(600, stride), d_len=d_len, seed=6, profile=profile)
    prm = eng.params(narrow_rows=R)
    run = lambda: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
    for _ in range(40):                      # settle the clock
        run()
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        run()
    eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / steps
    eng.timing(True); eng.timing_reset()
    for _ in range(5):
        run()
    kt = eng.kernel_times()
    p = eng.last_path()
    print("R=%d step %.3f ms  narrow span %.3f ms  rest %.3f ms  rows taken %d  handed back %d" % (
        R, ms, kt["narrow"][0] / 5, sum(v[0] for k, v in kt.items() if k != "narrow") / 5, p["narrow_rows"], p["n_fallback"]))
