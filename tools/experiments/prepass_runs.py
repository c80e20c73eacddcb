#!/usr/bin/env python3
"""One process = one fresh context: where the buffers land and how long k_prepass takes (its time differs between
processes of the same binary by up to 14 %)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

n, stride, L = 10_000_000, 320, 300
order = sys.argv[1] if len(sys.argv) > 1 else "q_first"
with Engine(0) as eng:
    if order == "q_first":
        d_q = eng.alloc(n * stride); d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    else:
        d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n); d_q = eng.alloc(n * stride)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
    for _ in range(3):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
    eng.timing(True); eng.timing_reset()
    for _ in range(20):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
    t = eng.kernel_times()
    print("%s q=%#x ns=%#x: prepass %.3f ms, dp %.3f ms" % (order, d_q.ptr, d_ns.ptr, t["prepass"][0] / t["prepass"][1], t["dp"][0] / t["dp"][1]))
