#!/usr/bin/env python3
"""Does k_prepass' time depend on where the matrix lies?  Same process, same data: (a) the matrix at different offsets
inside one allocation, (b) fresh allocations."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

n, stride, L = 10_000_000, 320, 300


def run(eng, ptr, d_ee, d_ns, d_pass, reps=8):
    eng.synth_fill(ptr, n, stride, fixed_len=L, seed=2)
    for _ in range(2):
        eng.filter_device(ptr, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
    eng.timing(True); eng.timing_reset()
    for _ in range(reps):
        eng.filter_device(ptr, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False)
    t = eng.kernel_times()
    eng.timing(False)
    return t["prepass"][0] / t["prepass"][1], t["dp"][0] / t["dp"][1]


with Engine(0) as eng:
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    big = eng.alloc(n * stride + (8 << 20))
    print("base pointer %#x" % big.ptr)
    for off in (0, 16, 256, 4096, 65536, 1 << 20, 2 << 20, (2 << 20) + 4096, 4 << 20, 0):
        pre, dp = run(eng, big.ptr + off, d_ee, d_ns, d_pass)
        print("offset %9d: prepass %.3f ms, dp %.3f ms" % (off, pre, dp), flush=True)
    big.free()
    for k in range(6):
        b = eng.alloc(n * stride + k * (3 << 20))
        pre, dp = run(eng, b.ptr, d_ee, d_ns, d_pass)
        print("fresh allocation %d at %#x: prepass %.3f ms, dp %.3f ms" % (k, b.ptr, pre, dp), flush=True)
        b.free()
