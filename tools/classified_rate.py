#!/usr/bin/env python3
"""Raw FASTQ text resident in HBM -> results: decode + filter against classified at source (HIP events per kernel).
    python tools/classified_rate.py [reads]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
stride, L = 320, 300
with Engine(0) as eng:
    d_q, d_seq, d_qual, d_out = (eng.alloc(n * stride) for _ in range(4))
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
    eng.encode_ascii_device(d_q, n, stride, d_seq, d_qual)
    prm = eng.params()

    def two_pass():
        eng.decode_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L)
        eng.filter_device(d_out, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)

    def at_source():
        eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)

    for rep in range(int(os.environ.get("REPS", "3"))):
        for name, fn in (("decode + filter", two_pass), ("classified at source", at_source)):
            if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
                continue
            fn(); eng.synchronize()
            t = time.perf_counter()
            for _ in range(20):
                fn()
            eng.synchronize()
            dt = (time.perf_counter() - t) / 20
            eng.timing(True); eng.timing_reset()
            for _ in range(5):
                fn()
            kt = {k: round(v[0] / max(v[1], 1), 3) for k, v in eng.kernel_times().items() if v[1]}
            eng.timing(False)
            print("%-22s %.3f ms per step = %.3e reads/s   kernels(ms) %s" % (name, dt * 1e3, n / dt, kt), flush=True)
