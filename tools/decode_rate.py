#!/usr/bin/env python3
"""SURVEY f-4: on-device decode of raw FASTQ quality bytes + base letters into the packed quality matrix
(k_decode_ascii) on a resident batch: bytes moved = 2 reads + 1 write per base."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
stride, L = 320, 300
with Engine(0) as eng:
    d_seq, d_qual, d_out = eng.alloc(n * stride), eng.alloc(n * stride), eng.alloc(n * stride)
    eng.lib.mpb_memset(eng.ctx, d_seq.ptr, ord("A"), n * stride)
    eng.synth_fill(d_qual, n, stride, fixed_len=L, seed=2)          # bytes 0..40: as ASCII with offset 0 they decode to themselves
    for _ in range(2):
        eng.decode_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, fastq_offset=0)
    eng.synchronize()
    t = time.perf_counter()
    reps = 10
    for _ in range(reps):
        eng.decode_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, fastq_offset=0)
    eng.synchronize()
    dt = (time.perf_counter() - t) / reps
    print("k_decode_ascii: %.3f ms per %d x %d-byte rows = %.2f TB/s moved (2 matrices read, 1 written), %.3e reads/s"
          % (dt * 1e3, n, stride, 3 * n * stride / dt / 1e12, n / dt))
