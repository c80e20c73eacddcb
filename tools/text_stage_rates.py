#!/usr/bin/env python3
"""Rates of the CLI text stages one by one (read + index, pack, format) against the thread count: tools/text_stage_rates.py FILE.fastq"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from moira_amd import fastio as F
from concurrent.futures import ThreadPoolExecutor
path = sys.argv[1]
for th in (1, 2, 4, 8, 16):
    t = time.perf_counter()
    tot = 0
    with open(path, 'rb') as fh:
        for buf, idx in F.FastqChunks(fh, 262144, threads=th):
            tot += len(idx)
    dt = time.perf_counter() - t
    print("FastqChunks threads=%d: %.3f s = %.2e reads/s" % (th, dt, tot / dt), flush=True)
with open(path, 'rb') as fh:
    buf, idx = next(iter(F.FastqChunks(fh, 262144, threads=8)))
sel = np.arange(len(idx))
for th in (1, 2, 4, 8, 16):
    pool = ThreadPoolExecutor(th) if th > 1 else None
    F.pack_parallel(pool, th, buf, idx, sel, 33, 0, False, 256)
    t = time.perf_counter()
    for _ in range(5):
        F.pack_parallel(pool, th, buf, idx, sel, 33, 0, False, 256)
    dt = (time.perf_counter() - t) / 5
    print("pack_parallel threads=%d: %.2e reads/s" % (th, len(idx) / dt), flush=True)
    for kind, name in ((F.FMT_FASTQ, 'fastq'), (F.FMT_QUAL, 'qual')):
        F.format_parallel(pool, th, buf, idx, sel, kind)
        t = time.perf_counter()
        for _ in range(3):
            F.format_parallel(pool, th, buf, idx, sel, kind)
        dt = (time.perf_counter() - t) / 3
        print("   format_parallel %s threads=%d: %.2e reads/s" % (name, th, len(idx) / dt), flush=True)
