#!/usr/bin/env python3
"""The natural-order narrow pass against the sorted pipeline on resident batches of several quality mixes: wall time per
step (back-to-back calls, one synchronisation at the end), per-kernel HIP-event times, reads handed back, the library's own
choice, and what the choice model predicted.  Output: the table behind DESIGN §4 and profiles/r05_narrow_choice.txt.

    python tools/narrow_rate.py [reads] [steps]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
stride, L = 320, 300
ALG = n * (L + 13)                      # algorithmic bytes per step


def host_mix(name, m, rng):
    """m x stride matrices the generator has no profile for"""
    q = np.zeros((m, stride), np.uint8)
    if name == "flat_q30":                                  # every read needs 3 rows
        q[:, :L] = 30
    elif name == "q28_35":                                  # 3 rows, a few 4
        q[:, :L] = rng.integers(28, 36, (m, L), dtype=np.uint8)
    elif name == "q25_32":                                  # 4 .. 5 rows
        q[:, :L] = rng.integers(25, 33, (m, L), dtype=np.uint8)
    elif name == "hq_plus_5pct_bad":                        # clean reads, one in twenty a bad one (Q8..20)
        q[:, :L] = rng.integers(33, 41, (m, L), dtype=np.uint8)
        bad = rng.random(m) < 0.05
        q[bad, :L] = rng.integers(8, 21, (int(bad.sum()), L), dtype=np.uint8)
    elif name == "hq_with_config2_Ns":                      # clean scores, BASELINE's 0.1 % ambiguous bases
        q[:, :L] = rng.integers(33, 41, (m, L), dtype=np.uint8)
        q[:, :L][rng.random((m, L)) < 66 / 65536] = 0
    return q


def timed(eng, d_q, bufs, params, k):
    d_ee, d_ns, d_pass = bufs
    eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)
    eng.synchronize()
    t0 = time.perf_counter()
    for _ in range(k):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)
    eng.synchronize()
    ms = (time.perf_counter() - t0) * 1e3 / k
    path = eng.last_path()
    eng.timing(True); eng.timing_reset()
    for _ in range(3):
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)
    kt = {k_: v[0] / max(1, v[1]) * (v[1] / 3.0) for k_, v in eng.kernel_times().items() if v[1]}
    eng.timing(False)
    return ms, path, kt


rng = np.random.default_rng(1)
mixes = [("synthetic profile 1 (clean: Q33-40)", 1), ("synthetic profile 0 (BASELINE config 2)", 0),
         "flat_q30", "q28_35", "q25_32", "hq_plus_5pct_bad", "hq_with_config2_Ns"]
print("%d reads x %d bases, stride %d; algorithmic bytes per step %.3f GB; %d timed steps per row" % (n, L, stride, ALG / 1e9, steps))
for mix in mixes:
    # a context per mix: the library reuses its choice while the batch SHAPE stays, and every mix here has the same shape
    with Engine(0) as eng:
        d_q = eng.alloc(n * stride)
        bufs = (eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n))
        if isinstance(mix, tuple):
            label = mix[0]
            eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2, profile=mix[1])
        else:
            label = mix
            m = 1_000_000
            h = host_mix(mix, m, rng)
            for off in range(0, n, m):                      # the same million rows, repeated
                k = min(m, n - off)
                from moira_amd import _lib as L_
                L_.check(eng.lib.mpb_memcpy_h2d(eng.ctx, d_q.ptr + off * stride, h.ctypes.data, k * stride))
        print("\n== %s" % label)
        rows = []
        for name, kw in (("sorted pipeline", dict(no_narrow=True)), ("narrow R=2", dict(narrow_rows=2)),
                         ("narrow R=3", dict(narrow_rows=3)), ("narrow R=4", dict(narrow_rows=4)), ("library's choice", dict())):
            if name == "library's choice":
                eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=bufs[0], d_ns=bufs[1], d_pass=bufs[2], want_counts=False)
                first = eng.last_path()
            ms, path, kt = timed(eng, d_q, bufs, eng.params(**kw), steps)
            extra = ""
            if name == "library's choice":
                extra = "  sample %s -> rows %d" % (first["sample_hist"], first["narrow_rows"])
            print("%-18s %7.3f ms/step  %5.2f TB/s alg = %.3f of 8 TB/s | rows %d, handed back %d (%.2f %%) | %s%s"
                  % (name, ms, ALG / ms / 1e9, ALG / ms / 1e9 / 8.0, path["narrow_rows"], path["n_fallback"],
                     100.0 * path["n_fallback"] / n, " ".join("%s %.3f" % (k_, v) for k_, v in sorted(kt.items())), extra), flush=True)
