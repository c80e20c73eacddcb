#!/usr/bin/env python3
"""Run ON THE GPU BOX: where the RAGGED narrow pass stops paying -- 5 M clean reads of U{50..600} bases (stride 640) with a share of
(a) bad reads (Q8-20), (b) BASELINE-config-2-like reads mixed in; and the reference's paired contigs: the sorted pipeline against the
pass forced with 2 / 3 / 4 rows against the library's own choice (which should be within a few per cent of the best of the four).
    python tools/experiments/ragged_mix.py"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import pb_oracle as O  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n, stride, m = 5_000_000, 640, 500_000
clean, cl = O.synth_fill(m, stride, min_len=50, max_len=600, seed=6, profile=1)
c2, _ = O.synth_fill(m, stride, min_len=50, max_len=600, seed=6, profile=0)       # same lengths (the length draw does not depend on the profile)
rng = np.random.default_rng(1)
bad = clean.copy()
bad[bad > 0] = rng.integers(8, 21, int((bad > 0).sum()), dtype=np.uint8)


def run(eng, d_q, d_len, bufs, **kw):
    prm = eng.params(**kw)
    f = lambda: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=bufs[0], d_ns=bufs[1], d_pass=bufs[2], params=prm, want_counts=False)
    for _ in range(3):
        f()
    eng.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        f()
    eng.synchronize()
    return (time.perf_counter() - t) * 100, eng.last_path()


for name, other in (("bad Q8-20", bad), ("config-2-like", c2)):
    for share in (0.0, 0.02, 0.05, 0.1, 0.2, 0.3):
        pick = rng.random(m) < share
        h = np.where(pick[:, None], other, clean)
        with Engine(0) as eng:
            d_q, d_len = eng.alloc(n * stride), eng.alloc(n * 4)
            bufs = (eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n))
            for off in range(0, n, m):
                eng.lib.mpb_memcpy_h2d(eng.ctx, d_q.ptr + off * stride, h.ctypes.data, h.nbytes)
                eng.lib.mpb_memcpy_h2d(eng.ctx, d_len.ptr + off * 4, cl.ctypes.data, cl.nbytes)
            res = {}
            for label, kw in (("sorted", dict(no_narrow=True)), ("R2", dict(narrow_rows=2)), ("R3", dict(narrow_rows=3)), ("R4", dict(narrow_rows=4)),
                              ("choice", {})):
                ms, path = run(eng, d_q, d_len, bufs, **kw)
                res[label] = (ms, path["narrow_rows"], path["n_fallback"])
            best = min(v[0] for k, v in res.items() if k != "choice")
            print("%-14s share %.2f: " % (name, share) + "  ".join("%s %.3f ms (rows %d, back %d)" % (k, v[0], v[1], v[2]) for k, v in res.items())
                  + "   choice / best = %.3f" % (res["choice"][0] / best), flush=True)
