#!/bin/bash
# Run ON THE GPU BOX: how small may the share of reads the narrow pass finishes be before the sorted pipeline wins?
# 10 M x 300: clean reads (Q33-40) with a share of (a) bad reads (Q8-20) (b) BASELINE-config-2-like reads mixed in.
python3 - <<'PY'
import os, sys, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "oracle"))
import numpy as np
import pb_oracle as O
from moira_amd.engine import Engine
from moira_amd import _lib as L_
n, stride, L, m = 10_000_000, 320, 300, 1_000_000
rng = np.random.default_rng(1)
clean = np.zeros((m, stride), np.uint8); clean[:, :L] = rng.integers(33, 41, (m, L), dtype=np.uint8)
bad = np.zeros((m, stride), np.uint8); bad[:, :L] = rng.integers(8, 21, (m, L), dtype=np.uint8)
c2, _ = O.synth_fill(m, stride, fixed_len=L, seed=2)
def run(eng, d_q, bufs, **kw):
    prm = eng.params(**kw)
    f = lambda: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=bufs[0], d_ns=bufs[1], d_pass=bufs[2], params=prm, want_counts=False)
    for _ in range(3): f()
    eng.synchronize(); t = time.perf_counter()
    for _ in range(10): f()
    eng.synchronize()
    return (time.perf_counter() - t) * 100, eng.last_path()
for name, other in (("bad Q8-20", bad), ("config-2-like", c2)):
    for share in (0.05, 0.1, 0.2, 0.3, 0.4, 0.5):
        pick = rng.random(m) < share
        h = np.where(pick[:, None], other, clean)
        with Engine(0) as eng:
            d_q = eng.alloc(n * stride); bufs = (eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n))
            for off in range(0, n, m):
                L_.check(eng.lib.mpb_memcpy_h2d(eng.ctx, d_q.ptr + off * stride, h.ctypes.data, m * stride))
            ts, _ = run(eng, d_q, bufs, no_narrow=True)
            res = []
            for R in (2, 3, 4):
                t, p = run(eng, d_q, bufs, narrow_rows=R)
                res.append("R=%d %.3f ms (%.0f %% handed back)" % (R, t, 100.0 * p["n_fallback"] / n))
            ta, pa = run(eng, d_q, bufs)
            print("%s share %.2f: sorted %.3f ms | %s | library: rows %d %.3f ms" % (name, share, ts, " | ".join(res), pa["narrow_rows"], ta), flush=True)
PY
