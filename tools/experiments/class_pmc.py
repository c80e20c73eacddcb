#!/usr/bin/env python3
"""One single-class batch (rows of the config-2 workload whose row budget is CAP, replicated on the device) run
through the pipeline a few times: the workload for `rocprofv3 --pmc ... -- python3 tools/experiments/class_pmc.py CAP`."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402
from moira_amd import _lib as L_  # noqa: E402

cap = int(sys.argv[1])
target = int(sys.argv[2]) if len(sys.argv) > 2 else 16_000_000
n0, stride, L = 4_000_000, 320, 300
with Engine(0) as eng:
    d_q = eng.alloc(n0 * stride)
    nmax = max(n0, target)
    d_ee, d_ns, d_pass = eng.alloc(nmax * 8), eng.alloc(nmax * 4), eng.alloc(nmax)
    eng.synth_fill(d_q, n0, stride, fixed_len=L, seed=2)
    eng.filter_device(d_q, n0, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    caps = eng.read_budgets(n0)
    host = d_q.download(np.uint8, n0 * stride).reshape(n0, stride)
    d_q.free()
    idx = np.nonzero(caps == cap)[0]
    G = 1 if cap <= 16 else 2 if cap <= 32 else 4 if cap <= 64 else 8 if cap <= 128 else 16
    uniq = np.ascontiguousarray(host[idx[:1_000_000]])
    d_b = eng.alloc(target * stride)
    m = 0
    while m + len(uniq) <= target // G:
        L_.check(eng.lib.mpb_memcpy_h2d(eng.ctx, d_b.ptr + m * stride, uniq.ctypes.data, uniq.nbytes))
        m += len(uniq)
    for _ in range(3):
        eng.filter_device(d_b, m, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    print("cap %d: %d reads, class histogram %s" % (cap, m, {k: v for k, v in eng.class_histogram().items() if v}))
