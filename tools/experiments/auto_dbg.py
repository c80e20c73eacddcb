import os, sys, time
sys.path.insert(0, os.getcwd())
from moira_amd.engine import Engine
n, stride, L = 10_000_000, 320, 300
with Engine(0) as eng:
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2, profile=1)
    for name, prm in (("forced2", eng.params(narrow_rows=2)), ("auto", eng.params()), ("forced2", eng.params(narrow_rows=2)), ("auto", eng.params())):
        ts = []
        for k in range(12):
            eng.synchronize(); t0 = time.perf_counter()
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
            eng.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
        p = eng.last_path()
        print(name, " ".join("%.2f" % t for t in ts), p["narrow_rows"], p["sampled"], p["n_fallback"])
        eng.synchronize(); t0 = time.perf_counter()
        for k in range(20):
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        eng.synchronize(); print(name, "20 back to back: %.3f ms/step" % ((time.perf_counter() - t0) * 1e3 / 20))
