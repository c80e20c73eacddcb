#!/usr/bin/env python3
"""Run ON THE GPU BOX: a soak of the resident one-read servers next to batch work on the same GPU -- P worker processes call
bernoulli.calculate_errors_PB through the broker (direct serving) for SECONDS seconds while this process runs 10 M-read
batch filters in a loop and its own per-read calls in between; every answer is compared with the oracle's.
  python3 tools/experiments/serve_soak.py [SECONDS] [P]"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
import numpy as np  # noqa: E402


def reads(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        n = int(rng.integers(1, 600))
        lo, hi = [(2, 41), (25, 41), (1, 8), (30, 42)][int(rng.integers(0, 4))]
        q = [int(v) for v in rng.integers(lo, hi, n)]
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        amb = rng.random(n) < 0.01
        s[amb] = ord("N")
        out.append((s.tobytes().decode(), q, float([0.005, 0.05][int(rng.integers(0, 2))])))
    return out


def worker(name, seed, seconds, out):
    import pb_oracle as O
    from moira_amd import broker
    cl = broker.client(0, name=name, idle_exit=5.0)
    rs = reads(seed, 200)
    want = [O.ee_rowwise(s, q, a)[:2] for s, q, a in rs]
    bad = calls = 0
    t_end = time.time() + seconds
    while time.time() < t_end:
        for r, w in zip(rs, want):
            bad += cl.calculate_errors_PB(*r) != w
            calls += 1
    cl.close()
    out.put((calls, bad))


if __name__ == "__main__":
    seconds = float(sys.argv[1]) if len(sys.argv) > 1 else 20.0
    P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    import pb_oracle as O
    from moira_amd import broker
    from moira_amd.engine import Engine
    name = "soak_%d" % os.getpid()
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(name, 900 + k, seconds, out)) for k in range(P)]
    for p in procs:
        p.start()
    n, stride, L = 10_000_000, 320, 300
    rs = reads(7, 100)
    want = [O.ee_rowwise(s, q, a)[:2] for s, q, a in rs]
    batches = own = own_bad = 0
    with Engine(0) as eng:
        d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
        c0 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        t_end = time.time() + seconds
        while time.time() < t_end:
            for _ in range(20):
                c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
                batches += 1
                assert c.n_pass == c0.n_pass
            for r, w in zip(rs, want):
                own_bad += eng.calculate_errors_PB(*r) != w
                own += 1
    res = [out.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(60)
    st = broker.stats(name)
    broker.shutdown(name)
    calls, bad = sum(r[0] for r in res), sum(r[1] for r in res)
    print("soak %.0f s: %d worker processes made %d calls through the broker (%.3g calls/s, %d wrong; broker: %s); beside them "
          "%d batch filters of 10 M reads and %d per-read calls of this process (%d wrong)"
          % (seconds, P, calls, calls / seconds, bad, st, batches, own, own_bad))
    sys.exit(1 if bad or own_bad else 0)
