#!/usr/bin/env python3
"""What an UNCHANGED moira.py gets from the drop-in: its Pool(processors) workers each call
bernoulli.calculate_errors_PB per read (moira/moira.py:398-399, 817) -- here P worker processes make per-read calls at
the same time; aggregate calls per second by P.
    python tools/per_read_concurrency.py [--json] [P ...]        # round 4: through the broker (one GPU-owning process)
    MOIRA_PB_BROKER=0 python tools/per_read_concurrency.py 1 2 4 5   # round 3: a context per worker (at most 5 workers:
                                                                     # the GPU boxes allow 6 processes on the card)"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(k, n, start, out, ready):
    os.environ.setdefault("MOIRA_PB_BROKER", "1")
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "moira_amd", "dropin"))
    import numpy as np
    import bernoulli
    rng = np.random.default_rng(k)
    reads = []
    for _ in range(64):
        seq = "".join(rng.choice(list("ACGT"), 300))
        quals = [int(x) for x in np.clip(38 - (np.arange(300) / 300) ** 3 * 20 - rng.integers(0, 6, 300), 2, 40)]
        reads.append((seq, quals))
    for s, q in reads[:20]:
        bernoulli.calculate_errors_PB(s, q, 0.005)
    ready.put(k)                                       # imports done, context / broker attachment made, warmed up
    start.wait()
    t = time.perf_counter()
    for i in range(n):
        s, q = reads[i & 63]
        bernoulli.calculate_errors_PB(s, q, 0.005)
    out.put(time.perf_counter() - t)


def measure(p, n=20000, timeout=180):
    """P worker processes, n calls each, started together once every one of them is warmed up -> (calls/s in all, us per
    call in the slowest worker, broker statistics or None)."""
    ctx = mp.get_context("spawn")                      # no GPU state is inherited
    start, out, ready = ctx.Event(), ctx.Queue(), ctx.Queue()
    procs = [ctx.Process(target=worker, args=(k, n, start, out, ready)) for k in range(p)]
    for pr in procs:
        pr.start()
    for _ in procs:
        ready.get(timeout=timeout)
    start.set()
    times = [out.get(timeout=timeout) for _ in procs]
    for pr in procs:
        pr.join()
    st = None
    if os.environ.get("MOIRA_PB_BROKER") != "0":
        sys.path.insert(0, ROOT)
        from moira_amd import broker
        st = broker.stats(os.environ["MOIRA_PB_BROKER_NAME"])
    return p * n / max(times), max(times) / n * 1e6, st


def main():
    argv = [a for a in sys.argv[1:] if a != "--json"]
    as_json = "--json" in sys.argv[1:]
    ps = [int(a) for a in argv] or [1, 2, 4, 8, 16]
    direct = os.environ.get("MOIRA_PB_BROKER") == "0"
    os.environ.setdefault("MOIRA_PB_BROKER_NAME", "conc_%d" % os.getpid())
    rows = []
    for p in ps:
        p = min(p, 5) if direct else p
        rate, us, st = measure(p)
        rows.append({"workers": p, "calls_per_s": rate, "us_per_call_in_a_worker": us, "broker": st})
        if not as_json:
            extra = "  [broker: %d reads in %d launches + %d alone since it started]" % (st["served"], st["batches"], st["solo"]) if st else ""
            print("%d worker process(es): %.1f us per call in a worker, %.3e calls/s in all%s" % (p, us, rate, extra), flush=True)
    if not direct:
        from moira_amd import broker
        broker.shutdown(os.environ["MOIRA_PB_BROKER_NAME"])
    if as_json:
        import json
        print(json.dumps(rows))


if __name__ == "__main__":
    main()
