#!/usr/bin/env python3
"""What an UNCHANGED moira.py gets from the drop-in: its Pool(processors) workers each call
bernoulli.calculate_errors_PB per read (moira/moira.py:398-399, 817) -- here P worker processes make per-read calls at
the same time; aggregate calls per second by P.
    python tools/per_read_concurrency.py [P ...]                 # round 4: through the broker (one GPU-owning process)
    MOIRA_PB_BROKER=0 python tools/per_read_concurrency.py 1 2 4 5   # round 3: a context per worker (at most 5 workers:
                                                                     # the GPU boxes allow 6 processes on the card)"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(k, n, start, out):
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
    start.wait()
    t = time.perf_counter()
    for i in range(n):
        s, q = reads[i & 63]
        bernoulli.calculate_errors_PB(s, q, 0.005)
    out.put(time.perf_counter() - t)


def main():
    ps = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8, 16]
    n = 20000
    direct = os.environ.get("MOIRA_PB_BROKER") == "0"
    os.environ.setdefault("MOIRA_PB_BROKER_NAME", "conc_%d" % os.getpid())
    ctx = mp.get_context("spawn")                      # no GPU state is inherited: every worker opens its own context
    for p in ps:
        p = min(p, 5) if direct else p
        start, out = ctx.Event(), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(k, n, start, out)) for k in range(p)]
        for pr in procs:
            pr.start()
        time.sleep(8)                                  # imports + context creation + warm-up
        start.set()
        times = [out.get() for _ in procs]
        for pr in procs:
            pr.join()
        extra = ""
        if not direct:
            sys.path.insert(0, ROOT)
            from moira_amd import broker
            st = broker.stats(os.environ["MOIRA_PB_BROKER_NAME"])
            if st:
                extra = "  [broker: %d reads in %d launches + %d alone since it started]" % (st["served"], st["batches"], st["solo"])
        print("%d worker process(es): %.1f us per call in a worker, %.3e calls/s in all%s" % (p, max(times) / n * 1e6, p * n / max(times), extra), flush=True)
    if not direct:
        from moira_amd import broker
        broker.shutdown(os.environ["MOIRA_PB_BROKER_NAME"])


if __name__ == "__main__":
    main()
