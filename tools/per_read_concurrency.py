#!/usr/bin/env python3
"""What an UNCHANGED moira.py gets from the drop-in: its Pool(processors) workers each call
bernoulli.calculate_errors_PB per read (moira/moira.py:398-399, 817) -- here P worker processes (each with its own
context on the same GPU) make per-read calls at the same time; aggregate calls per second by P.
    python tools/per_read_concurrency.py [P ...]        (at most 5 workers: the GPU boxes allow 6 processes on the card)"""
import multiprocessing as mp
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(k, n, start, out):
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
    ps = [int(a) for a in sys.argv[1:]] or [1, 2, 4, 5]
    n = 20000
    ctx = mp.get_context("spawn")                      # no GPU state is inherited: every worker opens its own context
    for p in ps:
        p = min(p, 5)
        start, out = ctx.Event(), ctx.Queue()
        procs = [ctx.Process(target=worker, args=(k, n, start, out)) for k in range(p)]
        for pr in procs:
            pr.start()
        time.sleep(8)                                  # imports + context creation + warm-up
        start.set()
        times = [out.get() for _ in procs]
        for pr in procs:
            pr.join()
        print("%d worker process(es): %.1f us per call in a worker, %.3e calls/s in all" % (p, max(times) / n * 1e6, p * n / max(times)), flush=True)


if __name__ == "__main__":
    main()
