"""Host-side split / gather of a read batch across the GPUs of one node.

The path shards with no exchange step: reads are independent, so rank r of W takes
the contiguous index range [r*n/W, (r+1)*n/W), filters it on its own MI355X, and results are
concatenated in read order.  No collective is on the data path.  The only (optional)
collective is a 3 x int64 all-reduce of the pass / fail / overflow totals, which is what
moira prints at the end of a run (moira/moira.py:508-519) -- 24 bytes over RCCL/xGMI.

This replaces moira's `Pool(processors)` + per-read `apply_async` + `.get()` barrier
(moira/moira.py:398-399,431-454): one process per GPU instead of one task per read.

The functions take a `filter_fn(q, lens) -> (ee, ns, passed)` so the same split/gather logic
is exercised on CPU ranks under gloo in the tests (with the oracle as filter_fn) and on GPU
ranks under RCCL in production (Engine.filter).
"""
import numpy as np


def shard_bounds(n, world, rank):
    """Contiguous, balanced, order-preserving partition of range(n): rank -> [lo, hi)."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank/world")
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def engine_filter_fn(engine, **params):
    """filter_fn backed by the HIP library."""
    def fn(q, lens):
        r = engine.filter(q, lens=lens, **params)
        return r.ee, r.ns, r.passed
    return fn


def filter_sharded(q, lens, filter_fn, dist=None, gather=True):
    """Filter the whole batch (every rank holds the same q/lens, as every rank of a job can
    read the same input file), each rank computing only its shard.

    Returns (ee, ns, passed, totals).  With gather=True every rank gets the full-length arrays
    (all_gather of the variable-size shards, order preserved); with gather=False the arrays
    cover only this rank's shard.  totals = (n_pass, n_fail) over ALL ranks."""
    n = len(lens)
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_bounds(n, world, rank)
    ee, ns, passed = filter_fn(q[lo:hi], lens[lo:hi])
    ee = np.ascontiguousarray(ee, np.float64)
    ns = np.ascontiguousarray(ns, np.int32)
    passed = np.ascontiguousarray(passed, bool)
    n_pass = int(passed.sum())
    totals = (n_pass, (hi - lo) - n_pass)
    if dist is None or world == 1:
        return ee, ns, passed, totals
    import torch
    t = torch.tensor(totals, dtype=torch.int64)
    dev = None
    if dist.get_backend() == "nccl":
        dev = torch.device("cuda", torch.cuda.current_device())
        t = t.to(dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    totals = tuple(int(x) for x in t.tolist())
    if not gather:
        return ee, ns, passed, totals
    parts = [None] * world
    dist.all_gather_object(parts, (ee, ns, passed))      # results only: 13 bytes per read
    return (np.concatenate([p[0] for p in parts]), np.concatenate([p[1] for p in parts]),
            np.concatenate([p[2] for p in parts]), totals)
