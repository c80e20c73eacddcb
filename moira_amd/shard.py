"""Host-side split / gather of a read batch across the GPUs of one node.

The path shards with no exchange step: reads are independent, so rank r of W owns the
contiguous index range [r*n/W, (r+1)*n/W), filters it on its own MI355X, and results are either kept /
written per rank or concatenated in read order.  No collective is on the data path.  The only collectives
are optional and small: a 3 x int64 all-reduce of the pass / fail / overflow totals, which is what moira
prints at the end of a run (moira/moira.py:508-519) -- 24 bytes over RCCL/xGMI -- and, when a caller wants
every rank to see all results, three `all_gather_into_tensor` calls on fixed-width padded buffers
(13 bytes per read; nothing is pickled).

This replaces moira's `Pool(processors)` + per-read `apply_async` + `.get()` barrier
(moira/moira.py:398-399,431-454): one process per GPU instead of one task per read.

Ownership: a rank only ever materialises ITS range.  `filter_sharded_owned` takes a
`load_fn(lo, hi) -> (q, lens)` (read the slice of a file, generate it, ...) and calls it exactly once
with the rank's bounds; `filter_synth_shard` is the device-resident form used for BASELINE config 4
(1 B reads over 8 GPUs: each rank generates its 125 M reads in its own HBM).  `filter_sharded` keeps the
replicated-input signature for callers that already hold the whole batch (it slices views, it does not
copy) -- at config-4 scale use the owned forms.

The functions take a `filter_fn(q, lens) -> (ee, ns, passed)` so the same split/gather logic is
exercised on CPU ranks under gloo in the tests (with the oracle as filter_fn) and on GPU ranks under
RCCL in production (Engine.filter).
"""
import numpy as np


class MultiEngine:
    """Several contexts -- normally one per GPU of the node -- driven from ONE host process ("one host
    thread (or process) + one HIP stream per device").  `filter()` / `filter_poisson()` take the same arguments as
    Engine's and return the same FilterResult: the batch is cut into contiguous shards in read order
    (`shard_bounds`), one host thread per context runs the chunked H2D / kernels / D2H pipeline on its shard, and every
    shard writes into its slice of the result arrays, so nothing is gathered afterwards and there is no collective.
    For a host-fed caller this is what moira's `Pool(args.processors)` (moira/moira.py:398-399) was: all the GPUs
    (and all their PCIe links) of a node behind one call -- without torchrun.  The split itself lives in the C ABI
    (`mpb_filter_host_multi`), so a C caller gets it too.

        with MultiEngine() as me:                 # every visible GPU;  MultiEngine([0, 1]) / MultiEngine([0, 0, 0])
            r = me.filter(q, lens=lens, alpha=0.005)

    Listing a device more than once gives it that many contexts (tests do that on a one-GPU box)."""

    def __init__(self, devices=None):
        import ctypes as C
        from . import _lib as L
        from .engine import Engine
        if devices is None:
            devices = list(range(L.load().mpb_device_count()))
        devices = [int(d) for d in devices]
        if not devices:
            raise L.NoDeviceError("no HIP device visible (this library has no CPU path)")
        self.engines = []
        try:
            for d in devices:
                self.engines.append(Engine(d))
        except Exception:
            self.close()
            raise
        self.devices = devices
        self.lib = self.engines[0].lib
        self._ctxs = (C.c_void_p * len(self.engines))(*[e.ctx for e in self.engines])
        self.batched_only = False

    def close(self):
        for e in getattr(self, "engines", []):
            e.close()
        self.engines = []

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # what a host-side caller (cli.py, buckets.py) uses of an Engine, forwarded to the first context
    def params(self, **kw):
        return self.engines[0].params(**kw)

    def pack(self, *a, **kw):
        return self.engines[0].pack(*a, **kw)

    def pack_batch_ascii(self, *a, **kw):
        return self.engines[0].pack_batch_ascii(*a, **kw)

    def calculate_errors_PB(self, contig, contig_quals, alpha):
        return self.engines[0].calculate_errors_PB(contig, contig_quals, alpha)

    def calculate_errors_poisson(self, sequence, quals, alpha):
        return self.engines[0].calculate_errors_poisson(sequence, quals, alpha)

    def _run(self, poisson, q, lens, fixed_len, out, kw):
        import ctypes as C
        from . import _lib as L
        from .engine import FilterResult, check_host_batch
        if not poisson:
            kw.setdefault("batched_only", self.batched_only)
        params = kw.pop("params", None) or self.params(**kw)
        q, n, stride, lens, (ee, ns, ps) = check_host_batch(q, lens, fixed_len, out, limit=None if poisson else L.MAX_LEN)
        counts = L.FilterCounts()
        L.check(self.lib.mpb_filter_host_multi(self._ctxs, len(self.engines), q.ctypes.data, n, stride,
                                               lens.ctypes.data if lens is not None else None,
                                               0 if lens is not None else int(fixed_len), C.byref(params),
                                               ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, C.byref(counts),
                                               1 if poisson else 0))
        return FilterResult(ee, ns, ps.view(bool), counts.n_pass, counts.n_overflow)

    def filter(self, q, lens=None, fixed_len=None, out=None, **kw):
        return self._run(False, q, lens, fixed_len, out, kw)

    def filter_poisson(self, q, lens=None, fixed_len=None, out=None, **kw):
        return self._run(True, q, lens, fixed_len, out, kw)

    def shards(self, n):
        """[(lo, hi)] the contexts take of a batch of n reads."""
        return [shard_bounds(n, len(self.engines), r) for r in range(len(self.engines))]


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


def _coll_device(dist):
    import torch
    if dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def reduce_totals(totals, dist=None):
    """Sum a tuple of per-rank integer totals over all ranks (the one optional 24-byte collective)."""
    if dist is None or dist.get_world_size() == 1:
        return tuple(int(x) for x in totals)
    import torch
    t = torch.tensor([int(x) for x in totals], dtype=torch.int64, device=_coll_device(dist))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return tuple(int(x) for x in t.tolist())


def gather_results(ee, ns, passed, n_total, dist):
    """Every rank gets the full-length (ee, ns, passed) in read order.  Shards differ by at most one read, so
    each rank pads its three arrays to ceil(n_total / W) and three all_gather_into_tensor calls move
    13 bytes per read; the padding is cut out with the known bounds.  No pickling, no per-rank Python lists."""
    import torch
    world, rank = dist.get_world_size(), dist.get_rank()
    width = -(-n_total // world) if n_total else 0
    dev = _coll_device(dist)
    out = []
    for arr, tdt in ((np.ascontiguousarray(ee, np.float64), torch.float64),
                     (np.ascontiguousarray(ns, np.int32), torch.int32),
                     (np.ascontiguousarray(passed, bool).view(np.uint8), torch.uint8)):
        mine = torch.zeros(max(width, 1), dtype=tdt)
        mine[:len(arr)] = torch.from_numpy(arr)
        mine = mine.to(dev)
        full = torch.empty(max(width, 1) * world, dtype=tdt, device=dev)
        dist.all_gather_into_tensor(full, mine)
        full = full.cpu().numpy().reshape(world, max(width, 1))
        parts = []
        for r in range(world):
            lo, hi = shard_bounds(n_total, world, r)
            parts.append(full[r, :hi - lo])
        out.append(np.concatenate(parts) if parts else full[:0])
    del rank
    return out[0], out[1], out[2].astype(bool)


def filter_sharded_owned(n_total, load_fn, filter_fn, dist=None, gather=False):
    """Each rank loads and filters ONLY its own range: `load_fn(lo, hi) -> (q, lens)` is called once, with
    this rank's bounds.  Returns (ee, ns, passed, totals, (lo, hi)); the arrays cover the rank's range
    (gather=False: a rank then writes its own part of the output, as moira's writers do per file) or the whole
    batch on every rank (gather=True).  totals = (n_pass, n_fail) over ALL ranks."""
    world = dist.get_world_size() if dist is not None else 1
    rank = dist.get_rank() if dist is not None else 0
    lo, hi = shard_bounds(n_total, world, rank)
    q, lens = load_fn(lo, hi)
    if len(lens) != hi - lo:
        raise ValueError("load_fn returned %d reads for the range [%d, %d)" % (len(lens), lo, hi))
    ee, ns, passed = filter_fn(q, lens)
    ee = np.ascontiguousarray(ee, np.float64)
    ns = np.ascontiguousarray(ns, np.int32)
    passed = np.ascontiguousarray(passed, bool)
    n_pass = int(passed.sum())
    totals = reduce_totals((n_pass, (hi - lo) - n_pass), dist)
    if gather and world > 1:
        ee, ns, passed = gather_results(ee, ns, passed, n_total, dist)
    return ee, ns, passed, totals, (lo, hi)


def filter_sharded(q, lens, filter_fn, dist=None, gather=True):
    """Replicated-input form (every rank can see the same q/lens, e.g. a memory-mapped file): each rank
    computes only its slice (a view, not a copy).  Returns (ee, ns, passed, totals); see filter_sharded_owned."""
    n = len(lens)
    ee, ns, passed, totals, _ = filter_sharded_owned(n, lambda lo, hi: (q[lo:hi], lens[lo:hi]), filter_fn,
                                                     dist=dist, gather=gather)
    return ee, ns, passed, totals


def filter_synth_shard(engine, n_total, world, rank, length, stride, seed, download=True, **params):
    """BASELINE config 4 on one rank: generate this rank's range of the synthetic batch IN ITS OWN HBM (read
    ids lo..hi-1 of the counter-based generator; no host copy of the inputs exists anywhere) and filter it
    there.  Returns (ee, ns, passed, (n_pass, n_fail, n_overflow)) for the range, or only the totals when
    download=False (13 bytes per read stay on the device)."""
    lo, hi = shard_bounds(n_total, world, rank)
    m = hi - lo
    if m == 0:
        return np.empty(0), np.empty(0, np.int32), np.empty(0, bool), (0, 0, 0)
    d_q, d_ee, d_ns, d_pass = engine.alloc(m * stride), engine.alloc(m * 8), engine.alloc(m * 4), engine.alloc(m)
    try:
        engine.synth_fill(d_q, m, stride, fixed_len=length, seed=seed, first_read=lo)
        c = engine.filter_device(d_q, m, stride, fixed_len=length, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                 params=engine.params(**params))
        totals = (c.n_pass, c.n_fail, c.n_overflow)
        if not download:
            return None, None, None, totals
        return (d_ee.download(np.float64, m), d_ns.download(np.int32, m),
                d_pass.download(np.uint8, m).astype(bool), totals)
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()
