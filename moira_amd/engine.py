"""Host-side mirror of the reference's filter path, batch-shaped.

Reference interface mirrored (same argument meaning and error behaviour):
  * `bernoulli.calculate_errors_PB(contig, contig_quals, alpha)`  moira/bernoullimodule.c:66-114
  * the filter half of `process_data`                             moira/moira.py:806-831
  * the keep/discard predicate of `write_results`                 moira/moira.py:911,925-926,949-950
What replaces moira's per-read `Pool.apply_async` dispatch (moira/moira.py:431-454) is:
pack a chunk of reads into an (N x L_max) uint8 matrix -> one library call -> arrays back.

Everything is computed by libmoira_pb.so on the GPU; this module only marshals.
"""
import ctypes as C
import math

import numpy as np

from . import _lib as L


def _round_up(x, m):
    return (x + m - 1) // m * m


class DeviceBuffer:
    """A device allocation owned by an Engine (freed with it or by .free())."""

    def __init__(self, engine, nbytes):
        self.engine, self.nbytes = engine, int(nbytes)
        p = C.c_void_p()
        L.check(engine.lib.mpb_malloc(engine.ctx, self.nbytes, C.byref(p)))
        self.ptr = p.value

    def upload(self, arr):
        arr = np.ascontiguousarray(arr)
        assert arr.nbytes <= self.nbytes
        L.check(self.engine.lib.mpb_memcpy_h2d(self.engine.ctx, self.ptr, arr.ctypes.data, arr.nbytes))
        return self

    def download(self, dtype, count, offset=0):
        """`count` items of `dtype` starting `offset` BYTES into the allocation."""
        out = np.empty(count, dtype)
        assert offset >= 0 and offset + out.nbytes <= self.nbytes
        L.check(self.engine.lib.mpb_memcpy_d2h(self.engine.ctx, out.ctypes.data, self.ptr + int(offset), out.nbytes))
        return out

    def free(self):
        if self.ptr:
            L.check(self.engine.lib.mpb_free(self.engine.ctx, self.ptr))
            self.ptr = None


class FilterResult:
    """Per-read outputs of one batch: ee (after +Ns / floor, i.e. what process_data returns),
    ns, passed (bool) and the batch totals."""

    def __init__(self, ee, ns, passed, n_pass, n_overflow):
        self.ee, self.ns, self.passed = ee, ns, passed
        self.n_pass, self.n_fail, self.n_overflow = n_pass, len(ee) - n_pass, n_overflow


def check_host_batch(q, lens, fixed_len, out, limit):
    """Argument checks shared by every host-batch entry (one GPU or several): -> (q, n, stride, lens, (ee, ns, ps)).
    `limit`: longest read the method takes (None: as long as the row)."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    if q.ndim != 2:
        raise ValueError("q must be a 2-D (reads x stride) uint8 matrix")
    n, stride = q.shape
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.int32)
        if lens.shape != (n,):
            raise ValueError("lens must have one entry per read")
        if n and (lens.min() < 0 or lens.max() > stride):
            raise ValueError("a length does not fit the row stride")
        if n and limit is not None and lens.max() > limit:
            raise ValueError("reads longer than %d bases are not supported (longest: %d)" % (limit, int(lens.max())))
    elif fixed_len is None:
        raise ValueError("give lens or fixed_len")
    if out is None:
        res = np.empty(n, np.float64), np.empty(n, np.int32), np.empty(n, np.uint8)
    else:                                   # caller-owned result arrays (a streaming caller reuses them: fresh
        ee, ns, ps = out                    # arrays cost a page fault per 4 KiB, ~3 ms per 100 MB of results)
        if (ee.dtype, ns.dtype, ps.dtype) != (np.float64, np.int32, np.uint8) or \
                not (len(ee) >= n and len(ns) >= n and len(ps) >= n) or \
                not (ee.flags.c_contiguous and ns.flags.c_contiguous and ps.flags.c_contiguous):
            raise ValueError("out must be contiguous (float64, int32, uint8) arrays of at least n entries")
        res = ee[:n], ns[:n], ps[:n]
    return q, n, stride, lens, res


class Engine:
    """One context on one MI355X (one per process/rank; not thread-safe)."""

    batched_only = False      # True: never take the one-read-per-wave path for small batches (tests, diagnostics)

    def __init__(self, device=0):
        self.lib = L.load()
        ctx = C.c_void_p()
        L.check(self.lib.mpb_create(int(device), C.byref(ctx)))
        self.ctx = ctx
        self.device = int(device)
        self._pinned = {}

    def close(self):
        if getattr(self, "ctx", None):
            for p in list(getattr(self, "_pinned", {}).values()):
                self.lib.mpb_host_free(self.ctx, p)
            self._pinned = {}
            self.lib.mpb_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # ---- parameters -------------------------------------------------------------------------
    @staticmethod
    def params(alpha=0.005, uncert=0.01, maxerrors=None, ambigs="treat_as_errors", round_=False,
               fast_fma=False, test_underpredict=False, decision_only=False, batched_only=False, count_cells=False,
               no_narrow=False, narrow_rows=0, narrow_split=0):
        if ambigs not in L.AMBIG:
            raise ValueError("ambigs must be one of %s" % sorted(L.AMBIG))
        flags = (L.FLAG_ROUND if round_ else 0) | (L.FLAG_FAST_FMA if fast_fma else 0) | \
                (L.FLAG_TEST_UNDERPREDICT if test_underpredict else 0) | \
                (L.FLAG_DECISION_ONLY if decision_only else 0) | \
                (L.FLAG_BATCHED_ONLY if batched_only else 0) | (L.FLAG_COUNT_CELLS if count_cells else 0) | \
                (L.FLAG_NO_NARROW if no_narrow else 0) | L.FLAG_NARROW_ROWS(narrow_rows) | ((int(narrow_split) & 255) << 12)
        return L.FilterParams(float(alpha), float(uncert),
                              math.nan if maxerrors is None else float(maxerrors),
                              L.AMBIG[ambigs], flags)

    # ---- memory -----------------------------------------------------------------------------
    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def host_alloc(self, shape, dtype=np.uint8):
        """A numpy array over PINNED host memory (mpb_host_alloc): `filter()` DMA-s such a batch from where it
        lies instead of staging it.  Release it with host_free() (or with the engine)."""
        shape = (int(shape),) if np.isscalar(shape) else tuple(int(x) for x in shape)
        dt = np.dtype(dtype)
        nbytes = int(np.prod(shape)) * dt.itemsize
        p = C.c_void_p()
        L.check(self.lib.mpb_host_alloc(self.ctx, nbytes, C.byref(p)))
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        arr = np.frombuffer(buf, dtype=dt, count=int(np.prod(shape))).reshape(shape)
        self._pinned[arr.ctypes.data] = p.value
        return arr

    def host_free(self, arr):
        p = self._pinned.pop(arr.ctypes.data, None)
        if p is not None and self.ctx:
            L.check(self.lib.mpb_host_free(self.ctx, p))

    def synchronize(self):
        L.check(self.lib.mpb_synchronize(self.ctx))

    def stream(self):
        p = C.c_void_p()
        L.check(self.lib.mpb_stream(self.ctx, C.byref(p)))
        return p.value

    # ---- packing ----------------------------------------------------------------------------
    def pack(self, seqs, quals, stride=None):
        """[(seq str|None)], [list[int]] -> (q uint8[n, stride], lens int32[n]).
        Q0 -> 1, 'N' -> 0, 'n' -> 255 (see include/moira_pb.h)."""
        n = len(quals)
        lens = np.array([len(x) for x in quals], np.int32)
        if stride is None:
            stride = _round_up(max(int(lens.max()) if n else 1, 1), 16)
        q = np.zeros((n, stride), np.uint8)
        for i in range(n):
            qi = np.ascontiguousarray(quals[i], dtype=np.int32)
            s = seqs[i] if seqs is not None else None
            if s is not None:
                if len(s) != len(qi):
                    raise ValueError("contig and contig_quals must have the same length")
                s = s.encode() if isinstance(s, str) else s
            L.check(self.lib.mpb_pack_read(s, qi.ctypes.data, len(qi), q[i].ctypes.data, stride))
        return q, lens

    def pack_batch_ascii(self, seqs, qual_strs, fastq_offset=33, stride=None, max_len=0):
        """Reads with raw FASTQ quality strings -> (q uint8[n, stride], lens): one C loop."""
        n = len(qual_strs)
        off = np.zeros(n + 1, np.int64)
        if n:
            off[1:] = np.cumsum([len(x) for x in qual_strs])
        lens_full = off[1:] - off[:-1]
        longest = int(min(lens_full.max(), max_len) if (n and max_len > 0) else (lens_full.max() if n else 1))
        if stride is None:
            stride = _round_up(max(longest, 1), 16)
        q = np.empty((n, stride), np.uint8)
        lens = np.empty(n, np.int32)
        if seqs is not None and any(len(s) != len(x) for s, x in zip(seqs, qual_strs)):
            raise ValueError("contig and contig_quals must have the same length")
        L.check(self.lib.mpb_pack_batch_ascii("".join(seqs).encode() if seqs is not None else None,
                                              "".join(qual_strs).encode("latin-1"), off.ctypes.data, n,
                                              int(fastq_offset), int(max_len), stride, q.ctypes.data,
                                              lens.ctypes.data))
        return q, lens

    def decode_ascii_device(self, d_seq, d_qual, n, stride, d_out, d_len=None, fixed_len=0, fastq_offset=33,
                            d_err=None):
        ptr = lambda b: (b.ptr if isinstance(b, DeviceBuffer) else b)
        L.check(self.lib.mpb_decode_ascii_device(self.ctx, ptr(d_seq), ptr(d_qual), n, stride,
                                                 ptr(d_len) if d_len is not None else None, int(fixed_len),
                                                 int(fastq_offset), ptr(d_out),
                                                 ptr(d_err) if d_err is not None else None))

    def encode_ascii_device(self, d_q, n, stride, d_seq_out, d_qual_out, fastq_offset=33):
        ptr = lambda b: (b.ptr if isinstance(b, DeviceBuffer) else b)
        L.check(self.lib.mpb_encode_ascii_device(self.ctx, ptr(d_q), n, stride, int(fastq_offset), ptr(d_seq_out), ptr(d_qual_out)))

    def filter_ascii_device(self, d_seq, d_qual, n, stride, d_q_out, d_len=None, fixed_len=0, fastq_offset=33,
                            d_ee=None, d_ns=None, d_pass=None, d_err=None, params=None, want_counts=True):
        """Raw FASTQ text resident in HBM -> results, classified at source: the decode pass classifies the reads, the
        filter starts at the scan (mpb_decode_classify_device + mpb_filter_device_classified)."""
        params = params or self.params()
        ptr = lambda b: (b.ptr if isinstance(b, DeviceBuffer) else b)
        opt = lambda b: (ptr(b) if b is not None else None)
        L.check(self.lib.mpb_decode_classify_device(self.ctx, ptr(d_seq), ptr(d_qual), n, stride, opt(d_len), int(fixed_len),
                                                    int(fastq_offset), C.byref(params), ptr(d_q_out), ptr(d_ee), ptr(d_ns),
                                                    ptr(d_pass), opt(d_err)))
        counts = L.FilterCounts()
        L.check(self.lib.mpb_filter_device_classified(self.ctx, ptr(d_q_out), n, stride, opt(d_len), int(fixed_len),
                                                      C.byref(params), ptr(d_ee), ptr(d_ns), ptr(d_pass),
                                                      C.byref(counts) if want_counts else None))
        return counts if want_counts else None

    # ---- the hot path -------------------------------------------------------------------------
    def filter_device(self, d_q, n, stride, d_len=None, fixed_len=0, d_ee=None, d_ns=None, d_pass=None,
                      params=None, want_counts=True):
        """Filter a batch already resident in HBM.  d_* are DeviceBuffer or raw int pointers."""
        params = params or self.params()
        ptr = lambda b: (b.ptr if isinstance(b, DeviceBuffer) else b)
        counts = L.FilterCounts()
        L.check(self.lib.mpb_filter_device(self.ctx, ptr(d_q), n, stride, ptr(d_len) if d_len is not None else None,
                                           int(fixed_len), C.byref(params), ptr(d_ee), ptr(d_ns), ptr(d_pass),
                                           C.byref(counts) if want_counts else None))
        return counts if want_counts else None

    def pack_coded(self, seqs, quals, stride=None, max_len=0):
        """Reads whose scores may exceed 254 (any non-negative int, as moira/bernoullimodule.c:92-108 takes them) ->
        (q uint8[n, stride], lens int32[n], code_scores int32[256]): every distinct score above 254 of the BATCH gets a
        byte code the batch does not use; `code_scores[c]` is what code c stands for.  ValueError (MPB_E_RANGE) when the
        batch has more such scores than free codes.  Pass code_scores on to filter()."""
        n = len(quals)
        off = np.zeros(n + 1, np.int64)
        if n:
            off[1:] = np.cumsum([len(x) for x in quals])
        if seqs is not None and any(len(s) != len(x) for s, x in zip(seqs, quals)):
            raise ValueError("contig and contig_quals must have the same length")
        lens_full = off[1:] - off[:-1]
        longest = int(min(lens_full.max(), max_len) if (n and max_len > 0) else (lens_full.max() if n else 1))
        if stride is None:
            stride = _round_up(max(longest, 1), 16)
        flat = np.concatenate([np.asarray(x, np.int64) for x in quals]) if n and off[-1] else np.empty(0, np.int64)
        if len(flat) and (flat.min() < -(1 << 31) or flat.max() >= (1 << 31)):
            raise ValueError("a quality score does not fit a C int")
        flat = np.ascontiguousarray(flat, np.int32)
        q = np.empty((n, stride), np.uint8)
        lens = np.empty(n, np.int32)
        codes = np.empty(256, np.int32)
        L.check(self.lib.mpb_pack_batch_coded("".join(seqs).encode() if seqs is not None else None, flat.ctypes.data,
                                              off.ctypes.data, n, int(max_len), stride, q.ctypes.data, lens.ctypes.data,
                                              codes.ctypes.data))
        return q, lens, codes

    def filter(self, q, lens=None, fixed_len=None, out=None, code_scores=None, **kw):
        """Filter a packed host matrix q (n x stride uint8).  Returns FilterResult.  code_scores: what pack_coded returned
        (a batch that carries scores above 254)."""
        kw.setdefault("batched_only", self.batched_only)
        params = kw.pop("params", None) or self.params(**kw)
        q, n, stride, lens, (ee, ns, ps) = check_host_batch(q, lens, fixed_len, out, limit=L.MAX_LEN)
        counts = L.FilterCounts()
        if code_scores is not None:
            code_scores = np.ascontiguousarray(code_scores, np.int32)
            if code_scores.shape != (256,):
                raise ValueError("code_scores must hold 256 entries")
        L.check(self.lib.mpb_filter_host_coded(self.ctx, q.ctypes.data, n, stride,
                                               lens.ctypes.data if lens is not None else None,
                                               0 if lens is not None else int(fixed_len), C.byref(params),
                                               code_scores.ctypes.data if code_scores is not None else None,
                                               ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, C.byref(counts)))
        return FilterResult(ee, ns, ps.view(bool), counts.n_pass, counts.n_overflow)   # pass bytes are 0 / 1

    def filter_poisson(self, q, lens=None, fixed_len=None, out=None, **kw):
        """--error_calc poisson (moira/moira.py:1637-1679): lambda summed on the GPU in base order,
        scalar CDF tail on the host with the reference's libm calls.  Returns FilterResult."""
        params = kw.pop("params", None) or self.params(**kw)
        q, n, stride, lens, (ee, ns, ps) = check_host_batch(q, lens, fixed_len, out, limit=None)
        counts = L.FilterCounts()
        L.check(self.lib.mpb_filter_poisson_host(self.ctx, q.ctypes.data, n, stride,
                                                 lens.ctypes.data if lens is not None else None,
                                                 0 if lens is not None else int(fixed_len), C.byref(params),
                                                 ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, C.byref(counts)))
        return FilterResult(ee, ns, ps.view(bool), counts.n_pass, 0)

    def calculate_errors_PB(self, contig, contig_quals, alpha):
        """Exact twin of bernoulli.calculate_errors_PB -> (expected_errors, Ns).
        ref: moira/bernoullimodule.c:66-114 (argument and error behaviour)."""
        from .broker import marshal_read               # the argument rules, shared with the broker entry
        seq, qi, addr, alpha = marshal_read(contig, contig_quals, alpha)
        ee, ns = C.c_double(), C.c_int32()
        L.check(self.lib.mpb_calculate_errors_PB(self.ctx, seq, addr, len(qi), alpha,
                                                 C.byref(ee), C.byref(ns)))
        return ee.value, ns.value

    def calculate_errors_poisson(self, sequence, quals, alpha):
        """Twin of moira.py's calculate_errors_poisson(sequence, quals, alpha) -> (expected_errors, Ns)
        (moira/moira.py:1637-1679): lambda summed on the GPU in base order, the scalar tail on the host.  Any
        non-negative int is a score; OverflowError where the Python function raises it."""
        sequence = str(sequence)
        qi = np.array([int(v) for v in quals], np.int64).astype(np.int32) if len(quals) else np.empty(0, np.int32)
        alpha = float(alpha)
        if len(sequence) != len(qi):
            raise ValueError("sequence and quals must have the same length")
        if alpha <= 0 or alpha >= 1:                           # (the bare Python function also takes alpha == 1)
            raise ValueError("Alpha must be between 0 (not included) and 1.")
        ee, ns = C.c_double(), C.c_int32()
        L.check(self.lib.mpb_calculate_errors_poisson(self.ctx, sequence.encode(), qi.ctypes.data, len(qi), alpha,
                                                      C.byref(ee), C.byref(ns)))
        if ee.value != ee.value:
            raise OverflowError("Lambda ** expected_errors or its factorial leaves the float range (moira.py:1671)")
        return ee.value, ns.value

    # ---- synthetic workload -----------------------------------------------------------------------
    def synth_fill(self, d_q, n, stride, fixed_len=0, min_len=0, max_len=0, d_len=None, seed=1, first_read=0, profile=0):
        """profile: 0 = the model of include/mpb_synth.h (BASELINE's configs), 1 = the clean run (Q33..Q40)."""
        ptr = lambda b: (b.ptr if isinstance(b, DeviceBuffer) else b)
        L.check(self.lib.mpb_synth_fill_device_profile(self.ctx, ptr(d_q), n, stride, fixed_len, min_len, max_len,
                                                       ptr(d_len) if d_len is not None else None, seed, first_read, profile))

    def last_path(self):
        """Which pass the last filter_device call took: dict(narrow_rows, narrow_split, sampled, n_fallback, sample_hist)."""
        info = L.PathInfo()
        L.check(self.lib.mpb_last_path(self.ctx, C.byref(info)))
        return {"narrow_rows": info.narrow_rows, "sampled": bool(info.sampled), "n_fallback": info.n_fallback,
                "sample_hist": list(info.sample_hist), "narrow_split": info.narrow_split}

    # ---- measurement ----------------------------------------------------------------------------
    def timing(self, on=True):
        L.check(self.lib.mpb_timing_enable(self.ctx, 1 if on else 0))

    def timing_reset(self):
        L.check(self.lib.mpb_timing_reset(self.ctx))

    def kernel_times(self):
        """{name: (total_ms, launches)} since the last reset (synchronises)."""
        out = {}
        for kid, name in L.KERNEL_NAMES.items():
            ms, cnt = C.c_double(), C.c_int64()
            L.check(self.lib.mpb_kernel_time(self.ctx, kid, C.byref(ms), C.byref(cnt)))
            out[name] = (ms.value, cnt.value)
        return out

    def device_lut(self):
        """({1-p}, {p'}) as uploaded to this device (256 doubles each; ref: moira/bernoullimodule.c:202,140-145)."""
        a, b = np.empty(256), np.empty(256)
        L.check(self.lib.mpb_device_lut(self.ctx, a.ctypes.data, b.ctypes.data))
        return a, b

    def algorithmic_cells(self):
        """DP cells the algorithm needs (sum_k min(k + 1, J) per read) for the last filter_device call made with
        params(count_cells=True)."""
        v = C.c_int64()
        L.check(self.lib.mpb_last_algorithmic_cells(self.ctx, C.byref(v)))
        return v.value

    def read_budgets(self, n):
        """Row budget (class cap) of each of the first n reads of the last filter_device call."""
        out = np.empty(n, np.int32)
        L.check(self.lib.mpb_last_read_budgets(self.ctx, out.ctypes.data, n))
        return out

    def class_histogram(self):
        caps = np.zeros(64, np.int32)
        cnts = np.zeros(64, np.int64)
        k = self.lib.mpb_last_class_histogram(self.ctx, caps.ctypes.data, cnts.ctypes.data, 64)
        if k < 0:
            L.check(k)
        return {int(caps[i]): int(cnts[i]) for i in range(k)}


def host_lut():
    """The library's Phred -> ({1-p}, {p'}) table computed on the host (no device needed)."""
    a, b = np.empty(256), np.empty(256)
    L.check(L.load().mpb_host_lut(a.ctypes.data, b.ctypes.data))
    return a, b


_default_engine = None


def default_engine():
    """Process-wide engine on device LOCAL_RANK (or 0)."""
    global _default_engine
    if _default_engine is None:
        import os
        _default_engine = Engine(int(os.environ.get("LOCAL_RANK", "0")))
    return _default_engine
