"""ctypes front-end of libmoira_io.so (include/moira_io.h): the CLI's text handling in C.

A chunk of the input file stays ONE bytes object; records are rows of an int64 index into it.
`FastqChunks` yields (buf, idx) chunks; `pack` builds the uint8 quality matrix of selected records;
`format_records` renders selected records as fasta / qual / fastq bytes.
(ref: moira/moira.py:1152-1204 parse_fastq, :842-970 write_results.)
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmoira_io.so")
E_INVALID, E_UNSUPPORTED, E_RANGE, E_SPACE = -1, -2, -3, -4
REC_OK, REC_EMPTY_SEQ, REC_EMPTY_QUAL, REC_LENGTH_MISMATCH = 0, 1, 2, 3
HDR_OFF, HDR_LEN, SEQ_OFF, SEQ_LEN, QUAL_OFF, QUAL_LEN, IDX_COLS = 0, 1, 2, 3, 4, 5, 6
FMT_FASTA, FMT_QUAL, FMT_FASTQ, FMT_NAMES, FMT_REPORT = 0, 1, 2, 3, 4
_lib = None


class Unsupported(Exception):
    """The byte-level parser met content it does not reproduce (lone CR, non-ASCII): use the line parser."""


class RecordError(Exception):
    """A record failed one of the reference's checks; .kind is REC_*, .header the normalised header."""

    def __init__(self, kind, header):
        Exception.__init__(self, kind, header)
        self.kind, self.header = kind, header


PROTOTYPES = {
    "mio_version": (C.c_char_p, []),
    "mio_last_error": (C.c_char_p, []),
    "mio_fastq_index": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mio_fastq_index_mt": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32]),
    "mio_pread_mt": (C.c_int64, [C.c_int32, C.c_int64, C.c_void_p, C.c_int64, C.c_int32]),
    "mio_inflate_create": (C.c_void_p, []),
    "mio_inflate_destroy": (None, [C.c_void_p]),
    "mio_inflate_error": (C.c_char_p, []),
    "mio_inflate_gzip": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_int64, C.c_int64,
                                     C.c_void_p, C.c_void_p]),
    "mio_crc32": (C.c_uint32, [C.c_uint32, C.c_void_p, C.c_int64]),
    "mio_bgzf_scan": (C.c_int64, [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mio_bgzf_inflate_mt": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32]),
    "mio_pack": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32,
                             C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mio_py2_hash": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "mio_collapse_create": (C.c_void_p, []),
    "mio_collapse_destroy": (None, [C.c_void_p]),
    "mio_collapse_count": (C.c_int64, [C.c_void_p]),
    "mio_collapse_set_threads": (C.c_int32, [C.c_void_p, C.c_int32]),
    "mio_collapse_add": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "mio_collapse_export": (C.c_int32, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mio_format_report": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_char_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mio_collapse_format": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_char_p,
                                        C.c_int32,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]),
    "mio_first_header_mismatch": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "mio_fasta_qual_index": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p,
                                         C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mio_format": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                               C.c_int32, C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                               C.c_void_p]),
}


def load():
    global _lib
    if _lib is None:
        from . import build as _build
        if _build.io_stale():
            _build.build_io()
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _err():
    return load().mio_last_error().decode()


_scratch = {}


def _scratch_u8(name, nbytes):
    """A grow-only uint8 work buffer per purpose: fresh pages for every chunk cost more than the loops
    that fill them.  The returned view is valid until the next call with the same name."""
    a = _scratch.get(name)
    if a is None or len(a) < nbytes:
        a = _scratch[name] = np.empty(max(nbytes, 1 << 20) * 5 // 4, np.uint8)
    return a[:nbytes]


def _ptr(buf):
    """bytes (a file chunk) or a uint8 numpy array (contigs built in C) as a C pointer argument."""
    return buf.ctypes.data if isinstance(buf, np.ndarray) else buf


def header_of(buf, row):
    """The normalised header of one index row (moira.py:1175)."""
    return bytes(buf[row[HDR_OFF]:row[HDR_OFF] + row[HDR_LEN]]).decode("ascii").replace(":", "_")


def index(buf, final, max_records, threads=1):
    """-> (idx int64[n, 6], consumed bytes, bad) ; bad is None or a RecordError for the record after the n-th.
    threads > 1: the same index built by that many threads (mio_fastq_index_mt)."""
    L = load()
    # one row more than asked for: the row after the last describes a record that failed the reference's checks
    idx = np.empty((max(max_records, 1) + 1, IDX_COLS), np.int64)
    consumed = C.c_int64(0)
    bad = C.c_int32(0)
    if threads > 1:
        n = L.mio_fastq_index_mt(_ptr(buf), len(buf), 1 if final else 0, max_records, idx.ctypes.data,
                                 C.addressof(consumed), C.addressof(bad), int(threads))
    else:
        n = L.mio_fastq_index(_ptr(buf), len(buf), 1 if final else 0, max_records, idx.ctypes.data, C.addressof(consumed),
                              C.addressof(bad))
    if n == E_UNSUPPORTED:
        raise Unsupported(_err())
    if n < 0:
        raise ValueError(_err())
    err = RecordError(bad.value, header_of(buf, idx[n])) if bad.value else None
    return idx[:n], consumed.value, err


def pack(buf, idx, sel=None, fastq_offset=33, max_len=0, lower_n_is_base=False, stride=None, reuse=False, out=None):
    """-> (q uint8[nsel, stride], lens int32[nsel], has_upper_N bool[nsel]).  reuse=True: q lives in a
    work buffer that the next pack(reuse=True) overwrites."""
    L = load()
    if sel is not None:
        sel = np.ascontiguousarray(sel, np.int64)
    n = len(sel) if sel is not None else len(idx)
    if stride is None:
        ql = idx[:, QUAL_LEN] if sel is None else idx[sel, QUAL_LEN]
        longest = int(ql.max()) if n else 1
        if max_len > 0:
            longest = min(longest, max_len)
        stride = (max(longest, 1) + 15) // 16 * 16
    if out is not None:                 # rows of a matrix the caller owns (pack_parallel)
        q, lens, flags = out
    else:
        q = _scratch_u8("pack", n * stride).reshape(n, stride) if reuse else np.empty((n, stride), np.uint8)
        lens = np.empty(n, np.int32)
        flags = np.empty(n, np.uint8)
    bad = C.c_int64(-1)
    idx = np.ascontiguousarray(idx)
    rc = L.mio_pack(_ptr(buf), idx.ctypes.data, sel.ctypes.data if sel is not None else None, n, int(fastq_offset),
                    int(max_len), 1 if lower_n_is_base else 0, stride, q.ctypes.data, lens.ctypes.data,
                    flags.ctypes.data, C.addressof(bad))
    if rc:
        raise ValueError(_err())
    return q, lens, (flags if out is not None else flags.astype(bool))


PARALLEL_MIN = 4096       # records below which splitting a call over threads is not worth it


def _parts(n, k):
    k = max(1, min(k, n))
    return [(n * t // k, n * (t + 1) // k) for t in range(k)]


def pack_parallel(pool, threads, buf, idx, sel, fastq_offset, max_len, lower_n_is_base, stride):
    """pack() with the records split over `threads` calls (the C loop releases the GIL)."""
    sel = np.ascontiguousarray(sel, np.int64)
    n = len(sel)
    if pool is None or threads <= 1 or n < PARALLEL_MIN:
        return pack(buf, idx, sel, fastq_offset, max_len, lower_n_is_base, stride, reuse=True)
    q = _scratch_u8("pack", n * stride).reshape(n, stride)
    lens, flags = np.empty(n, np.int32), np.empty(n, np.uint8)
    jobs = [pool.submit(pack, buf, idx, sel[a:b], fastq_offset, max_len, lower_n_is_base, stride, False,
                        (q[a:b], lens[a:b], flags[a:b])) for a, b in _parts(n, threads)]
    for j in jobs:
        j.result()
    return q, lens, flags.astype(bool)


def first_header_mismatch(fbuf, fidx, rbuf, ridx):
    """Position of the first pair whose header tokens differ, or -1."""
    fidx, ridx = np.ascontiguousarray(fidx), np.ascontiguousarray(ridx)
    return load().mio_first_header_mismatch(_ptr(fbuf), fidx.ctypes.data, _ptr(rbuf), ridx.ctypes.data, len(fidx))


def format_parallel(pool, threads, buf, idx, sel, kind, relabel_index=None, ee=None, label_id=None, scratch="format", **kw):
    """format_records() with the selection split over `threads` calls; returns the pieces in order.  `scratch` names
    the work buffers: callers that format side by side (one per output file) pass different names."""
    n = len(sel)
    if pool is None or threads <= 1 or n < PARALLEL_MIN:
        return [format_records(buf, idx, sel, kind, relabel_index=relabel_index, ee=ee, label_id=label_id, scratch=scratch, **kw)]
    cut = lambda a, lo, hi: None if a is None else a[lo:hi]
    jobs = [pool.submit(format_records, buf, idx, sel[a:b], kind, relabel_index=cut(relabel_index, a, b),
                        ee=cut(ee, a, b), label_id=cut(label_id, a, b), scratch="%s%d" % (scratch, t), **kw)
            for t, (a, b) in enumerate(_parts(n, threads))]
    return [j.result() for j in jobs]


def py2_hashes(buf, idx, max_len=0):
    """CPython-2.7 hash(str) of every record's (truncated) sequence, as Python ints."""
    out = np.empty(len(idx), np.uint64)
    if load().mio_py2_hash(_ptr(buf), idx.ctypes.data, len(idx), int(max_len), out.ctypes.data):
        raise ValueError(_err())
    return out.tolist()


def format_records(buf, idx, sel, kind, fastq_offset=33, max_len=0, relabel=None, relabel_index=None, ee=None,
                   labels=None, label_id=None, out_offset=None, scratch="format", clamp_q0=True):
    """Selected records as one bytes-like object (kind: FMT_FASTA / FMT_QUAL / FMT_FASTQ), valid until the
    next formatting call."""
    L = load()
    sel = np.ascontiguousarray(sel, np.int64)
    n = len(sel)
    if n == 0:
        return b""
    idx = np.ascontiguousarray(idx)
    if relabel is not None:
        relabel_index = np.ascontiguousarray(relabel_index, np.int64)
    if ee is not None:
        ee = np.ascontiguousarray(ee, np.float64)
    lab_arr = None
    if label_id is not None:
        label_id = np.ascontiguousarray(label_id, np.int32)
        enc = [s.encode() for s in labels]
        lab_arr = (C.c_char_p * len(enc))(*enc)
    L_ = np.minimum(idx[sel, SEQ_LEN], max_len) if max_len > 0 else idx[sel, SEQ_LEN]
    per = 4 if kind == FMT_QUAL else (2 if kind == FMT_FASTQ else 1)
    cap = int(L_.sum()) * per + int(idx[sel, HDR_LEN].sum()) + n * (64 + (len(relabel) if relabel else 0)
                                                                   + (max(map(len, labels)) if labels else 0))
    out = _scratch_u8(scratch, cap)
    needed = C.c_int64(0)
    args = (_ptr(buf), idx.ctypes.data, sel.ctypes.data, n, kind, int(fastq_offset),
            int(fastq_offset if out_offset is None else out_offset), 1 if clamp_q0 else 0, int(max_len),
            relabel.encode() if relabel is not None else None,
            relabel_index.ctypes.data if relabel is not None else None,
            ee.ctypes.data if ee is not None else None,
            C.cast(lab_arr, C.c_void_p) if lab_arr is not None else None,
            label_id.ctypes.data if label_id is not None else None)
    w = L.mio_format(*args, out.ctypes.data, cap, C.addressof(needed))
    if w == E_SPACE:
        cap = needed.value
        out = _scratch_u8(scratch, cap)
        w = L.mio_format(*args, out.ctypes.data, cap, C.addressof(needed))
    if w < 0:
        raise ValueError(_err())
    return memoryview(out)[:w]


def format_report(buf, idx, sel, aux, relabel=None, relabel_index=None, ee=None):
    """contigs.report lines of records written one by one (n_seqs = 1); aux int32[n_records, 3]."""
    L = load()
    sel = np.ascontiguousarray(sel, np.int64)
    n = len(sel)
    if n == 0:
        return b""
    idx = np.ascontiguousarray(idx)
    aux = np.ascontiguousarray(aux, np.int32)
    if relabel is not None:
        relabel_index = np.ascontiguousarray(relabel_index, np.int64)
    if ee is not None:
        ee = np.ascontiguousarray(ee, np.float64)
    cap = int(idx[sel, HDR_LEN].sum()) + n * (96 + (len(relabel) if relabel else 0))
    out = np.empty(cap, np.uint8)
    needed = C.c_int64(0)
    w = L.mio_format_report(_ptr(buf), idx.ctypes.data, sel.ctypes.data, n, relabel.encode() if relabel else None,
                            relabel_index.ctypes.data if relabel is not None else None,
                            ee.ctypes.data if ee is not None else None, aux.ctypes.data, out.ctypes.data, cap,
                            C.addressof(needed))
    if w < 0:
        raise ValueError(_err())
    return memoryview(out)[:w]


class Collapse:
    """The reference's `uniques` dict (moira/moira.py:459-475) kept in C: add chunks, export the groups
    in output order, format them."""

    def __init__(self, threads=1):
        self.lib = load()
        self.h = self.lib.mio_collapse_create()
        if not self.h:
            raise MemoryError("mio_collapse_create failed")
        self.lib.mio_collapse_set_threads(self.h, int(threads))

    def close(self):
        if self.h:
            self.lib.mio_collapse_destroy(self.h)
            self.h = None

    __del__ = close

    def __len__(self):
        return self.lib.mio_collapse_count(self.h)

    def add(self, buf, idx, ee, flags, max_len=0, aux=None):
        idx = np.ascontiguousarray(idx)
        ee = np.ascontiguousarray(ee, np.float64)
        flags = np.ascontiguousarray(flags, np.uint8)
        if aux is not None:
            aux = np.ascontiguousarray(aux, np.int32)
        if self.lib.mio_collapse_add(self.h, _ptr(buf), idx.ctypes.data, len(idx), int(max_len), ee.ctypes.data,
                                     flags.ctypes.data, aux.ctypes.data if aux is not None else None):
            raise ValueError(_err())

    def export(self):
        """-> (ee, length, abundance, has_upper_N, aux[., 3]) per group, in output order."""
        n = len(self)
        ee, ln, size, fl = np.empty(n), np.empty(n, np.int64), np.empty(n, np.int64), np.empty(n, np.uint8)
        aux = np.empty((n, 3), np.int32)
        if self.lib.mio_collapse_export(self.h, ee.ctypes.data, ln.ctypes.data, size.ctypes.data, fl.ctypes.data,
                                        aux.ctypes.data):
            raise ValueError(_err())
        self._len, self._size = ln, size
        return ee, ln, size, fl.astype(bool), aux

    def format(self, sel, kind, fastq_offset=33, relabel=None, usearch=False, labels=None, label_id=None,
               lstrip_gt=None, out_offset=None, clamp_q0=True, scratch="format"):
        sel = np.ascontiguousarray(sel, np.int64)
        n = len(sel)
        if n == 0:
            return b""
        lab_arr = None
        if label_id is not None:
            label_id = np.ascontiguousarray(label_id, np.int32)
            enc = [s.encode() for s in labels]
            lab_arr = (C.c_char_p * len(enc))(*enc)
        if lstrip_gt is not None:
            lstrip_gt = np.ascontiguousarray(lstrip_gt, np.uint8)
        args = (self.h, sel.ctypes.data, n, kind, int(fastq_offset),
                int(fastq_offset if out_offset is None else out_offset), 1 if clamp_q0 else 0,
                relabel.encode() if relabel else None,
                1 if usearch else 0, C.cast(lab_arr, C.c_void_p) if lab_arr is not None else None,
                label_id.ctypes.data if label_id is not None else None,
                lstrip_gt.ctypes.data if lstrip_gt is not None else None)
        needed = C.c_int64(0)
        cap = int(self._len[sel].sum()) * (4 if kind == FMT_QUAL else 2) + n * 160
        out = _scratch_u8(scratch, cap)
        w = self.lib.mio_collapse_format(*args, out.ctypes.data, cap, C.addressof(needed))
        if w == E_SPACE:
            cap = needed.value
            out = _scratch_u8(scratch, cap)
            w = self.lib.mio_collapse_format(*args, out.ctypes.data, cap, C.addressof(needed))
        if w < 0:
            raise ValueError(_err())
        return memoryview(out)[:w]


def collapse_format_parallel(pool, threads, groups, sel, kind, label_id=None, lstrip_gt=None, scratch="format", **kw):
    """Collapse.format() with the selection split over `threads` calls (the object is only read).  `scratch` names the
    work buffers: callers that format several files side by side pass different names."""
    n = len(sel)
    if pool is None or threads <= 1 or n < PARALLEL_MIN:
        return [groups.format(sel, kind, label_id=label_id, lstrip_gt=lstrip_gt, scratch=scratch, **kw)]
    cut = lambda a, lo, hi: None if a is None else a[lo:hi]
    jobs = [pool.submit(groups.format, sel[a:b], kind, label_id=cut(label_id, a, b), lstrip_gt=cut(lstrip_gt, a, b),
                        scratch="%s%d" % (scratch, t), **kw) for t, (a, b) in enumerate(_parts(n, threads))]
    return [j.result() for j in jobs]


class GzipReader:
    """read(n) over a gzip file, decoded by libmoira_io's own inflate (csrc/inflate.cpp) instead of zlib: about three
    times the text rate, which is what a run on compressed input is bounded by (one stream per file cannot be split).
    `raw` is the compressed file opened in binary mode.  Multi-member files and zero padding are read as gzip reads
    them; CRC-32 and length of every member are checked; corrupt or truncated input raises OSError."""
    WINDOW = 32768
    MAX_BGZF = 1 << 16                                     # members per parallel batch

    def __init__(self, raw, in_block=1 << 23, threads=1):
        self.L = load()
        self.threads = max(1, int(threads))
        # BGZF members (bgzip, Illumina's writers, this package's own outputs) state their size, so a batch of them
        # inflates on `threads` threads; the first member that is not BGZF switches to the one-stream decoder for good
        self.bgzf = self.threads > 1 and in_block >= (1 << 17)
        self.bgzf_batches = 0
        self.raw = raw
        self.st = self.L.mio_inflate_create()
        if not self.st:
            raise MemoryError("mio_inflate_create failed")
        self.cap = max(int(in_block), 4096)
        self.inbuf = np.empty(self.cap + 8, np.uint8)
        self.in_pos = self.in_len = 0
        self.eof_in = self.done = False
        self.hist = np.empty(0, np.uint8)
        self.ready = b""

    def _more_input(self):
        """Unconsumed tail to the front, then as much new data as fits.  False when the file had no more."""
        tail = self.in_len - self.in_pos
        if tail and self.in_pos:
            self.inbuf[:tail] = self.inbuf[self.in_pos:self.in_len].copy()
        self.in_pos, self.in_len = 0, tail
        if self.eof_in:
            return False
        if self.in_len == self.cap:                       # a full buffer that still needs input cannot happen (a symbol is < 48 bits)
            raise OSError("gzip input: a header (FNAME / FCOMMENT / FEXTRA) or block header does not fit the %d-byte input "
                          "block" % self.cap)
        got = self.raw.readinto(memoryview(self.inbuf)[self.in_len:self.cap])
        if not got:
            self.eof_in = True
            return False
        self.in_len += got
        return True

    def read(self, n=1 << 25):
        """At most n bytes, at least one unless the file has ended (b""): a request is served from what one decoding
        step produced -- n bytes from the one-stream decoder, the whole members that fit n from a BGZF batch."""
        if n <= 0:
            return b""
        if not self.ready and not self.done:
            # the decoder needs room for a whole match (258 bytes) to make progress: small requests are served from a
            # block decoded ahead
            self.ready = self._decode(max(n, 1 << 16))
        if len(self.ready) <= n:
            out, self.ready = self.ready, b""                  # (no copy: the usual case for block-sized requests)
        else:
            out, self.ready = self.ready[:n], self.ready[n:]
        return out

    def _decode_bgzf(self, n):
        """Up to about n bytes from whole BGZF members, or None once the input is not (or no longer) BGZF."""
        if not hasattr(self, "_bg"):
            m = self.MAX_BGZF
            self._bg = (np.empty(m, np.int64), np.empty(m, np.int32), np.empty(m + 1, np.int64), C.c_int32(0))
        offs, sizes, out_offs, why = self._bg
        while True:
            if self.in_pos == self.in_len and not self._more_input():
                if not self.bgzf_batches:                      # no member at all: the serial decoder's error
                    self.bgzf = False
                    return None
                self.done = True                               # ended on a member boundary
                return b""
            nb = self.L.mio_bgzf_scan(self.inbuf.ctypes.data + self.in_pos, self.in_len - self.in_pos, len(offs), n,
                                      offs.ctypes.data, sizes.ctypes.data, out_offs.ctypes.data, C.addressof(why))
            if nb < 0:
                raise OSError("gzip input: " + self.L.mio_inflate_error().decode())
            if nb > 0:
                break
            if why.value == 0 and self._more_input():
                continue                                       # the member at the end of the buffer was incomplete
            self.bgzf = False                                  # not BGZF, or a truncated tail: the serial decoder says which
            while self.bgzf_batches:                           # zero padding after a member is ignored, as gzip does
                view = self.inbuf[self.in_pos:self.in_len]
                self.in_pos += int(np.argmax(view != 0)) if view.any() else len(view)
                if self.in_pos < self.in_len:
                    break
                if not self._more_input():
                    self.done = True
                    return b""
            return None
        total = int(out_offs[nb])
        out = np.empty(max(total, 1), np.uint8)
        rc = self.L.mio_bgzf_inflate_mt(self.inbuf.ctypes.data + self.in_pos, offs.ctypes.data, sizes.ctypes.data,
                                        out_offs.ctypes.data, nb, out.ctypes.data, self.threads)
        if rc < 0:
            raise OSError("gzip input: " + self.L.mio_inflate_error().decode())
        self.in_pos += int(offs[nb - 1]) + int(sizes[nb - 1])
        self.bgzf_batches += 1
        return out[:total].tobytes()

    def _decode(self, n):
        while self.bgzf:
            data = self._decode_bgzf(n)
            if data is None:
                break
            if data or self.done:
                return data                                    # (an empty member, e.g. the end marker, yields nothing: next)
        h = len(self.hist)
        out = np.empty(h + n, np.uint8)
        out[:h] = self.hist
        produced = 0
        used, made = C.c_int64(0), C.c_int64(0)
        while produced < n:
            if self.in_pos == self.in_len and not self.eof_in:
                self._more_input()
            was_final = self.eof_in                            # did THIS call tell the decoder that the input ends here?
            rc = self.L.mio_inflate_gzip(self.st, self.inbuf.ctypes.data + self.in_pos, self.in_len - self.in_pos,
                                         1 if was_final else 0, out.ctypes.data, h + produced, h + n,
                                         C.addressof(used), C.addressof(made))
            if rc < 0:
                raise OSError("gzip input: " + self.L.mio_inflate_error().decode())
            self.in_pos += used.value
            produced += made.value
            if rc == 2:
                self.done = True
                break
            if rc == 1:
                break
            # rc == 0: the decoder wants more input.  Without `final` it may stop short of a block header, a gzip header
            # or a trailer it cannot see whole (include/moira_io.h: zero progress is legal until final != 0), so "no more
            # input and no progress" only means a truncated file when the call already ran with final = 1; when the end of
            # the file has just been discovered, the decoder is called once more, with final = 1
            if not self._more_input() and used.value == 0 and made.value == 0 and was_final:
                raise OSError("gzip input: truncated file")
        keep = min(self.WINDOW, h + produced)
        self.hist = out[h + produced - keep:h + produced].copy()
        return out[h:h + produced].tobytes()

    def close(self):
        if self.st:
            self.L.mio_inflate_destroy(self.st)
            self.st = None
        self.raw.close()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def __del__(self):
        try:
            if self.st:
                self.L.mio_inflate_destroy(self.st)
                self.st = None
        except Exception:
            pass


def _plain_file_fd(fh):
    """File descriptor of an uncompressed regular file opened for reading, else None."""
    import io
    import stat
    if not isinstance(fh, (io.BufferedReader, io.FileIO)):
        return None
    try:
        fd = fh.fileno()
        return fd if stat.S_ISREG(os.fstat(fd).st_mode) else None
    except (OSError, ValueError, io.UnsupportedOperation):
        return None


class FastqChunks:
    """Iterate a binary FASTQ stream as (buf, idx) chunks of at most `max_records` records.
    Raises Unsupported before anything has been yielded from a chunk the C parser cannot take.
    threads > 1: the index of a block is built by that many threads, and an uncompressed regular file is read with
    that many concurrent preads into a fresh buffer per block (buf is then a uint8 array; no block is ever copied or
    concatenated, only the partial record at its end is carried over).  Compressed input stays one decompressing
    stream, read block by block."""

    def __init__(self, fh, max_records, block_bytes=None, threads=1):
        self.fh, self.max_records, self.threads = fh, max_records, max(1, int(threads))
        self.block = block_bytes or (1 << 25)
        # an uncompressed file read by several threads: 128 MiB, so that a chunk of 262144 x 250-bp records is one block
        self.plain_block = block_bytes or (1 << 27)

    def _iter_plain(self, fd):
        L = load()
        size = os.fstat(fd).st_size
        off = self.fh.tell()
        block = self.plain_block
        carry = np.empty(0, np.uint8)
        while True:
            want = max(0, min(block, size - off))
            buf = np.empty(len(carry) + want, np.uint8)
            buf[:len(carry)] = carry
            got = L.mio_pread_mt(fd, off, buf.ctypes.data + len(carry), want, self.threads) if want else 0
            if got < 0:
                raise OSError(_err())
            off += got
            eof = got < want or off >= size
            data = buf[:len(carry) + got]
            pos = 0
            while pos < len(data):
                view = data[pos:]
                idx, consumed, err = index(view, eof, self.max_records, self.threads)
                if len(idx):
                    yield view, idx
                if err is not None:
                    raise err
                pos += consumed
                if len(idx) < self.max_records:
                    break                                   # the rest is an incomplete record: read on
            if eof:
                return                                      # trailing lines that do not make a record are dropped
            carry = data[pos:].copy()

    def __iter__(self):
        fd = _plain_file_fd(self.fh) if self.threads > 1 else None
        if fd is not None:
            yield from self._iter_plain(fd)
            return
        tail = b""
        eof = False
        while not eof or tail:
            if not eof:
                more = self.fh.read(self.block)
                if more:
                    tail = tail + more if tail else more
                else:
                    eof = True
            while tail:
                idx, consumed, err = index(tail, eof, self.max_records, self.threads)
                if len(idx):
                    yield tail, idx
                if err is not None:
                    raise err
                tail = tail[consumed:]
                if len(idx) < self.max_records:
                    break                       # the rest is an incomplete record: read on
            if eof:
                return                          # trailing lines that do not make a record are dropped


class PairedRecordError(Exception):
    """RecordError of file `which` (0 forward, 1 reverse) in a paired run."""

    def __init__(self, which, err):
        Exception.__init__(self, which, err)
        self.which, self.err = which, err


class PairedFastqChunks:
    """Two FASTQ streams in lockstep: (fbuf, fidx, rbuf, ridx) with the same number of records each.
    Stops with the shorter file, as zip() does in the reference's parser (moira/moira.py:1158-1160); a
    record that fails the reference's checks raises only when its pair is reached, forward file first."""

    def __init__(self, ffh, rfh, max_records, block_bytes=1 << 24, threads=1):
        self.fh, self.max_records, self.block, self.threads = (ffh, rfh), max_records, block_bytes, max(1, int(threads))

    def __iter__(self):
        tail, eof, want = [b"", b""], [False, False], [True, True]
        while True:
            for k in (0, 1):
                if want[k] and not eof[k]:
                    more = self.fh[k].read(self.block)
                    if more:
                        tail[k] = tail[k] + more if tail[k] else more
                    else:
                        eof[k] = True
            got = [index(tail[k], eof[k], self.max_records, self.threads) for k in (0, 1)]
            n = min(len(got[0][0]), len(got[1][0]))
            # does file k hold a complete record number n (sound or not)?
            has_next = [len(got[k][0]) > n or got[k][2] is not None for k in (0, 1)]
            err = None
            if has_next[0] and has_next[1]:
                for k in (0, 1):
                    if err is None and got[k][2] is not None and len(got[k][0]) == n:
                        err = PairedRecordError(k, got[k][2])
                if err is not None and err.which == 1:       # then forward record n is sound: the reference names it
                    err.err.forward_header = header_of(tail[0], got[0][0][n])
            if n:
                for k in (0, 1):
                    if len(got[k][0]) > n:                   # keep what the other file has not reached yet
                        got[k] = index(tail[k], eof[k], n, self.threads)
                yield tail[0], got[0][0], tail[1], got[1][0]
                tail = [tail[k][got[k][1]:] for k in (0, 1)]
            if err is not None:
                raise err
            if n == self.max_records:
                want = [len(tail[k]) < self.block for k in (0, 1)]
                continue
            if any(not has_next[k] and eof[k] for k in (0, 1)):
                return                                       # one file is exhausted: zip() stops here
            want = [not has_next[k] for k in (0, 1)]


class FastaQualChunks:
    """A fasta file and its qual file as (buf, idx) chunks of exactly `max_records` records (fewer only at
    the end): buf is a uint8 array built by mio_fasta_qual_index (header | sequence | one byte per quality,
    FASTQ offset 0).  Raises Unsupported for anything the line parser treats specially."""

    def __init__(self, ffh, qfh, max_records, block_bytes=1 << 24):
        self.fh, self.max_records, self.block = (ffh, qfh), max_records, block_bytes

    def __iter__(self):
        L = load()
        tail, eof = [b"", b""], [False, False]
        out = idx = None
        have = used = 0
        starved = True
        while True:
            for k in (0, 1):
                if not eof[k] and (starved or len(tail[k]) < self.block):
                    more = self.fh[k].read(self.block)
                    if more:
                        tail[k] = tail[k] + more if tail[k] else more
                    else:
                        eof[k] = True
            final = eof[0] and eof[1]
            if out is None:
                out = np.empty(self.max_records * 64 + 4 * self.block, np.uint8)
                idx = np.empty((self.max_records, IDX_COLS), np.int64)
                have = used = 0
            fc, qc, ou = C.c_int64(0), C.c_int64(0), C.c_int64(0)
            n = L.mio_fasta_qual_index(tail[0], len(tail[0]), tail[1], len(tail[1]), 1 if final else 0,
                                       self.max_records - have, out.ctypes.data + used, len(out) - used,
                                       idx.ctypes.data + have * IDX_COLS * 8, C.addressof(fc), C.addressof(qc),
                                       C.addressof(ou))
            if n == E_UNSUPPORTED:
                raise Unsupported(_err())
            if n < 0:
                raise ValueError(_err())
            idx[have:have + n, [HDR_OFF, SEQ_OFF, QUAL_OFF]] += used      # offsets were relative to this call's slice
            have += n
            used += ou.value
            tail = [tail[0][fc.value:], tail[1][qc.value:]]
            starved = n == 0                                             # nothing complete in what is buffered
            full = have == self.max_records
            if full or (final and n == 0):
                if have:
                    yield out, idx[:have]
                out = None
                if final and not full:
                    return
            elif n == 0 and not final and len(out) - used < 2 * self.block:
                bigger = np.empty(2 * len(out), np.uint8)                 # records longer than expected: more room
                bigger[:used] = out[:used]
                out = bigger
