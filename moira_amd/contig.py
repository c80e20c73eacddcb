"""ctypes front-end of libmoira_contig.so (include/moira_contig.h): CPU contig construction.

Mirrors the reference's Python-level interfaces (same names, argument meaning, return shapes):
  reverse_complement(sequence, quals=None)                     moira/moira.py:1207-1235
  nw_align(seq_1, seq_2, match, mismatch, gap)                 moira/nw_align.pyx:49 / moira.py:1238
  make_contig(forward_aligned, forward_quals, reverse_aligned, reverse_quals, insert, deltaq,
              consensus_qscore, qscore_cap, trim_overlap)      moira/moira.py:1376-1558
plus contigs_batch(): the paired half of process_data for a whole chunk on all host cores.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmoira_contig.so")
CONSENSUS = {"best": 0, "sum": 1, "posterior": 2}
_lib = None


class LengthMismatchError(Exception):
    def __str__(self):
        return "Sequence and qualities are of different lengths."


def load():
    global _lib
    if _lib is None:
        from . import build as _build
        if _build.contig_stale():
            _build.build_contig()
        L = C.CDLL(LIB_PATH)
        vp, i32 = C.c_void_p, C.c_int32
        L.mct_last_error.restype = C.c_char_p
        L.mct_reverse_complement.argtypes = [C.c_char_p, vp, i32, vp, vp]
        L.mct_nw_align.argtypes = [C.c_char_p, i32, C.c_char_p, i32, i32, i32, i32, vp, vp, vp, vp]
        L.mct_nw_align_scalar.argtypes = L.mct_nw_align.argtypes
        L.mct_contigs_from_fastq.argtypes = [C.c_int64, vp, vp, vp, vp] + [i32] * 10 + [C.c_int64, vp, vp, vp, vp, vp]
        L.mct_make_contig.argtypes = [C.c_char_p, vp, C.c_char_p, vp, i32, i32, i32, i32, i32, i32,
                                      vp, vp, vp, vp, vp, vp]
        L.mct_contigs_batch.argtypes = [C.c_int64, vp, vp, vp, vp, vp, vp, i32, i32, i32, i32, i32, i32, i32,
                                        i32, i32, i32, vp, vp, vp, vp, vp, vp]
        _lib = L
    return _lib


def _check(rc):
    if rc:
        raise ValueError(load().mct_last_error().decode())


def reverse_complement(sequence, quals=None):
    L = load()
    s = str(sequence).encode()
    out = C.create_string_buffer(len(s) + 1)
    if quals:
        q = np.ascontiguousarray(list(quals), dtype=np.int32)
        if len(q) != len(s.replace(b"-", b"").replace(b".", b"")):
            raise LengthMismatchError()
        oq = np.empty(len(q), np.int32)
        _check(L.mct_reverse_complement(s, q.ctypes.data, len(s), C.addressof(out), oq.ctypes.data))
        return out.value.decode(), [int(x) for x in oq]
    _check(L.mct_reverse_complement(s, None, len(s), C.addressof(out), None))
    return out.value.decode()


def nw_align(seq_1, seq_2, match, mismatch, gap, scalar=False):
    """scalar=True: the row-by-row 32-bit loop instead of the anti-diagonal SIMD walk (same result)."""
    L = load()
    a, b = str(seq_1).encode(), str(seq_2).encode()
    n = len(a) + len(b) + 1
    o1, o2 = C.create_string_buffer(n), C.create_string_buffer(n)
    alen, score = C.c_int32(), C.c_int32()
    fn = L.mct_nw_align_scalar if scalar else L.mct_nw_align
    _check(fn(a, len(a), b, len(b), int(match), int(mismatch), int(gap),
              C.addressof(o1), C.addressof(o2), C.addressof(alen), C.addressof(score)))
    return o1.value.decode(), o2.value.decode(), score.value


def make_contig(forward_aligned, forward_quals, reverse_aligned, reverse_quals, insert, deltaq,
                consensus_qscore, qscore_cap, trim_overlap):
    L = load()
    if consensus_qscore not in CONSENSUS:
        raise ValueError('consensus_qscore must be "best", "sum" or "posterior".')
    fa, ra = str(forward_aligned).encode(), str(reverse_aligned).encode()
    fq = np.ascontiguousarray(list(forward_quals), dtype=np.int32)
    rq = np.ascontiguousarray(list(reverse_quals), dtype=np.int32)
    if len(fa.replace(b"-", b"")) != len(fq) or len(ra.replace(b"-", b"")) != len(rq):
        raise LengthMismatchError()
    if len(fa) != len(ra):
        raise ValueError("aligned reads differ in length")
    n = len(fa)
    contig = C.create_string_buffer(n + 1)
    cq = np.empty(max(n, 1), np.int32)
    clen, ov, gaps, mism = C.c_int32(), C.c_int32(), C.c_int32(), C.c_int32()
    _check(L.mct_make_contig(fa, fq.ctypes.data, ra, rq.ctypes.data, n, int(insert), int(deltaq),
                             CONSENSUS[consensus_qscore], int(qscore_cap), 1 if trim_overlap else 0,
                             C.addressof(contig), cq.ctypes.data, C.addressof(clen), C.addressof(ov),
                             C.addressof(gaps), C.addressof(mism)))
    m = clen.value
    return contig.raw[:m].decode(), [int(x) for x in cq[:m]], ov.value, gaps.value, mism.value


def contigs_batch(fwd_seqs, fwd_quals, rev_seqs, rev_quals, match=1, mismatch=-1, gap=-2, insert=20,
                  deltaq=6, consensus_qscore="best", qscore_cap=40, trim_overlap=False, threads=None):
    """Paired half of process_data (moira/moira.py:789-801) for a chunk.
    Returns (contigs list[str], quals int32[n, cap], lens, overlap, gaps, mismatches)."""
    L = load()
    n = len(fwd_seqs)
    threads = threads or usable_cpus()

    def cat(seqs, quals):
        off = np.zeros(n + 1, np.int64)
        off[1:] = np.cumsum([len(s) for s in seqs])
        s = "".join(seqs).encode()
        if n and all(hasattr(x, "offset") for x in quals):          # raw FASTQ strings: one vector op
            q = np.frombuffer("".join(x.s for x in quals).encode("latin-1"), np.uint8).astype(np.int32)
            q -= np.int32(quals[0].offset) if len({x.offset for x in quals}) == 1 else \
                np.repeat(np.array([x.offset for x in quals], np.int32), off[1:] - off[:-1])
        elif n and all(isinstance(x, np.ndarray) for x in quals):
            q = np.concatenate(quals).astype(np.int32, copy=False)
        else:
            import itertools
            q = np.fromiter(itertools.chain.from_iterable(ql.ints() if hasattr(ql, "ints") else ql for ql in quals),
                            dtype=np.int32, count=int(off[-1]))
        return s, q, off
    fs, fq, fo = cat(fwd_seqs, fwd_quals)
    rs, rq, ro = cat(rev_seqs, rev_quals)
    cap = int(max((fo[1:] - fo[:-1]) + (ro[1:] - ro[:-1]), default=0)) + 1
    contigs = np.zeros((n, cap), np.uint8)
    cq = np.zeros((n, cap), np.int32)
    clen, ov, gaps, mism = (np.zeros(n, np.int32) for _ in range(4))
    _check(L.mct_contigs_batch(n, fs, fq.ctypes.data, fo.ctypes.data, rs, rq.ctypes.data, ro.ctypes.data,
                               match, mismatch, gap, insert, deltaq, CONSENSUS[consensus_qscore],
                               qscore_cap, 1 if trim_overlap else 0, threads, cap,
                               contigs.ctypes.data, cq.ctypes.data, clen.ctypes.data, ov.ctypes.data,
                               gaps.ctypes.data, mism.ctypes.data))
    seqs = [contigs[i, :clen[i]].tobytes().decode() for i in range(n)]
    return seqs, cq, clen, ov, gaps, mism


def usable_cpus():
    """CPUs this process may actually use: the cgroup quota when there is one (a GPU box hands out 16 of its
    256 hardware threads), else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


class QualityRange(Exception):
    """A quality that the byte-level contig path cannot carry (negative, or q + offset > 255)."""


def contigs_from_fastq(fbuf, fidx, rbuf, ridx, fastq_offset=33, match=1, mismatch=-1, gap=-2, insert=20, deltaq=6,
                       consensus_qscore="best", qscore_cap=40, trim_overlap=False, threads=None):
    """Contigs of a chunk of paired records that are still FASTQ text (buffers + record indices of
    moira_amd.fastio).  Returns (cbuf uint8[n * rec_cap], cidx int64[n, 6], aux int32[n, 3]): contigs as a
    buffer + record index again (qualities as bytes q + offset), aux = overlap length, gaps, mismatches."""
    L = load()
    if consensus_qscore not in CONSENSUS:
        raise ValueError('consensus_qscore must be "best", "sum" or "posterior".')
    fidx, ridx = np.ascontiguousarray(fidx), np.ascontiguousarray(ridx)
    n = len(fidx)
    threads = threads or usable_cpus()
    rec_cap = int((fidx[:, 1] + 2 * (fidx[:, 3] + ridx[:, 3])).max()) + 8 if n else 8     # header + 2 x (l1 + l2)
    cbuf = np.empty(n * rec_cap, np.uint8)
    cidx = np.empty((n, 6), np.int64)
    aux = np.empty((3, n), np.int32)
    ptr = lambda b: b.ctypes.data if isinstance(b, np.ndarray) else b
    rc = L.mct_contigs_from_fastq(n, ptr(fbuf), fidx.ctypes.data, ptr(rbuf), ridx.ctypes.data, int(fastq_offset),
                                  match, mismatch, gap, insert, deltaq, CONSENSUS[consensus_qscore], qscore_cap,
                                  1 if trim_overlap else 0, threads, rec_cap, cbuf.ctypes.data, cidx.ctypes.data,
                                  aux[0].ctypes.data, aux[1].ctypes.data, aux[2].ctypes.data)
    if rc == -7:
        raise QualityRange(L.mct_last_error().decode())
    _check(rc)
    return cbuf, cidx, np.ascontiguousarray(aux.T)
