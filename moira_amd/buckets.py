"""Length-bucketed layout for ragged batches (BASELINE config 5: 50-600 bp reads).

A single padded matrix with stride max(len) wastes HBM bytes and DP steps on short reads.
Reads are grouped by ceil(len / 64) * 64, each bucket is its own padded matrix
with its own stride, one library call per bucket, results scattered back to read order.
"""
import numpy as np


class QualStr:
    """A read's qualities kept as the raw FASTQ string (+ offset); integers only on demand.
    `ints()` gives what process_data hands on: `ord(c) - offset` (moira/moira.py:1177) with the
    Q0 -> Q1 clamp (moira/moira.py:814)."""
    __slots__ = ("s", "offset")

    def __init__(self, s, offset):
        self.s, self.offset = s, offset

    def __len__(self):
        return len(self.s)

    def __getitem__(self, sl):
        return QualStr(self.s[sl], self.offset)

    def ints(self):
        a = np.frombuffer(self.s.encode("latin-1"), np.uint8).astype(np.int32) - self.offset
        return np.maximum(a, 1).tolist()

    _tables = {}

    def qual_line(self):
        """' '.join(map(str, self.ints())) without building the integers (the .qual writer's hot spot)."""
        tab = QualStr._tables.get(self.offset)
        if tab is None:
            tab = QualStr._tables[self.offset] = [str(max(b - self.offset, 1)) for b in range(256)]
        return " ".join(map(tab.__getitem__, self.s.encode("latin-1")))

    def fastq_line(self):
        """''.join(chr(q + offset) for q in self.ints()): the input string with Q0 shown as Q1."""
        z = chr(self.offset)
        return self.s.replace(z, chr(self.offset + 1)) if z in self.s else self.s


def pack_any(engine, seqs, quals, stride):
    """Pack reads whose qualities are QualStr (one C loop) or integer sequences."""
    if quals and all(isinstance(x, QualStr) for x in quals) and len({x.offset for x in quals}) == 1:
        return engine.pack_batch_ascii(seqs, [x.s for x in quals], quals[0].offset, stride=stride)
    return engine.pack(seqs, [x.ints() if isinstance(x, QualStr) else x for x in quals], stride=stride)


def bucket_of(lens, quantum=64):
    lens = np.asarray(lens)
    return np.maximum((lens + quantum - 1) // quantum * quantum, quantum).astype(np.int64)


def filter_bucketed(engine, seqs, quals, quantum=64, method="poisson_binomial", **params):
    """Pack + filter a ragged list of reads bucket by bucket.  Returns (ee, ns, passed)."""
    run = engine.filter_poisson if method == "poisson" else engine.filter
    n = len(quals)
    lens = np.array([len(x) for x in quals], np.int64)
    strides = bucket_of(lens, quantum)
    ee = np.empty(n, np.float64)
    ns = np.empty(n, np.int32)
    passed = np.empty(n, bool)
    for stride in np.unique(strides):
        idx = np.nonzero(strides == stride)[0]
        q, ln = pack_any(engine, [seqs[i] for i in idx] if seqs is not None else None,
                         [quals[i] for i in idx], int(stride))
        r = run(q, lens=ln, **params)
        ee[idx], ns[idx], passed[idx] = r.ee, r.ns, r.passed
    return ee, ns, passed


def filter_matrix_bucketed(engine, q, lens, quantum=64, **params):
    """Same for reads that are already packed in one wide matrix: re-slice per bucket."""
    q = np.asarray(q)
    lens = np.asarray(lens, np.int32)
    n = len(lens)
    strides = np.minimum(bucket_of(lens, quantum), q.shape[1])
    ee = np.empty(n, np.float64)
    ns = np.empty(n, np.int32)
    passed = np.empty(n, bool)
    for stride in np.unique(strides):
        idx = np.nonzero(strides == stride)[0]
        sub = np.ascontiguousarray(q[idx, :int(stride)])
        if sub.shape[1] % 16:
            pad = 16 - sub.shape[1] % 16
            sub = np.pad(sub, ((0, 0), (0, pad)))
        r = engine.filter(sub, lens=lens[idx], **params)
        ee[idx], ns[idx], passed[idx] = r.ee, r.ns, r.passed
    return ee, ns, passed
