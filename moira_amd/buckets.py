"""Length-bucketed layout for ragged batches (BASELINE config 5: 50-600 bp reads).

A single padded matrix with stride max(len) wastes HBM bytes and DP steps on short reads.
Reads are grouped by ceil(len / 64) * 64, each bucket is its own padded matrix
with its own stride, one library call per bucket, results scattered back to read order.
"""
import numpy as np


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
        q, ln = engine.pack([seqs[i] for i in idx] if seqs is not None else None,
                            [quals[i] for i in idx], stride=int(stride))
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
