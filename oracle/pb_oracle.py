"""ctypes front-end of the CPU oracle (oracle/pb_oracle.c) + a pure-Python restatement.

TEST INFRASTRUCTURE ONLY: importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg, never from the moira_amd package.

`ee_python()` restates moira's Python twin (ref: moira/moira.py:1561-1634,
1723-1733) with plain loops for small cases; the C library is the fast oracle.
"""
import ctypes as C
import math
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libpb_oracle.so")
REF_DIR = os.path.join(_HERE, "_ref")

AMBIG = {"treat_as_errors": 0, "ignore": 1, "disallow": 2}
FLAG_ROUND = 1


class Params(C.Structure):
    _fields_ = [("alpha", C.c_double), ("uncert", C.c_double), ("maxerrors", C.c_double),
                ("ambig_mode", C.c_int32), ("flags", C.c_uint32)]


def build(force=False):
    """Compile the oracle (and, where /root/reference exists, oracle/_ref)."""
    if force or not os.path.exists(_LIB_PATH) or \
            os.path.getmtime(_LIB_PATH) < os.path.getmtime(os.path.join(_HERE, "pb_oracle.c")):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle"])
    if os.path.exists("/root/reference/moira/bernoullimodule.c") and \
            (force or not os.path.exists(os.path.join(REF_DIR, "bernoulli.so"))):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        i32p, u8p, dp = C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
        for name in ("pbo_ee_refshape", "pbo_ee_rowwise"):
            f = getattr(L, name)
            f.restype = C.c_int
            f.argtypes = [C.c_char_p, i32p, C.c_int32, C.c_double, dp, i32p, i32p]
        L.pbo_filter_batch.restype = C.c_int
        L.pbo_filter_batch.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int32,
                                       C.POINTER(Params), C.c_int, C.c_int,
                                       C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.pbo_synth_fill.restype = None
        L.pbo_synth_fill.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                     C.c_int32, C.c_void_p, C.c_uint64, C.c_int64]
        L.pbo_synth_fill_profile.restype = None
        L.pbo_synth_fill_profile.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                             C.c_int32, C.c_void_p, C.c_uint64, C.c_int64, C.c_int32]
        L.pbo_pack_read.restype = C.c_int
        L.pbo_pack_read.argtypes = [C.c_char_p, i32p, C.c_int32, u8p, C.c_int32]
        L.pbo_build_lut.restype = None
        L.pbo_build_lut.argtypes = [dp, dp]
        L.pbo_max_threads.restype = C.c_int
        _lib = L
    return _lib


def reference_module():
    """The REAL reference extension (oracle/_ref/bernoulli.so), or None if it was never built."""
    path = os.path.join(REF_DIR, "bernoulli.so")
    if not os.path.exists(path):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("bernoulli", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def _ee(fn, seq, quals, alpha):
    q = np.ascontiguousarray(quals, dtype=np.int32)
    s = seq.encode() if isinstance(seq, str) else seq
    if s is not None and len(s) != len(q):
        raise ValueError("contig and contig_quals must have the same length")
    ee, ns, rows = C.c_double(), C.c_int32(), C.c_int32()
    rc = fn(s, q.ctypes.data_as(C.POINTER(C.c_int32)), len(q), alpha,
            C.byref(ee), C.byref(ns), C.byref(rows))
    if rc != 0:
        raise ValueError("oracle rc=%d" % rc)
    return ee.value, ns.value, rows.value


def ee_refshape(seq, quals, alpha):
    """(ee, Ns, rows) by the reference-shaped loop nest."""
    return _ee(lib().pbo_ee_refshape, seq, quals, alpha)


def ee_rowwise(seq, quals, alpha):
    """(ee, Ns, rows) by the two-term recurrence."""
    return _ee(lib().pbo_ee_rowwise, seq, quals, alpha)


def make_params(alpha=0.005, uncert=0.01, maxerrors=None, ambigs="treat_as_errors", round_=False):
    return Params(alpha, uncert, float("nan") if maxerrors is None else float(maxerrors),
                  AMBIG[ambigs], FLAG_ROUND if round_ else 0)


def filter_batch(q, lens=None, fixed_len=None, shape=0, threads=1, **kw):
    """Oracle over a packed (n x stride) uint8 matrix -> (ee, ns, pass, rows)."""
    q = np.ascontiguousarray(q, dtype=np.uint8)
    n, stride = q.shape
    if lens is not None:
        lens = np.ascontiguousarray(lens, dtype=np.int32)
    ee = np.empty(n, np.float64)
    ns = np.empty(n, np.int32)
    ps = np.empty(n, np.uint8)
    rows = np.empty(n, np.int32)
    prm = make_params(**kw)
    rc = lib().pbo_filter_batch(q.ctypes.data, n, stride,
                                lens.ctypes.data if lens is not None else None,
                                0 if fixed_len is None else int(fixed_len),
                                C.byref(prm), shape, threads,
                                ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, rows.ctypes.data)
    if rc != 0:
        raise ValueError("oracle rc=%d" % rc)
    return ee, ns, ps, rows


def synth_fill(n, stride, fixed_len=0, min_len=0, max_len=0, seed=1, first_read=0, profile=0):
    """Host twin of the device synthetic generator -> (q, lens).  profile: include/mpb_synth.h (0 = BASELINE's model,
    1 = the clean run, Q33..Q40)."""
    q = np.empty((n, stride), np.uint8)
    lens = np.empty(n, np.int32)
    lib().pbo_synth_fill_profile(q.ctypes.data, n, stride, fixed_len, min_len, max_len,
                                 lens.ctypes.data, seed, first_read, profile)
    return q, lens


def pack_read(seq, quals, row_bytes):
    q = np.ascontiguousarray(quals, dtype=np.int32)
    row = np.empty(row_bytes, np.uint8)
    s = seq.encode() if isinstance(seq, str) else seq
    rc = lib().pbo_pack_read(s, q.ctypes.data_as(C.POINTER(C.c_int32)), len(q),
                             row.ctypes.data_as(C.POINTER(C.c_uint8)), row_bytes)
    if rc != 0:
        raise ValueError("oracle pack rc=%d" % rc)
    return row


def lut():
    a = np.empty(256)
    b = np.empty(256)
    dp = C.POINTER(C.c_double)
    lib().pbo_build_lut(a.ctypes.data_as(dp), b.ctypes.data_as(dp))
    return a, b


def ee_python(seq, quals, alpha):
    """Pure-Python restatement (small cases only), ref: moira/moira.py:1561-1634.

    Follows the C reference in counting 'n' as ambiguous (bernoullimodule.c:196)
    and clamping Q0 to 1 (:104-107); follows the Python twin where C is undefined
    (leading 0 in the CDF list, moira.py:1611)."""
    probs = []
    n_amb = 0
    for base, q in zip(seq, quals):
        if q < 0:
            raise ValueError("Qualities must have positive values.")
        if base in "Nn":
            n_amb += 1
            continue
        probs.append(10 ** ((q if q else 1) / -10.0))
    if not probs:
        return 0.0, n_amb
    a = [(1 - p) ** 1 for p in probs]
    b = [((1 - 1 + 1) / float(1)) * (p / (1 - p)) * ((1 - p) ** 1) for p in probs]
    acc = [0.0]
    prev = None
    j = 0
    thr = 1 - alpha
    while True:
        cur = [0.0] * len(probs)
        cur[0] = a[0] if j == 0 else (b[0] if j == 1 else 0.0)
        for k in range(1, len(probs)):
            s = 0 + a[k] * cur[k - 1]
            if j >= 1:
                s = s + b[k] * prev[k - 1]
            cur[k] = s
        acc.append(acc[-1] + cur[-1] if j else cur[-1])
        if acc[-1] > thr:
            break
        j += 1
        prev = cur
        if j > len(probs):
            return math.nan, n_amb
    r = (j - 1) + ((j - (j - 1)) * (thr - acc[-2]) / (acc[-1] - acc[-2]))
    return (0.0 if r < 0 else r), n_amb
