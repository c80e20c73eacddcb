"""Classified at source: raw FASTQ text resident in HBM is decoded AND classified in one
pass (mpb_decode_classify_device), the filter then starts at the scan (mpb_filter_device_classified).  Must equal
decode + mpb_filter_device + oracle bit for bit; a stale classification must never be consumed."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from moira_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


def _text_batch(rng, n, stride, max_len, qlo=0, qhi=42, p_amb=(0.03, 0.01)):
    lens = rng.integers(0, max_len + 1, n).astype(np.int32)
    pn, pl = p_amb
    rest = (1 - pn - pl) / 4
    seq = rng.choice(np.frombuffer(b"ACGTNn", np.uint8), (n, stride), p=[rest, rest, rest, rest, pn, pl])
    qual = (rng.integers(qlo, qhi, (n, stride)) + 33).astype(np.uint8)
    return seq, qual, lens


def _want(oracle, seq, qual, lens, stride):
    want = np.zeros((len(lens), stride), np.uint8)
    for i in range(len(lens)):
        s = seq[i, :lens[i]].tobytes().decode()
        want[i] = oracle.pack_read(s, [int(v) - 33 for v in qual[i, :lens[i]]], stride)
    return want


@pytest.mark.parametrize("n,stride,max_len,qlo,qhi", [(3000, 160, 150, 0, 42), (20000, 320, 300, 2, 41), (5000, 608, 600, 1, 12),
                                                       (700, 2048, 2040, 20, 41), (300, 4096, 4096, 0, 6), (1, 16, 7, 30, 40)])
def test_decode_classify_equals_decode_then_filter(eng, oracle, n, stride, max_len, qlo, qhi):
    rng = np.random.default_rng(n + stride)
    seq, qual, lens = _text_batch(rng, n, stride, max_len, qlo, qhi)
    want = _want(oracle, seq, qual, lens, stride)
    d_seq, d_qual = eng.alloc(n * stride).upload(seq), eng.alloc(n * stride).upload(qual)
    d_out = eng.alloc(n * stride).upload(np.full(n * stride, 0xAB, np.uint8))        # every byte of the row must be written
    d_len, d_err = eng.alloc(n * 4).upload(lens), eng.alloc(4).upload(np.zeros(1, np.int32))
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    for kw in (dict(), dict(alpha=0.05, ambigs="disallow"), dict(maxerrors=3.0, round_=True, ambigs="ignore")):
        prm = eng.params(**kw)
        c = eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                    d_err=d_err, params=prm)
        assert np.array_equal(d_out.download(np.uint8, n * stride).reshape(n, stride), want)
        ee, ns, ps, _ = oracle.filter_batch(want, lens=lens, threads=8, **kw)
        assert same(d_ee.download(np.float64, n), ee), kw
        assert np.array_equal(d_ns.download(np.int32, n), ns) and np.array_equal(d_pass.download(np.uint8, n), ps)
        assert c.n_pass == int(ps.sum()) and c.n_reads == n
    assert d_err.download(np.int32, 1)[0] == 0
    # fixed length through the same pair
    L = int(max_len)
    wantf = _want(oracle, seq, qual, np.full(n, L, np.int32), stride)
    eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
    ee, ns, ps, _ = oracle.filter_batch(wantf, fixed_len=L, threads=8)
    assert same(d_ee.download(np.float64, n), ee) and np.array_equal(d_pass.download(np.uint8, n), ps)
    assert np.array_equal(d_out.download(np.uint8, n * stride).reshape(n, stride), wantf)
    for b in (d_seq, d_qual, d_out, d_len, d_err, d_ee, d_ns, d_pass):
        b.free()


def test_undecodable_bytes_are_counted_as_by_the_plain_decode(eng):
    rng = np.random.default_rng(5)
    n, stride = 2000, 160
    seq, qual, lens = _text_batch(rng, n, stride, 150)
    qual[rng.integers(0, n, 50), rng.integers(0, 40, 50)] = 20          # below the offset: Q < 0
    d_seq, d_qual = eng.alloc(n * stride).upload(seq), eng.alloc(n * stride).upload(qual)
    d_a, d_b = eng.alloc(n * stride), eng.alloc(n * stride)
    d_len = eng.alloc(n * 4).upload(lens)
    e1, e2 = eng.alloc(4).upload(np.zeros(1, np.int32)), eng.alloc(4).upload(np.zeros(1, np.int32))
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.decode_ascii_device(d_seq, d_qual, n, stride, d_a, d_len=d_len, d_err=e1)
    eng.filter_ascii_device(d_seq, d_qual, n, stride, d_b, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, d_err=e2)
    assert np.array_equal(d_a.download(np.uint8, n * stride), d_b.download(np.uint8, n * stride))
    assert e1.download(np.int32, 1)[0] == e2.download(np.int32, 1)[0] > 0


def test_a_stale_classification_is_refused(eng, oracle):
    from moira_amd import _lib as L
    rng = np.random.default_rng(6)
    n, stride = 5000, 160
    seq, qual, lens = _text_batch(rng, n, stride, 150)
    d_seq, d_qual, d_out = eng.alloc(n * stride).upload(seq), eng.alloc(n * stride).upload(qual), eng.alloc(n * stride)
    d_len = eng.alloc(n * 4).upload(lens)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    prm = eng.params()
    args = (eng.ctx, d_out.ptr, n, stride, d_len.ptr, 0, C.byref(prm), d_ee.ptr, d_ns.ptr, d_pass.ptr, None)
    assert eng.lib.mpb_filter_device_classified(*args) == L.E_INVALID          # nothing was classified yet
    classify = lambda p: eng.lib.mpb_decode_classify_device(eng.ctx, d_seq.ptr, d_qual.ptr, n, stride, d_len.ptr, 0, 33,
                                                            C.byref(p), d_out.ptr, d_ee.ptr, d_ns.ptr, d_pass.ptr, None)
    assert classify(prm) == 0
    other = eng.params(alpha=0.05)                                               # other parameters: another classing
    assert eng.lib.mpb_filter_device_classified(eng.ctx, d_out.ptr, n, stride, d_len.ptr, 0, C.byref(other), d_ee.ptr,
                                                d_ns.ptr, d_pass.ptr, None) == L.E_INVALID
    assert classify(prm) == 0
    q2, _ = oracle.synth_fill(3000, 320, fixed_len=300, seed=4)
    eng.filter(q2, fixed_len=300, batched_only=True)                             # another batch rebuilds the workspace
    assert eng.lib.mpb_filter_device_classified(*args) == L.E_INVALID
    assert b"classified" in eng.lib.mpb_last_error()
    assert classify(prm) == 0 and eng.lib.mpb_filter_device_classified(*args) == 0
    assert eng.lib.mpb_filter_device_classified(*args) == L.E_INVALID          # consumed: not twice
    eng.synchronize()


def test_encode_decode_round_trip_at_config2_size(eng):
    """Size-independent property at BASELINE configs[1]'s full size: decode(encode(q)) == q for the whole 10 M x 320
    matrix, and the classified-at-source results equal mpb_filter_device's on it (flags, Ns and ee bit for bit)."""
    n, stride, L = 10_000_000, 320, 300
    d_q, d_seq, d_qual, d_out = (eng.alloc(n * stride) for _ in range(4))
    ee = [eng.alloc(n * 8), eng.alloc(n * 8)]
    ns = [eng.alloc(n * 4), eng.alloc(n * 4)]
    ps = [eng.alloc(n), eng.alloc(n)]
    try:
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=2)
        eng.encode_ascii_device(d_q, n, stride, d_seq, d_qual)
        a = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=ee[0], d_ns=ns[0], d_pass=ps[0])
        b = eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, d_ee=ee[1], d_ns=ns[1], d_pass=ps[1])
        assert (a.n_pass, a.n_fail) == (b.n_pass, b.n_fail)
        step = 2_000_000
        for lo in range(0, n, step):
            off = lo * stride
            x = DeviceSlice(eng, d_q, off, step * stride).get()
            y = DeviceSlice(eng, d_out, off, step * stride).get()
            x.reshape(step, stride)[:, L:] = 0                      # the generator leaves zeros there too; be explicit
            assert np.array_equal(x, y)
        assert same(ee[0].download(np.float64, n), ee[1].download(np.float64, n))
        assert np.array_equal(ns[0].download(np.int32, n), ns[1].download(np.int32, n))
        assert np.array_equal(ps[0].download(np.uint8, n), ps[1].download(np.uint8, n))
    finally:
        for b_ in [d_q, d_seq, d_qual, d_out] + ee + ns + ps:
            b_.free()


class DeviceSlice:
    def __init__(self, eng, buf, off, nbytes):
        self.eng, self.ptr, self.nbytes = eng, buf.ptr + off, nbytes

    def get(self):
        from moira_amd import _lib as L
        out = np.empty(self.nbytes, np.uint8)
        L.check(self.eng.lib.mpb_memcpy_d2h(self.eng.ctx, out.ctypes.data, self.ptr, self.nbytes))
        return out


def test_rows_longer_than_a_tile_are_refused_by_the_fused_pass():
    """Round 4: mpb_filter_device takes rows of up to 65536 bytes; the classify-at-source pass parks whole rows in LDS and keeps
    its 16384-byte limit -- by name, not by computing on part of a row."""
    from moira_amd.engine import Engine
    with Engine(0) as eng:
        n, stride = 4, 32768
        bufs = [eng.alloc(n * stride) for _ in range(3)] + [eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
        with pytest.raises(ValueError, match="classify-at-source pass takes rows of up to 16384"):
            eng.filter_ascii_device(bufs[0], bufs[1], n, stride, bufs[2], fixed_len=30000, d_ee=bufs[3], d_ns=bufs[4], d_pass=bufs[5])
        for b in bufs:
            b.free()
