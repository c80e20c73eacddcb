"""--error_calc poisson (SURVEY f-3): exact host tail on CPU; lambda reduction on the GPU."""
import ctypes as C
import math

import numpy as np
import pytest

import golden_io as G
from moira_amd import _lib as L
from poisson_ref import calculate_errors_poisson          # oracle/: the reference's formula restated


def finish(lams, ns, lens, **kw):
    lib = L.load()
    from moira_amd.engine import Engine
    prm = Engine.params(**kw)
    lams = np.ascontiguousarray(lams, np.float64)
    ns = np.ascontiguousarray(ns, np.int32)
    lens = np.ascontiguousarray(lens, np.int32)
    ee = np.empty(len(lams))
    ps = np.empty(len(lams), np.uint8)
    L.check(lib.mpb_poisson_finish_host(lams.ctypes.data, ns.ctypes.data, lens.ctypes.data, 0, len(lams),
                                        C.byref(prm), ee.ctypes.data, ps.ctypes.data))
    return ee, ps


def py_lambda(seq, quals):
    lam = 0
    for b, q in zip(seq, quals):
        if b != "N":
            lam += 10 ** (q / -10.0)
    return float(lam)


def test_poisson_kat_host_tail():
    k = G.load_kat()["kat1"]
    assert calculate_errors_poisson(k["seq"], k["quals"], 0.005) == (6.932519986616133, 0)   # test_moira.py:45
    ee, _ = finish([py_lambda(k["seq"], k["quals"])], [0], [len(k["seq"])], alpha=0.005, ambigs="ignore")
    assert ee[0] == 6.932519986616133


def test_poisson_tail_matches_python_on_random_lambdas():
    rng = np.random.default_rng(4)
    lams = np.concatenate([[0.0, 1e-300, 1e-9, 0.005, 0.0050125], rng.uniform(0, 5, 500), rng.uniform(5, 120, 300)])
    for alpha in (0.005, 0.05, 1e-4):
        ee, _ = finish(lams, np.zeros(len(lams)), np.full(len(lams), 300), alpha=alpha, ambigs="ignore")
        for lam, e in zip(lams, ee):
            lam = float(lam)                      # Python float semantics (np.float64 ** int gives inf, not an error)
            acc, j = [0], 0
            try:
                while True:
                    acc.append(acc[-1] + (math.exp(-lam) * (lam ** j)) / math.factorial(j))
                    if acc[-1] > (1 - alpha):
                        break
                    j += 1
            except OverflowError:                 # the reference crashes here; the library reports NaN
                assert math.isnan(e), (lam, alpha)
                continue
            want = (j - 1) + ((j - (j - 1)) * ((1 - alpha) - acc[-2]) / (acc[-1] - acc[-2]))
            assert e == (0 if want < 0 else want), (lam, alpha)


@pytest.mark.gpu
def test_poisson_gpu_matches_python_reference_function():
    from moira_amd.engine import Engine
    s = G.load_set("rand_mixed")
    q, lens = s["q"][:600].copy(), s["lens"][:600]
    q[q == 255] = 17                       # the Python reference scores 'n' as a normal base
    with Engine(0) as eng:
        r = eng.filter_poisson(q, lens=lens, alpha=0.005, ambigs="treat_as_errors")
        with pytest.raises(ValueError, match="255"):
            eng.filter_poisson(s["q"][:600], lens=lens)
    for i in range(600):
        row = q[i, :lens[i]]
        seq = "".join("N" if v == 0 else "A" for v in row)
        quals = [20 if v == 0 else int(v) for v in row]
        e, ns = calculate_errors_poisson(seq, quals, 0.005)
        assert r.ee[i] == e + ns and r.ns[i] == ns
        assert bool(r.passed[i]) == (e + ns <= lens[i] * 0.01)


def _fixture_sets():
    z = G.load_set("poisson")
    return z["alphas"], [(z["q_" + t], z["lens_" + t], z["ee_" + t], z["ns_" + t], z["ovf_" + t]) for t in ("a", "b")]


def test_poisson_host_tail_against_reference_fixture():
    """tests/golden/poisson.npz: moira.py's own calculate_errors_poisson (moira/moira.py:1637-1679, imported by
    tests/golden/make_golden.py) on 2,300 reads x 3 alphas, incl. the long low-quality reads where it raises
    OverflowError (the library reports NaN there).  lambda is summed here exactly as the reference sums it."""
    alphas, sets = _fixture_sets()
    n_cases = n_ovf = 0
    for q, lens, ee_ref, ns_ref, ovf in sets:
        lams = np.array([py_lambda(["N" if v == 0 else "A" for v in q[i, :lens[i]]], [int(v) for v in q[i, :lens[i]]])
                         for i in range(len(lens))])
        for ai, alpha in enumerate(alphas):
            ee, _ = finish(lams, ns_ref, lens, alpha=float(alpha), ambigs="ignore")
            o = ovf[ai].astype(bool)
            assert np.all(np.isnan(ee[o])) and not np.isnan(ee[~o]).any()
            assert np.array_equal(ee[~o], ee_ref[ai][~o])
            n_cases += len(lens)
            n_ovf += int(o.sum())
    assert n_cases >= 6000 and n_ovf >= 100


@pytest.mark.gpu
def test_poisson_gpu_against_reference_fixture():
    """The GPU lambda reduction (k_lambda, sequential per read) + host tail against the same reference set."""
    from moira_amd.engine import Engine
    alphas, sets = _fixture_sets()
    with Engine(0) as eng:
        for q, lens, ee_ref, ns_ref, ovf in sets:
            for ai, alpha in enumerate(alphas):
                r = eng.filter_poisson(q, lens=lens, alpha=float(alpha), ambigs="ignore")
                o = ovf[ai].astype(bool)
                assert np.array_equal(r.ns, ns_ref)
                assert np.all(np.isnan(r.ee[o])) and np.array_equal(r.ee[~o], ee_ref[ai][~o])
                r2 = eng.filter_poisson(q, lens=lens, alpha=float(alpha), ambigs="treat_as_errors")
                assert np.array_equal(r2.ee[~o], ee_ref[ai][~o] + ns_ref[~o])
                assert np.array_equal(r2.passed[~o], (ee_ref[ai] + ns_ref)[~o] <= lens[~o] * 0.01)


def test_poisson_host_tail_threads_do_not_change_results():
    """Batches of >= 16384 reads are split over threads inside mpb_poisson_finish_host: same values as read by read."""
    rng = np.random.default_rng(9)
    lams = np.concatenate([rng.uniform(0, 6, 30000), rng.uniform(6, 200, 10000)])
    ns = rng.integers(0, 3, len(lams)).astype(np.int32)
    lens = np.full(len(lams), 300, np.int32)
    big, pbig = finish(lams, ns, lens, alpha=0.005)
    for lo in range(0, len(lams), 7919):
        small, psmall = finish(lams[lo:lo + 100], ns[lo:lo + 100], lens[lo:lo + 100], alpha=0.005)
        assert np.array_equal(big[lo:lo + 100], small, equal_nan=True) and np.array_equal(pbig[lo:lo + 100], psmall)


@pytest.mark.gpu
@pytest.mark.parametrize("stride", [16, 48, 128, 144, 320, 608, 1024, 2048])
def test_lambda_kernel_is_the_sequential_sum(stride):
    """k_lambda moves rows through LDS in 128-column panels and lets one lane walk each row: lambda must be the
    reference's left-to-right sum bit for bit (moira/moira.py:1663) for any stride / length / panel boundary,
    Ns the count of byte 0 inside the read, garbage past a read's end must not matter, a byte 255 must be reported."""
    from moira_amd.engine import Engine
    rng = np.random.default_rng(stride)
    n = 1000 if stride <= 1024 else 300
    lens = rng.integers(0, stride + 1, n).astype(np.int32)
    lens[:6] = (0, 1, min(stride, 127), min(stride, 128), min(stride, 129), stride)
    q = rng.integers(1, 60, (n, stride)).astype(np.uint8)
    q[rng.random((n, stride)) < 0.03] = 0
    pad = np.arange(stride)[None, :] >= lens[:, None]
    q[pad] = rng.integers(0, 256, int(pad.sum()), dtype=np.uint8)          # garbage (incl. 0 and 255) in the padding
    p = [0.0] + [10 ** (v / -10.0) for v in range(1, 256)]
    want = np.empty(n)
    want_ns = np.zeros(n, np.int32)
    for i in range(n):
        lam = 0
        for v in q[i, :lens[i]]:
            if v == 0:
                want_ns[i] += 1
            else:
                lam += p[int(v)]
        want[i] = float(lam)
    with Engine(0) as eng:
        def run(mat, d_len_ptr, fixed_len):
            d_q, d_lam, d_ns = eng.alloc(mat.nbytes).upload(mat), eng.alloc(n * 8), eng.alloc(n * 4)
            L.check(eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, d_len_ptr, fixed_len, d_lam.ptr, d_ns.ptr))
            out = d_lam.download(np.float64, n), d_ns.download(np.int32, n)
            for b in (d_q, d_lam, d_ns):
                b.free()
            return out
        d_len = eng.alloc(n * 4).upload(lens)
        lam, ns = run(q, d_len.ptr, 0)                                   # ragged
        assert np.array_equal(lam, want) and np.array_equal(ns, want_ns)
        full = q.copy()
        full[full == 255] = 17                                           # every row read in full: the padding is data now
        lam, ns = run(full, None, stride)                                # fixed length = the whole row
        w2 = np.array([float(sum(p[int(v)] for v in row)) for row in full])     # built-in sum: left to right from int 0
        assert np.array_equal(lam, w2) and np.array_equal(ns, (full == 0).sum(1))
        bad = q.copy()
        bad[3, 0] = 255
        l2 = lens.copy(); l2[3] = max(1, l2[3])
        d_q, d_lam, d_ns = eng.alloc(bad.nbytes).upload(bad), eng.alloc(n * 8), eng.alloc(n * 4)
        d_len.upload(l2)
        assert eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, d_len.ptr, 0, d_lam.ptr, d_ns.ptr) == L.E_INVALID


@pytest.mark.gpu
def test_poisson_host_entry_pipeline_multi_chunk(oracle):
    """mpb_filter_poisson_host runs through the same four-slot pipeline as the Poisson-binomial host entry (H2D |
    k_lambda | D2H overlapped, the scalar tail of a chunk on the host while the GPU works on the next ones): a ragged
    300 k-read batch = four chunks; every result equals the one-chunk path's, a sample equals the reference formula."""
    from moira_amd.engine import Engine
    n = 300_017
    q, lens = oracle.synth_fill(n, 608, min_len=50, max_len=600, seed=5)
    with Engine(0) as eng:
        r = eng.filter_poisson(q, lens=lens, alpha=0.005, ambigs="treat_as_errors")
        assert r.n_pass == int(r.passed.sum()) and 0 < r.n_pass < n
        for lo in range(0, n, 40_000):                      # slices small enough for a single chunk
            hi = min(n, lo + 12_000)
            s = eng.filter_poisson(q[lo:hi], lens=lens[lo:hi], alpha=0.005, ambigs="treat_as_errors")
            assert np.array_equal(r.ee[lo:hi], s.ee, equal_nan=True) and np.array_equal(r.ns[lo:hi], s.ns)
            assert np.array_equal(r.passed[lo:hi], s.passed)
        bad = q.copy()
        bad[n - 5, 3] = 255                                # a lower-case n in the LAST chunk still fails the call
        with pytest.raises(ValueError, match="255"):
            eng.filter_poisson(bad, lens=lens)
    for i in np.random.default_rng(1).integers(0, n, 1500):
        row = q[i, :lens[i]]
        try:
            e, k = calculate_errors_poisson("".join("N" if v == 0 else "A" for v in row), [20 if v == 0 else int(v) for v in row], 0.005)
        except OverflowError:                               # the reference crashes on this read; the library reports NaN
            assert np.isnan(r.ee[i]) and not r.passed[i]
            continue
        assert r.ee[i] == e + k and r.ns[i] == k and bool(r.passed[i]) == (e + k <= lens[i] * 0.01)
