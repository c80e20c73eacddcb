"""Pins the CPU oracle (oracle/pb_oracle.c) against the reference.

Sources of truth, strongest first:
  1. the reference's own known-answer tests (moira/test/test_moira.py:40,43,127,128) -> kat.json
  2. the reference's golden output files (moira/test/test_results/) -> reference_test_results/
  3. outputs of the real reference extension run in the build container (make_golden.py) -> *.npz
  4. the real reference extension itself when oracle/_ref/bernoulli.so is present (live check)
"""
import math
import os

import numpy as np
import pytest

import golden_io as G


def test_kat1_all_shapes(oracle):
    k = G.load_kat()["kat1"]
    for fn in (oracle.ee_refshape, oracle.ee_rowwise):
        ee, ns, rows = fn(k["seq"], k["quals"], k["alpha"])
        assert (ee, ns) == (k["ee"], k["ns"]) == (6.446879136706666, 0)
    assert oracle.ee_python(k["seq"], k["quals"], k["alpha"]) == (6.446879136706666, 0)


def test_kat2_kat3_process_data_values(oracle):
    kat = G.load_kat()
    for name, want in (("kat2_forward_truncate200", 0.9685179556745876),
                       ("kat3_paired_truncate200", 0.9643903629780557)):
        k = kat[name]
        assert k["ee_plus_ns"] == want
        ee, ns, _ = oracle.ee_rowwise(k["seq"], k["quals"], k["alpha"])
        assert ee + ns == want          # process_data adds Ns (moira/moira.py:827-828)
        ee2, ns2, _ = oracle.ee_refshape(k["seq"], k["quals"], k["alpha"])
        assert (ee2, ns2) == (ee, ns)


@pytest.mark.parametrize("name", G.NPZ_SETS)
def test_oracle_matches_reference_vectors(oracle, name):
    s = G.load_set(name)
    alpha = float(s["alpha"])
    ee, ns, ps, rows = oracle.filter_batch(s["q"], lens=s["lens"], alpha=alpha,
                                           ambigs="ignore", threads=4)
    exp = G.expected_value(s)
    assert not np.isnan(exp).any()
    assert np.array_equal(ee, exp)                       # bit-exact
    assert np.array_equal(ns, s["ns_ref"])
    ub = s["ub"].astype(bool)
    assert np.all(exp[ub] == 0.0)                        # Python-twin semantics of the UB case
    chk = ~np.isnan(s["ee_py"])
    assert np.array_equal(ee[chk], s["ee_py"][chk])      # Python twin agrees wherever it was run


@pytest.mark.parametrize("name", ["edge_alpha_0.005", "synth250", "rand_alpha05"])
def test_refshape_equals_rowwise(oracle, name):
    s = G.load_set(name)
    n = min(len(s["lens"]), 300)
    a = oracle.filter_batch(s["q"][:n], lens=s["lens"][:n], alpha=float(s["alpha"]), shape=0)
    b = oracle.filter_batch(s["q"][:n], lens=s["lens"][:n], alpha=float(s["alpha"]), shape=1)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def test_python_restatement_small(oracle):
    s = G.load_set("edge_alpha_0.005")
    exp = G.expected_value(s)
    for i in range(len(exp)):
        L = int(s["lens"][i])
        if L > 60:
            continue
        row = s["q"][i, :L]
        seq = "".join("N" if v == 0 else "n" if v == 255 else "A" for v in row)
        quals = [20 if v in (0, 255) else int(v) for v in row]
        ee, ns = oracle.ee_python(seq, quals, 0.005)
        assert ee == exp[i] and ns == s["ns_ref"][i]


@pytest.mark.parametrize("which,n_good,n_bad", [("forward", 122, 365), ("paired", 324, 76)])
def test_reference_golden_files_decisions(oracle, golden_dir, which, n_good, n_bad):
    """Every representative in the reference's *.qc.good files must pass and every one in
    *.qc.bad must fail (all labelled 'uncert > 0.010'), with the reference's test arguments
    (alpha 0.005, uncert 0.01, ambigs treat_as_errors; moira/test/test_moira.py:130-135)."""
    base = os.path.join(golden_dir, "reference_test_results", which + ".qc.")
    for kind, want, count in (("good", 1, n_good), ("bad", 0, n_bad)):
        recs = G.read_fasta_qual(base + kind)
        assert len(recs) == count
        stride = 16 * ((max(len(r[2]) for r in recs) + 15) // 16)
        q = np.stack([oracle.pack_read(r[2], r[3], stride) for r in recs])
        lens = np.array([len(r[2]) for r in recs], np.int32)
        ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, alpha=0.005, uncert=0.01,
                                            ambigs="treat_as_errors", threads=4)
        assert np.all(ps == want), (which, kind, int((ps != want).sum()))
        if kind == "bad":
            assert all(r[1] == "uncert > 0.010" for r in recs)


def test_lut_pins(oracle):
    a, b = oracle.lut()
    # p' == p bit-for-bit for every encodable score, and a == 1 - p
    for q in range(1, 255):
        p = math.pow(10, q / -10.0)
        assert b[q] == p and a[q] == 1 - p
    assert (a[0], b[0], a[255], b[255]) == (1.0, 0.0, 1.0, 0.0)


def test_lut_matches_the_committed_fixture(oracle):
    """the LUT values are fixtures (hex floats from the build container's libm).  A libm that rounds
    pow differently on another box fails HERE, by name, instead of as unexplained vector mismatches."""
    lut, _ = G.lut_fixture()
    a, b = oracle.lut()
    assert sorted(lut) == list(range(1, 255))
    for q, (p, am, bm) in lut.items():
        assert math.pow(10, q / -10.0) == p, q          # this box's libm
        assert a[q] == am and b[q] == bm, q             # the oracle's table


def test_product_lut_matches_the_committed_fixture():
    """The table libmoira_pb.so uploads (mpb_host_lut: the host-side construction, no device needed)."""
    from moira_amd.engine import host_lut
    lut, _ = G.lut_fixture()
    a, b = host_lut()
    for q, (p, am, bm) in lut.items():
        assert a[q] == am and b[q] == bm and b[q] == p, q
    assert (a[0], b[0], a[255], b[255]) == (1.0, 0.0, 1.0, 0.0)


def test_libm_probes_of_the_poisson_tail():
    """exp(-lam) and pow(lam, j) as moira/moira.py:1671 calls them: the host tail of --error_calc poisson uses this
    box's libm, so its values are pinned too."""
    _, probes = G.lut_fixture()
    assert len(probes) >= 50
    for lam, j, e, pw in probes:
        assert math.exp(-lam) == e, lam
        try:
            got = math.pow(lam, j)
        except OverflowError:
            got = math.inf
        assert got == pw, (lam, j)


def test_synthetic_generator_is_pinned(oracle):
    for name, kw, stride in (("synth300", dict(fixed_len=300, seed=2), 320),
                             ("synth250", dict(fixed_len=250, seed=1), 256),
                             ("synth_ragged", dict(min_len=50, max_len=600, seed=5), 608)):
        s = G.load_set(name)
        q, lens = oracle.synth_fill(len(s["lens"]), stride, **kw)
        assert np.array_equal(q, s["q"]) and np.array_equal(lens, s["lens"])
    # first_read offsets address the same stream
    q0, _ = oracle.synth_fill(100, 320, fixed_len=300, seed=2, first_read=50)
    assert np.array_equal(q0, G.load_set("synth300")["q"][50:150])


def test_predicate_modes(oracle):
    s = G.load_set("rand_mixed")
    q, lens = s["q"][:500], s["lens"][:500]
    raw, ns, _, _ = oracle.filter_batch(q, lens=lens, ambigs="ignore")
    ee_t, _, p_t, _ = oracle.filter_batch(q, lens=lens, ambigs="treat_as_errors")
    assert np.array_equal(ee_t, raw + ns)
    assert np.array_equal(p_t, (ee_t <= lens * 0.01).astype(np.uint8))
    ee_r, _, p_r, _ = oracle.filter_batch(q, lens=lens, ambigs="treat_as_errors", round_=True)
    assert np.array_equal(ee_r, np.floor(raw + ns))
    ee_m, _, p_m, _ = oracle.filter_batch(q, lens=lens, ambigs="ignore", maxerrors=2.5)
    assert np.array_equal(p_m, (raw <= 2.5).astype(np.uint8))
    _, _, p_d, _ = oracle.filter_batch(q, lens=lens, ambigs="disallow")
    has_upper = np.array([(q[i, :lens[i]] == 0).any() for i in range(len(lens))])
    assert np.array_equal(p_d, ((raw <= lens * 0.01) & ~has_upper).astype(np.uint8))


def test_live_reference_when_built(oracle):
    ref = oracle.reference_module()
    if ref is None:
        pytest.skip("oracle/_ref/bernoulli.so not built (needs /root/reference)")
    rng = np.random.default_rng(7)
    for _ in range(200):
        L = int(rng.integers(2, 320))
        quals = [int(x) for x in rng.integers(1, 42, L)]
        seq = "".join(rng.choice(list("ACGTN"), L, p=[.245, .245, .245, .245, .02]))
        ee, ns, rows = oracle.ee_rowwise(seq, quals, 0.005)
        if rows > 1:                       # rows == 1 is the reference's UB case
            assert ref.calculate_errors_PB(seq, quals, 0.005) == (ee, ns)


def test_oracle_takes_qualities_above_254_as_the_reference_does(oracle):
    """The per-read entry points of the oracle work on the reference's own inputs (sequence + ints), so a quality score
    the byte matrix cannot hold is just another int: pinned by the real extension's results (make_golden.py --bigq)."""
    reads = G.bigq_fixture()
    assert len(reads) >= 90 and sum(1 for r in reads if max(r[1]) > 254) == len(reads)
    for seq, quals, alpha, ee, ns, ub in reads:
        for fn in (oracle.ee_rowwise, oracle.ee_refshape):
            e, s, rows = fn(seq, quals, alpha)
            assert s == ns and (rows == 1) == bool(ub)
            if not ub or fn is oracle.ee_rowwise:           # the reference-shaped loop nest shares the C reference's UB case
                assert e == ee, (seq, quals, alpha)


def test_poisson_restatement_on_big_scores():
    """oracle/poisson_ref.py against the reference function's results on the big-score reads (Q0 = p 1, 'n' a base,
    OverflowError where the reference raises it)."""
    from poisson_ref import calculate_errors_poisson
    for seq, quals, alpha, ee, ns in G.bigq_poisson_fixture():
        if ee is None:
            with pytest.raises(OverflowError):
                calculate_errors_poisson(seq, quals, alpha)
        else:
            assert calculate_errors_poisson(seq, quals, alpha) == (ee, ns)
