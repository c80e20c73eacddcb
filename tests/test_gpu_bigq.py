"""Quality scores above 254: the byte matrix cannot hold them, the reference takes any int
(moira/bernoullimodule.c:92-108).  The per-read entry gives such a read its own code table for the call; results must be
the real reference's (tests/golden/bigq.json, made by make_golden.py --bigq) bit for bit."""
import io
import os

import numpy as np
import pytest

import golden_io as G

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from moira_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_reference_vectors_with_big_scores(eng):
    from moira_amd.dropin import bernoulli
    reads = G.bigq_fixture()
    for seq, quals, alpha, ee, ns, ub in reads:
        assert eng.calculate_errors_PB(seq, quals, alpha) == (ee, ns), (seq, quals, alpha)
    for seq, quals, alpha, ee, ns, ub in reads[::7]:
        assert bernoulli.calculate_errors_PB(seq, quals, alpha) == (ee, ns)
        assert bernoulli.calculate_errors(seq, quals, alpha) == (ee, ns)
    # the context's own table is back in place afterwards: the reference's known answer, and the uploaded table itself
    kat = G.load_kat()["kat1"]
    assert eng.calculate_errors_PB(kat["seq"], kat["quals"], kat["alpha"]) == (kat["ee"], kat["ns"])
    lut, _ = G.lut_fixture()
    a, b = eng.device_lut()
    assert all(a[q] == lut[q][1] and b[q] == lut[q][2] for q in lut)


def test_reference_vectors_with_big_scores_poisson(eng):
    """The same reads through the reference's calculate_errors_poisson (moira/moira.py:1637-1679): any int is a score,
    Q0 is p = 1 inside the function, 'n' is a base, OverflowError where Lambda ** j leaves the float range."""
    from poisson_ref import calculate_errors_poisson
    reads = G.bigq_poisson_fixture()
    assert sum(1 for r in reads if r[3] is None) >= 1
    for seq, quals, alpha, ee, ns in reads:
        if ee is None:
            with pytest.raises(OverflowError):
                eng.calculate_errors_poisson(seq, quals, alpha)
        else:
            assert eng.calculate_errors_poisson(seq, quals, alpha) == (ee, ns), (seq, quals, alpha)
    # ordinary scores take the same entry (no private table): against the restated formula
    rng = np.random.default_rng(3)
    for _ in range(60):
        L = int(rng.integers(0, 500))
        quals = rng.integers(0, 60, L).tolist()
        seq = "".join(rng.choice(list("ACGTNn"), L, p=[0.24, 0.24, 0.24, 0.24, 0.03, 0.01])) if L else ""
        alpha = float(rng.choice([0.005, 0.05, 0.3]))
        try:
            want = calculate_errors_poisson(seq, quals, alpha)
        except OverflowError:
            with pytest.raises(OverflowError):
                eng.calculate_errors_poisson(seq, quals, alpha)
            continue
        assert eng.calculate_errors_poisson(seq, quals, alpha) == want, (seq, quals, alpha)
    with pytest.raises(ValueError):
        eng.calculate_errors_poisson("ACGT", [30, -1, 30, 30], 0.005)
    with pytest.raises(ValueError):
        eng.calculate_errors_poisson("ACGT", [30, 30, 30], 0.005)
    # afterwards the batch entry still sums with the context's own table
    q, lens = eng.pack(["ACGT" * 5], [[20] * 20], 32)
    assert eng.filter_poisson(q, lens=lens, alpha=0.005, ambigs="ignore", uncert=1.0).ee[0] == calculate_errors_poisson("ACGT" * 5, [20] * 20, 0.005)[0]


def test_big_scores_on_every_path_against_the_oracle(eng, oracle):
    """Random reads: short (one read per wave), long rows, and more than 1024 DP rows (k_wide) -- each with a few scores
    between 255 and 2^31 - 1, with and without ambiguous bases."""
    rng = np.random.default_rng(254)
    big = [255, 299, 1000, 3239, 3240, 70000, 2 ** 31 - 1]
    cases = [(int(rng.integers(1, 400)), 2, 42) for _ in range(150)] + [(int(rng.integers(1025, 3000)), 20, 42) for _ in range(12)] + \
            [(1400, 1, 3), (2100, 1, 4), (1600, 1, 2)]
    most = 0
    for L, lo, hi in cases:
        quals = rng.integers(lo, hi, L).tolist()
        for pos in rng.integers(0, L, 1 + L // 50):
            quals[int(pos)] = int(rng.choice(big))
        seq = np.full(L, ord("A"), np.uint8)
        if rng.random() < 0.4:
            seq[rng.integers(0, L, 1 + L // 60)] = ord("N")
            seq[rng.integers(0, L, 1)] = ord("n")
        seq = seq.tobytes().decode()
        alpha = float(rng.choice([0.005, 0.05, 1e-4]))
        e, ns, rows = oracle.ee_rowwise(seq, quals, alpha)
        assert eng.calculate_errors_PB(seq, quals, alpha) == (e, ns), (L, lo, hi, alpha, rows)
        most = max(most, rows)
    assert most > 1024                                       # some case needs more rows than one wave holds


def test_too_many_distinct_scores_is_an_error_and_ints_wrap_like_the_c_cast(eng, oracle):
    with pytest.raises(ValueError, match="distinct quality values"):
        eng.calculate_errors_PB("A" * 600, list(range(1, 200)) + [1000 + i for i in range(401)], 0.005)
    # (int)PyInt_AsLong (bernoullimodule.c:97): 2^32 + 30 is 30 after the cast
    assert eng.calculate_errors_PB("ACGT", [2 ** 32 + 30] * 4, 0.005) == eng.calculate_errors_PB("ACGT", [30] * 4, 0.005)
    with pytest.raises(ValueError):                          # 2^31 wraps to a negative score
        eng.calculate_errors_PB("ACGT", [2 ** 31] * 4, 0.005)
    with pytest.raises(ValueError):                          # the batch entries keep their byte limit
        eng.pack(["ACGT"], [[30, 255, 30, 30]], 16)


def test_cli_scores_a_qual_file_with_big_scores(tmp_path, oracle):
    from moira_amd import cli
    from test_cli_golden import reference_args
    fa, qu = tmp_path / "r.fasta", tmp_path / "r.qual"
    rng = np.random.default_rng(11)
    recs = []
    with open(fa, "w") as f, open(qu, "w") as g:
        for k in range(60):
            L = int(rng.integers(30, 200))
            quals = rng.integers(20, 42, L).tolist()
            if k % 5 == 1:
                quals[int(rng.integers(0, L))] = int(rng.choice([255, 300, 4000]))
            seq = "".join(rng.choice(list("ACGT"), L))
            recs.append((seq, quals))
            f.write(">r%d\n%s\n" % (k, seq))
            g.write(">r%d\n%s\n" % (k, " ".join(map(str, quals))))
    out = str(tmp_path / "o")
    a = reference_args(paired=False, forward_fasta=str(fa), forward_qual=str(qu), output_prefix=out, collapse=False, uncert=0.02)
    assert cli.main(a, out=open(os.devnull, "w")) == 0
    good = {l[1:].split("\t")[0].strip() for l in open(out + ".qc.good.fasta") if l.startswith(">")}
    bad = {l[1:].split("\t")[0].strip() for l in open(out + ".qc.bad.fasta") if l.startswith(">")}
    assert len(good) + len(bad) == 60 and good and bad
    for k, (seq, quals) in enumerate(recs):
        e, ns, _ = oracle.ee_rowwise(seq, quals, 0.005)
        assert (("r%d" % k) in good) == (not (e > 0.02 * len(seq))), k        # moira.py:887 `expected_errors > maxerrors`
    text = open(out + ".qc.good.qual").read() + open(out + ".qc.bad.qual").read()
    assert " 300 " in text or " 255 " in text or " 4000 " in text
    # --error_calc poisson on the same files: decisions of the reference's formula
    from poisson_ref import calculate_errors_poisson
    a = reference_args(paired=False, forward_fasta=str(fa), forward_qual=str(qu), output_prefix=out + "p", collapse=False,
                       error_calc="poisson", uncert=0.02)
    assert cli.main(a, out=open(os.devnull, "w")) == 0
    good = {l[1:].split("\t")[0].strip() for l in open(out + "p.qc.good.fasta") if l.startswith(">")}
    for k, (seq, quals) in enumerate(recs):
        e, ns = calculate_errors_poisson(seq, quals, 0.005)
        assert (("r%d" % k) in good) == (not (e > 0.02 * len(seq))), k


def test_batches_with_big_scores_through_the_coded_entry(eng, oracle):
    """Round 4 (missing 3): the BATCH entries take scores above 254 too -- mpb_pack_batch_coded gives every
    distinct out-of-range score of the batch a spare byte code, mpb_filter_host_coded runs the pipeline on a private copy
    of the table.  The reference's own results for the big-score reads (tests/golden/bigq.json), in batches; then a large
    random batch (tile pipeline, all modes) against the per-read oracle; then the case that cannot fit."""
    reads = [r for r in G.bigq_fixture()]
    by_alpha = {}
    for seq, quals, alpha, ee, ns, ub in reads:
        by_alpha.setdefault(alpha, []).append((seq, quals, ee, ns))
    checked = 0
    for alpha, group in by_alpha.items():
        batch = []
        for item in group + [None]:
            trial = batch + [item] if item is not None else batch
            ok = item is not None
            if ok:
                try:
                    eng.pack_coded([t[0] for t in trial], [t[1] for t in trial])
                except ValueError:
                    ok = False                                   # no free code left: close the batch before this read
            if ok:
                batch = trial
                continue
            if batch:
                q, lens, codes = eng.pack_coded([t[0] for t in batch], [t[1] for t in batch])
                assert (codes[1:255] != np.arange(1, 255)).sum() >= 1
                r = eng.filter(q, lens=lens, alpha=alpha, ambigs="ignore", uncert=1.0, code_scores=codes)
                assert [(float(e), int(n)) for e, n in zip(r.ee, r.ns)] == [(t[2], t[3]) for t in batch]
                checked += len(batch)
            batch = [item] if item is not None else []
            if item is not None:
                try:
                    eng.pack_coded([item[0]], [item[1]])
                except ValueError:
                    batch = []                                   # a single read with more big values than free codes
    assert checked >= 80
    # a large batch: 6,000 random reads of which a tenth carry big scores -> the sorted, tiled pipeline on the private table
    rng = np.random.default_rng(77)
    big = [255, 256, 300, 999, 3239, 3240, 3241, 70000, 2 ** 31 - 1]
    seqs, quals = [], []
    for k in range(6000):
        n = int(rng.integers(1, 330))
        ql = rng.integers(2, 42, n).tolist()
        sq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        amb = rng.random(n) < 0.01
        sq[amb] = np.where(rng.random(int(amb.sum())) < 0.7, ord("N"), ord("n"))
        if k % 10 == 0:
            for pos in rng.integers(0, n, 1 + n // 50):
                ql[int(pos)] = int(rng.choice(big))
        if k % 97 == 0:
            ql[0] = 0
        seqs.append(sq.tobytes().decode())
        quals.append(ql)
    q, lens, codes = eng.pack_coded(seqs, quals)
    assert sorted(int(v) for v in codes[codes != np.arange(256)]) == sorted(big)
    for kw in ({}, {"ambigs": "disallow", "round_": True, "maxerrors": 2.0}, {"alpha": 0.05, "ambigs": "ignore"}):
        r = eng.filter(q, lens=lens, code_scores=codes, batched_only=True, **kw)
        alpha = kw.get("alpha", 0.005)
        for i in range(0, 6000, 7):
            e, ns, _ = oracle.ee_rowwise(seqs[i], quals[i], alpha)
            if kw.get("ambigs", "treat_as_errors") == "treat_as_errors":
                e = e + ns
            if kw.get("round_"):
                e = float(np.floor(e))
            assert (float(r.ee[i]), int(r.ns[i])) == (e, ns), (i, kw)
    # the context's own table is back afterwards
    kat = G.load_kat()["kat1"]
    assert eng.calculate_errors_PB(kat["seq"], kat["quals"], kat["alpha"]) == (kat["ee"], kat["ns"])
    # more distinct big scores than free codes: refused at pack time, by name
    with pytest.raises(ValueError, match="distinct scores above 254"):
        eng.pack_coded(["A" * 600], [list(range(1, 255)) + list(range(300, 646))])
    with pytest.raises(ValueError, match="positive"):
        eng.pack_coded(["ACGT"], [[3, -1, 3, 3]])
