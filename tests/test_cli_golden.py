"""End-to-end: the moira-compatible CLI must reproduce the reference's golden output files
byte for byte (moira/test/test_moira.py:73-113: forward dataset, paired dataset, compression).

The CPU variants inject the oracle as the per-chunk filter so the host logic (parsers, contig
construction, collapse rule, Python-2 dict order, writers) is tested without a GPU; the -m gpu
variants run the product path (HIP library) end to end.
"""
import bz2
import gzip
import os
import types

import numpy as np
import pytest

from moira_amd import cli

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
RES = os.path.join(GOLD, "reference_test_results")


def reference_args(**kw):
    """The argument namespace of the reference's tests (moira/test/test_moira.py:130-135)."""
    d = dict(alpha=0.005, match=1, gap=-2, mismatch=-1, insert=20, deltaq=6, consensus_qscore="best",
             paired=True, truncate=None, only_contig=False, error_calc="poisson_binomial",
             ambigs="treat_as_errors", round=False, silent=True, nowarnings=False, doc=False, uncert=0.01,
             maxerrors=None, processors=4, forward_fasta=None, forward_qual=None, reverse_fasta=None,
             reverse_qual=None, forward_fastq=None, reverse_fastq=None, output_format="fasta", collapse=True,
             pipeline="mothur", fastq_offset=33, relabel=None, output_compression="none", qscore_cap=40,
             min_overlap=None, trim_overlap=False, bootstrap=100, output_prefix=None, device=None,
             fast_discard=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def oracle_backend(oracle):
    def backend(seqs, quals, alpha, ambigs, round_):
        stride = 16 * ((max(len(s) for s in seqs) + 15) // 16)
        quals = [ql.ints() if hasattr(ql, "ints") else ql for ql in quals]
        q = np.stack([oracle.pack_read(s, ql, stride) for s, ql in zip(seqs, quals)])
        lens = np.array([len(s) for s in seqs], np.int32)
        ee, _, _, _ = oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)
        return ee

    def matrix(q, lens, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        # packed reads (the byte-level FASTQ path, moira_amd/fastio.py)
        return oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)[0]
    backend.matrix = matrix
    backend.methods = ("poisson_binomial",)
    return backend


def same_files(out_prefix, gold_prefix, kinds=("good.fasta", "good.qual", "good.names", "bad.fasta", "bad.qual", "bad.names")):
    for k in kinds:
        got = open("%s.qc.%s" % (out_prefix, k)).read()
        want = open(os.path.join(RES, "%s.qc.%s" % (gold_prefix, k))).read()
        assert got == want, k


def run_forward(tmp_path, backend):
    out = str(tmp_path / "forward")
    a = reference_args(paired=False, forward_fastq=os.path.join(GOLD, "test1.fastq.gz"), output_prefix=out)
    assert cli.main(a, backend=backend, out=open(os.devnull, "w")) == 0
    same_files(out, "forward")


def run_paired(tmp_path, backend, compression="none"):
    out = str(tmp_path / "paired")
    a = reference_args(paired=True, forward_fastq=os.path.join(GOLD, "test1.fastq.gz"),
                       reverse_fastq=os.path.join(GOLD, "test2.fastq.bz2"), output_prefix=out,
                       output_compression=compression)
    assert cli.main(a, backend=backend, out=open(os.devnull, "w")) == 0
    if compression == "none":
        same_files(out, "paired")
        rep = open(out + ".contigs.report").read().split("\n")
        assert rep[0] == "header\tn_seqs\toverlap_length\tgaps\tmismatches" and len(rep) == 402
    else:
        opener = gzip.open if compression == "gz" else bz2.open
        got = opener("%s.qc.good.fasta.%s" % (out, compression), "rt").read()
        assert got == open(os.path.join(RES, "paired.qc.good.fasta")).read()


def test_forward_dataset_host_logic(tmp_path, oracle):
    run_forward(tmp_path, oracle_backend(oracle))


def test_paired_dataset_host_logic_line_parser(tmp_path, oracle, monkeypatch):
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    run_paired(tmp_path, oracle_backend(oracle))


def test_forward_dataset_host_logic_line_parser(tmp_path, oracle, monkeypatch):
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")          # the per-line Python path, kept as the fallback
    run_forward(tmp_path, oracle_backend(oracle))


def test_paired_dataset_host_logic(tmp_path, oracle):
    run_paired(tmp_path, oracle_backend(oracle))


@pytest.mark.parametrize("comp", ["gz", "bz2"])
def test_compressed_output_host_logic(tmp_path, oracle, comp):
    run_paired(tmp_path, oracle_backend(oracle), comp)


def test_argument_validation_messages(capsys):
    a = reference_args(paired=False, forward_fastq=None)
    assert cli.main(a, backend=lambda *x: None) == 1
    assert "You must at least provide one fastq file" in capsys.readouterr().out
    a = reference_args(paired=False, forward_fastq="x.fastq", alpha=1.5, uncert=0)
    assert cli.main(a, backend=lambda *x: None) == 1
    out = capsys.readouterr().out
    assert "The alpha parameter must be between 0 (not included) and 1." in out
    assert "The uncert parameter must be between 0 (not included) and 1." in out
    ns = cli.parse_arguments(["-ffq", "a.fastq", "-c", "false", "-me", "3", "-n", "disallow", "-t", "200"])
    assert ns.collapse is False and ns.maxerrors == 3 and ns.ambigs == "disallow" and ns.truncate == 200
    assert cli.parse_arguments(["-ffq", "a.fastq"]).collapse is True


def test_modes_without_collapse_and_fastq_output(tmp_path, oracle):
    out = str(tmp_path / "nc")
    a = reference_args(paired=False, forward_fastq=os.path.join(GOLD, "test1.fastq.gz"), output_prefix=out,
                       collapse=False, output_format="fastq", pipeline="USEARCH", truncate=200, maxerrors=1.0,
                       uncert=0.01)
    assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
    good = open(out + ".qc.good.fastq").read().split("\n")
    bad = open(out + ".qc.bad.fastq").read().split("\n")
    assert (len(good) - 1) % 4 == 0 and (len(bad) - 1) % 4 == 0
    assert (len(good) - 1) // 4 + (len(bad) - 1) // 4 == 1000
    assert all(len(good[i + 1]) == 200 for i in range(0, len(good) - 1, 4))
    assert ";ee=" in good[0] and good[0].endswith(";size=1;")
    assert all("errors > 1.00" in bad[i] for i in range(0, len(bad) - 1, 4))


@pytest.mark.gpu
def test_forward_dataset_gpu(tmp_path):
    run_forward(tmp_path, None)


@pytest.mark.gpu
def test_paired_dataset_gpu(tmp_path):
    run_paired(tmp_path, None)


@pytest.mark.gpu
def test_cli_poisson_method_gpu(tmp_path):
    """--error_calc poisson through the GPU lambda reduction equals the reference's formula
    (oracle/poisson_ref.py) evaluated per read."""
    import math
    from poisson_ref import calculate_errors_poisson

    def formula_backend(seqs, quals, alpha, ambigs, round_, method="poisson"):
        assert method == "poisson"
        ee = []
        for s, ql in zip(seqs, quals):
            e, ns = calculate_errors_poisson(s, ql.ints() if hasattr(ql, "ints") else ql, alpha)
            e = e + ns if ambigs == "treat_as_errors" else e
            ee.append(math.floor(e) if round_ else e)
        return ee
    formula_backend.methods = ("poisson",)
    outs = []
    for name, be in (("gpu", None), ("host", formula_backend)):
        out = str(tmp_path / name)
        a = reference_args(paired=False, forward_fastq=os.path.join(GOLD, "test1.fastq.gz"), output_prefix=out,
                           error_calc="poisson", collapse=False)
        assert cli.main(a, backend=be, out=open(os.devnull, "w")) == 0
        outs.append([open("%s.qc.%s" % (out, k)).read() for k in ("good.fasta", "bad.fasta", "good.qual")])
    assert outs[0] == outs[1]


@pytest.mark.gpu
def test_cli_fast_discard_keeps_the_same_reads(tmp_path):
    outs = []
    for fd in (False, True):
        out = str(tmp_path / ("fd%d" % fd))
        a = reference_args(paired=False, forward_fastq=os.path.join(GOLD, "test1.fastq.gz"), output_prefix=out,
                           collapse=False, fast_discard=fd)
        assert cli.main(a, out=open(os.devnull, "w")) == 0
        outs.append([open("%s.qc.%s" % (out, k)).read() for k in ("good.fasta", "good.qual", "bad.fasta", "bad.qual")])
    assert outs[0] == outs[1]


def _fastq_to_fasta_qual(src_opener, src, fa, qu):
    with src_opener(src, "rt") as f, open(fa, "w") as ffa, open(qu, "w") as fqu:
        lines = [l.rstrip("\n") for l in f]
    with open(fa, "w") as ffa, open(qu, "w") as fqu:
        for i in range(0, len(lines), 4):
            h = lines[i][1:]
            ffa.write(">%s some description\n%s\n" % (h, lines[i + 1]))
            fqu.write(">%s\tother text\n%s\n" % (h, " ".join(str(ord(c) - 33) for c in lines[i + 3])))


def test_fasta_qual_input_reproduces_the_same_golden_files(tmp_path, oracle):
    """The fasta+qual reader (moira/moira.py:1093-1149): same reads as test1/test2.fastq, written as
    fasta + qual with trailing header text (which the reader must drop), give the same outputs."""
    f1, q1 = str(tmp_path / "r1.fasta"), str(tmp_path / "r1.qual")
    f2, q2 = str(tmp_path / "r2.fasta"), str(tmp_path / "r2.qual")
    _fastq_to_fasta_qual(gzip.open, os.path.join(GOLD, "test1.fastq.gz"), f1, q1)
    _fastq_to_fasta_qual(bz2.open, os.path.join(GOLD, "test2.fastq.bz2"), f2, q2)
    out = str(tmp_path / "fwd")
    a = reference_args(paired=False, forward_fasta=f1, forward_qual=q1, output_prefix=out)
    assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
    same_files(out, "forward")
    out = str(tmp_path / "prd")
    a = reference_args(paired=True, forward_fasta=f1, forward_qual=q1, reverse_fasta=f2, reverse_qual=q2,
                       output_prefix=out)
    assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
    same_files(out, "paired")
    # default output name = input name without its extension (moira.py:300-303)
    a = reference_args(paired=False, forward_fasta=f1, forward_qual=q1)
    assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
    assert os.path.exists(str(tmp_path / "r1.qc.good.fasta"))


def test_reader_errors(tmp_path):
    fa, qu = str(tmp_path / "a.fasta"), str(tmp_path / "a.qual")
    open(fa, "w").write(">r1\nACGT\n")
    open(qu, "w").write(">r2\n30 30 30 30\n")
    with pytest.raises(cli.NameMismatchError):
        list(cli.parse_fasta_and_qual(cli.open_input(fa), cli.open_input(qu)))
    open(qu, "w").write(">r1\n30 30 30\n")
    with pytest.raises(cli.LengthMismatchError):
        list(cli.parse_fasta_and_qual(cli.open_input(fa), cli.open_input(qu)))
    fq = str(tmp_path / "a.fastq")
    open(fq, "w").write("@r:1 extra\nACGT\n+\nIIII\n@r2\n\n+\n\n")
    it = cli.parse_fastq(cli.open_input(fq))
    assert next(it)[:3] == ("r_1", "ACGT", [40, 40, 40, 40])
    with pytest.raises(cli.EmptySeqError):
        next(it)
    fq2 = str(tmp_path / "b.fastq")
    open(fq2, "w").write("@other\nACGT\n+\nIIII\n")
    open(fq, "w").write("@r1\nACGT\n+\nIIII\n")
    with pytest.raises(cli.NameMismatchError):
        list(cli.parse_fastq(cli.open_input(fq), cli.open_input(fq2)))


def _config1_files(tmp_path):
    """BASELINE configs[0]: the 1 000 synthetic single-end 250 bp reads (seed 1) as fasta + qual, with the
    reference's own per-read results on exactly these reads (tests/golden/synth250.npz, produced by the
    real bernoullimodule.c)."""
    import golden_io as G
    s = G.load_set("synth250")
    q, lens = s["q"], s["lens"]
    rng = np.random.default_rng(1)
    fa, qu = [], []
    for i in range(len(lens)):
        L = int(lens[i])
        row = q[i, :L]
        seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, L)].copy()
        seq[row == 0] = ord("N")
        quals = np.where(row == 0, 2, row)                     # an N's quality is never looked at
        fa.append(">s%d\n%s\n" % (i, seq.tobytes().decode()))
        qu.append(">s%d\n%s\n" % (i, " ".join(map(str, quals))))
    (tmp_path / "c1.fasta").write_text("".join(fa))
    (tmp_path / "c1.qual").write_text("".join(qu))
    ee = G.expected_value(s) + s["ns_ref"]                     # --ambigs treat_as_errors (moira.py:827-828)
    return str(tmp_path / "c1.fasta"), str(tmp_path / "c1.qual"), ee, lens, float(s["alpha"])


def _run_config1(tmp_path, backend):
    fasta, qual, ee, lens, alpha = _config1_files(tmp_path)
    out = str(tmp_path / "c1out")
    a = reference_args(paired=False, forward_fasta=fasta, forward_qual=qual, output_prefix=out, collapse=False,
                       pipeline="USEARCH", alpha=alpha)
    assert cli.main(a, backend=backend, out=open(os.devnull, "w")) == 0
    want_good = {"s%d" % i for i in range(len(ee)) if ee[i] <= lens[i] * 0.01}
    got = {}
    for kind in ("good", "bad"):
        for line in open("%s.qc.%s.fasta" % (out, kind)):
            if line.startswith(">"):
                name, e = line[1:].split("\t")[0].split(";ee=")
                got[name] = (kind, e.split(";")[0])
    assert len(got) == len(ee)
    assert {k for k, v in got.items() if v[0] == "good"} == want_good
    for i in range(len(ee)):
        assert got["s%d" % i][1] == "%.2f" % ee[i]              # the reference's expected errors, to the printed digits


def test_config1_plumbing_host_logic(tmp_path, oracle):
    _run_config1(tmp_path, oracle_backend(oracle))


@pytest.mark.gpu
def test_config1_plumbing_gpu(tmp_path):
    _run_config1(tmp_path, None)


@pytest.mark.parametrize("line_parser", [False, True])
def test_over_long_read_fails_cleanly_before_filtering(tmp_path, oracle, monkeypatch, line_parser):
    """a read longer than the Poisson-binomial kernels cover (65535 bases since round 4; 16383 in round 3; 1023 before) used to
    abort the run from inside the library after earlier chunks had been written.  It now stops before the chunk is
    filtered, says which read and what to do, returns 1 and leaves no partial output files; --error_calc poisson has
    no such limit."""
    import io
    if line_parser:
        monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    fq = tmp_path / "long.fastq"
    with open(fq, "w") as f:
        for k, n in enumerate((300, 66000, 250)):
            f.write("@r%d\n%s\n+\n%s\n" % (k, "ACGT" * (n // 4) + "A" * (n % 4), "I" * n))
    out = str(tmp_path / "o")
    msg = io.StringIO()
    a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=out, collapse=False, silent=True)
    assert cli.main(a, backend=oracle_backend(oracle), out=msg, _no_fastio=line_parser) == 1
    text = msg.getvalue()
    assert "r1" in text and "66000 bases" in text and "--error_calc poisson" in text and "--truncate" in text
    assert not [p for p in os.listdir(tmp_path) if p.startswith("o.")]
    # with --truncate the same file goes through
    a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=out, collapse=False, truncate=200)
    assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w"), _no_fastio=line_parser) == 0
    assert os.path.exists(out + ".qc.good.fasta")


def test_quality_above_254_fails_cleanly(tmp_path, oracle):
    """a .qual file may hold any integer (the reference takes them, moira/bernoullimodule.c:92-108); the
    packed matrix holds one byte per base.  Such a run stops before the chunk is filtered, names the read and the score,
    returns 1 and leaves no partial output -- it is never scored with a clamped value."""
    import io
    fa, qu = tmp_path / "r.fasta", tmp_path / "r.qual"
    with open(fa, "w") as f, open(qu, "w") as g:
        for k, top in enumerate((40, 300, 41)):
            f.write(">r%d\n%s\n" % (k, "ACGT" * 10))
            g.write(">r%d\n%s\n" % (k, " ".join(str(top if i == 7 else 30) for i in range(40))))
    out = str(tmp_path / "o")
    msg = io.StringIO()

    def backend(seqs, quals, alpha, ambigs, round_, **kw):
        return oracle_backend(oracle)(seqs, quals, alpha, ambigs, round_)
    backend.methods = ("poisson_binomial", "poisson")
    a = reference_args(paired=False, forward_fasta=str(fa), forward_qual=str(qu), output_prefix=out, collapse=False, silent=True)
    assert cli.main(a, backend=backend, out=msg) == 1
    text = msg.getvalue()
    assert "r1" in text and "300" in text and "254" in text
    assert not [p for p in os.listdir(tmp_path) if p.startswith("o.")]


@pytest.mark.gpu
@pytest.mark.parametrize("line_parser", [False, True])
def test_long_reads_run_with_the_poisson_method_gpu(tmp_path, line_parser):
    """moira's README recommends --error_calc poisson for reads > 500 nt; the Poisson path has no 1023-base limit
    (only the Poisson-binomial kernels do), so a file with 1,100- and 2,500-base reads runs to completion with it."""
    from poisson_ref import calculate_errors_poisson
    fq = tmp_path / "long.fastq"
    recs = []
    rng = np.random.default_rng(4)
    for k, n in enumerate((300, 1100, 2500, 700)):
        q = "".join(chr(33 + int(v)) for v in rng.integers(25, 41, n))
        s = "".join(rng.choice(list("ACGT"), n))
        recs.append((s, q))
    with open(fq, "w") as f:
        for k, (s, q) in enumerate(recs):
            f.write("@r%d\n%s\n+\n%s\n" % (k, s, q))
    out = str(tmp_path / "o")
    a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=out, collapse=False, error_calc="poisson")
    assert cli.main(a, out=open(os.devnull, "w"), _no_fastio=line_parser) == 0
    good = open(out + ".qc.good.fasta").read() + open(out + ".qc.bad.fasta").read()
    assert all(">r%d" % k in good for k in range(4))
    for k, (s, q) in enumerate(recs):          # the decision the reference's function leads to
        e, ns = calculate_errors_poisson(s, [ord(c) - 33 for c in q], 0.005)
        where = "good" if e + ns <= len(s) * 0.01 else "bad"
        assert ">r%d" % k in open("%s.qc.%s.fasta" % (out, where)).read()


@pytest.mark.gpu
@pytest.mark.parametrize("line_parser", [False, True])
@pytest.mark.parametrize("method", ["poisson_binomial", "poisson_binomial_py"])
def test_long_reads_run_with_the_poisson_binomial_method_gpu(tmp_path, oracle, line_parser, method):
    """Round 3: reads of 1,100 / 2,500 / 4,000 bases go through the CLI under --error_calc poisson_binomial (and its
    _py alias, which the reference serves with the unlimited Python twin, moira/moira.py:820-821) instead of being
    refused; a long read of terrible quality (more than 1024 DP rows: k_wide) among them.  The decisions are the
    oracle's, which is pinned to the real reference at these lengths (tests/golden/long_reads.npz)."""
    fq = tmp_path / "long.fastq"
    rng = np.random.default_rng(8)
    recs = []
    for k, (n, lo, hi) in enumerate(((300, 25, 41), (1100, 30, 41), (2500, 33, 41), (2500, 1, 4), (4000, 35, 41), (700, 2, 41),
                                     (30000, 36, 41))):          # round 4: beyond 16383 bases
        q = "".join(chr(33 + int(v)) for v in rng.integers(lo, hi, n))
        s = "".join(rng.choice(list("ACGT"), n))
        recs.append((s, q))
    with open(fq, "w") as f:
        for k, (s, q) in enumerate(recs):
            f.write("@r%d\n%s\n+\n%s\n" % (k, s, q))
    out = str(tmp_path / "o")
    a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=out, collapse=False, error_calc=method)
    assert cli.main(a, out=open(os.devnull, "w"), _no_fastio=line_parser) == 0
    good, bad = open(out + ".qc.good.fasta").read(), open(out + ".qc.bad.fasta").read()
    n_good = 0
    for k, (s, q) in enumerate(recs):
        e, ns, rows = oracle.ee_rowwise(s, [ord(c) - 33 for c in q], 0.005)
        keep = e + ns <= len(s) * 0.01
        n_good += keep
        assert (">r%d\n" % k in good) == keep and (">r%d\t" % k in bad) == (not keep), (k, e, rows)
    assert 0 < n_good < len(recs)


def test_quality_above_254_is_scored_through_the_per_read_entry(tmp_path, oracle):
    """Under poisson_binomial a read whose scores do not fit the byte matrix takes the per-read entry (the reference
    takes any int, moira/bernoullimodule.c:92-108); the other reads of the chunk stay in the batch.  Here the oracle
    plays both parts; the GPU twin of this test is tests/test_gpu_bigq.py."""
    fa, qu = tmp_path / "r.fasta", tmp_path / "r.qual"
    tops = (40, 300, 41, 5000)
    with open(fa, "w") as f, open(qu, "w") as g:
        for k, top in enumerate(tops):
            f.write(">r%d\n%s\n" % (k, "ACGT" * 10))
            g.write(">r%d\n%s\n" % (k, " ".join(str(top if i == 7 else 3 if k == 3 else 30) for i in range(40))))
    calls = []
    backend = oracle_backend(oracle)

    def per_read(seq, quals, alpha):
        calls.append(max(quals))
        e, ns, _ = oracle.ee_rowwise(seq, quals, alpha)
        return e, ns
    backend.per_read = per_read
    out = str(tmp_path / "o")
    a = reference_args(paired=False, forward_fasta=str(fa), forward_qual=str(qu), output_prefix=out, collapse=False, uncert=0.1)
    assert cli.main(a, backend=backend, out=open(os.devnull, "w")) == 0
    assert calls == [300, 5000]
    good = open(out + ".qc.good.qual").read()
    bad = open(out + ".qc.bad.qual").read()
    assert ">r1\n" in good and " 300 " in good and ">r0\n" in good and ">r2\n" in good
    assert ">r3\t" in bad and " 5000 " in bad                     # Q3 everywhere else: far too many expected errors
    # a backend without the Poisson per-read entry still refuses under --error_calc poisson, naming the read
    import io
    msg = io.StringIO()
    a = reference_args(paired=False, forward_fasta=str(fa), forward_qual=str(qu), output_prefix=out + "p", collapse=False,
                       error_calc="poisson", silent=True)
    backend.methods = ("poisson_binomial", "poisson")
    assert cli.main(a, backend=backend, out=msg) == 1
    assert "r1" in msg.getvalue() and "300" in msg.getvalue()
