"""The CLI's non-default branches, pinned to the REFERENCE's output files.

tests/golden/flag_matrix/ holds what moira.py's own parse_fastq / parse_fasta_and_qual -> process_data -> write_results
write for 62 flag combinations (tests/golden/make_flag_matrix.py; every label of moira/moira.py:872-946, USEARCH headers :858-863,
--relabel :854-855, --round / --ambigs ignore :827-831, --only_contig :900-908, --min_overlap incl. the reference's
args.truncate slip in FASTQ mode :886-897, --trim_overlap, sum / posterior consensus scores, the Poisson method).
Every case must come out byte for byte through BOTH paths of the CLI: the byte-level path (moira_amd/fastio.py) and the
per-line Python path.  CPU: host logic with the oracle injected as the per-chunk filter; -m gpu: the product path.
"""
import hashlib
import math
import os
import types

import numpy as np
import pytest

import golden_io as G
from moira_amd import cli

GOLD = G.GOLDEN
MAN = G.flag_manifest()
CASES = sorted(MAN["cases"])


def _args(flags, **kw):
    d = dict(alpha=0.005, match=1, gap=-2, mismatch=-1, insert=20, deltaq=6, consensus_qscore="best",
             paired=False, truncate=None, only_contig=False, error_calc="poisson_binomial",
             ambigs="treat_as_errors", round=False, silent=True, nowarnings=False, doc=False, uncert=0.01,
             maxerrors=None, processors=4, forward_fasta=None, forward_qual=None, reverse_fasta=None,
             reverse_qual=None, forward_fastq=None, reverse_fastq=None, output_format="fasta", collapse=True,
             pipeline="mothur", fastq_offset=33, relabel=None, output_compression="none", qscore_cap=40,
             min_overlap=None, trim_overlap=False, bootstrap=100, output_prefix=None, device=None,
             fast_discard=False)
    d.update(flags)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.fixture(scope="module")
def inputs(tmp_path_factory):
    """{'shipped': (fwd, rev), 'derived': (fwd, rev)}; the derived pair is rebuilt from the committed test1 / test2 and
    must hash to what the generator fed the reference."""
    d = tmp_path_factory.mktemp("flag_inputs")
    r1 = G.read_fastq_records(os.path.join(GOLD, "test1.fastq.gz"))
    r2 = G.read_fastq_records(os.path.join(GOLD, "test2.fastq.bz2"))
    d1, d2 = G.derive_flag_inputs(r1, r2)
    paths = (str(d / "derived1.fastq"), str(d / "derived2.fastq"))
    G.write_fastq(paths[0], d1)
    G.write_fastq(paths[1], d2)
    got = [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in paths]
    assert got == MAN["inputs"]["derived"]["sha256"], "derived inputs differ from what the reference was run on"
    fq = ((str(d / "derived1.fasta"), str(d / "derived1.qual")), (str(d / "derived2.fasta"), str(d / "derived2.qual")))
    G.write_fasta_qual(fq[0][0], fq[0][1], d1)
    G.write_fasta_qual(fq[1][0], fq[1][1], d2)
    got = [hashlib.sha256(open(p, "rb").read()).hexdigest() for pair in fq for p in pair]
    assert got == MAN["inputs"]["derived_fasta_qual"]["sha256"], "derived fasta + qual inputs differ from the reference's"
    off64 = (str(d / "off64_1.fastq"), str(d / "off64_2.fastq"))
    quirks = (str(d / "quirks1.fastq"), str(d / "quirks2.fastq"))
    for k, recs in enumerate((d1, d2)):
        G.write_fastq_offset64(off64[k], recs)
        G.write_fastq_quirks(quirks[k], recs)
    for kind, pair in (("derived_offset64", off64), ("derived_quirks", quirks)):
        assert [hashlib.sha256(open(p, "rb").read()).hexdigest() for p in pair] == MAN["inputs"][kind]["sha256"], kind
    return {"shipped": (os.path.join(GOLD, "test1.fastq.gz"), os.path.join(GOLD, "test2.fastq.bz2")), "derived": paths,
            "derived_fasta_qual": fq, "derived_offset64": off64, "derived_quirks": quirks}


@pytest.fixture(scope="module")
def expected():
    return G.flag_outputs()


def oracle_backend(oracle):
    """Per-chunk filter for the CPU tests: the oracle (Poisson-binomial) and the reference's Poisson formula."""
    from poisson_ref import calculate_errors_poisson

    def poisson_one(seq, quals, alpha, ambigs, round_):
        e, ns = calculate_errors_poisson(seq, [int(v) for v in quals], alpha)
        e = e + ns if ambigs == "treat_as_errors" else e
        return math.floor(e) if round_ else e

    def backend(seqs, quals, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        quals = [ql.ints() if hasattr(ql, "ints") else ql for ql in quals]
        if method == "poisson":
            return [poisson_one(s, ql, alpha, ambigs, round_) for s, ql in zip(seqs, quals)]
        stride = 16 * ((max(len(s) for s in seqs) + 15) // 16)
        q = np.stack([oracle.pack_read(s, ql, stride) for s, ql in zip(seqs, quals)])
        lens = np.array([len(s) for s in seqs], np.int32)
        return oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)[0]

    def matrix(q, lens, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        if method == "poisson":                  # packed rows: 0 = 'N'; the byte-level path turns 'n' into a base first
            out = []
            for row, n in zip(q, lens):
                row = row[:n]
                seq = "".join("N" if v == 0 else "A" for v in row)
                out.append(poisson_one(seq, [int(v) if v else 1 for v in row], alpha, ambigs, round_))
            return np.array(out, np.float64)
        return oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)[0]
    backend.matrix = matrix
    backend.methods = ("poisson_binomial", "poisson")
    backend.per_read = lambda seq, quals, alpha: oracle.ee_rowwise(seq, [int(v) for v in quals], alpha)[:2]   # scores above 254
    return backend


def run_case(case, inputs, expected, tmp_path, backend, line_parser):
    spec = MAN["cases"][case]
    fwd, rev = inputs[spec["input"]]
    flags = dict(spec["flags"])
    paired = flags.get("paired", False) or flags.get("only_contig", False)
    out = str(tmp_path / "o")
    if isinstance(fwd, tuple):                   # the fasta + qual reader
        a = _args(flags, forward_fasta=fwd[0], forward_qual=fwd[1], reverse_fasta=rev[0] if paired else None,
                  reverse_qual=rev[1] if paired else None, output_prefix=out)
    else:
        a = _args(flags, forward_fastq=fwd, reverse_fastq=rev if paired else None, output_prefix=out)
    import io
    a.silent = False                             # the closing summary (moira/moira.py:508-519) is part of what is compared
    said = io.StringIO()
    assert cli.main(a, backend=backend, out=said, _no_fastio=line_parser) == 0
    n = spec["processed"]
    d_err, d_len, d_ov = spec["discarded_errors_minlength_minoverlap"]
    kept = n - d_err - d_len - d_ov
    lines = ["- Kept %d (%.2f%%) of the original sequences." % (kept, kept / n * 100)]
    if flags.get("truncate"):
        lines.append("- %d (%.2f%%) of the original sequences were discarded due to length < %s." % (d_len, d_len / n * 100, flags["truncate"]))
    if paired and flags.get("min_overlap"):
        lines.append("- %d (%.2f%%) of the original sequences were discarded due to paired-end reads having an overlap "
                     "length < %s." % (d_ov, d_ov / n * 100, flags["min_overlap"]))
    lines.append("- %d (%.2f%%) of the original sequences were discarded due to low quality." % (d_err, d_err / n * 100))
    text = said.getvalue()
    for line in lines:                           # the reference's counts (manifest) in the reference's words
        assert line in text, (line, text[-600:])
    want = expected[case]
    made = sorted(p[len("o."):] for p in os.listdir(tmp_path) if p.startswith("o."))
    assert made == sorted(spec["files"]), "output file set"
    for stem, info in spec["files"].items():
        got = open("%s.%s" % (out, stem), "rb").read()
        assert hashlib.sha256(want[stem]).hexdigest() == info["sha256"]            # the archive is what the manifest says
        if got != want[stem]:
            g, w = got.split(b"\n"), want[stem].split(b"\n")
            k = next((i for i in range(min(len(g), len(w))) if g[i] != w[i]), min(len(g), len(w)))
            raise AssertionError("%s / %s differs from the reference at line %d:\n got  %r\n want %r" % (
                case, stem, k + 1, g[k][:200] if k < len(g) else None, w[k][:200] if k < len(w) else None))


@pytest.mark.parametrize("line_parser", [False, True], ids=["bytes", "lines"])
@pytest.mark.parametrize("case", CASES)
def test_flag_matrix_host_logic(case, line_parser, inputs, expected, tmp_path, oracle):
    run_case(case, inputs, expected, tmp_path, oracle_backend(oracle), line_parser)


@pytest.fixture(scope="module")
def gpu_backend():
    b = cli.make_gpu_backend(None)
    yield b
    b.engine.close()


@pytest.mark.gpu
@pytest.mark.parametrize("line_parser", [False, True], ids=["bytes", "lines"])
@pytest.mark.parametrize("case", CASES)
def test_flag_matrix_gpu(case, line_parser, inputs, expected, tmp_path, gpu_backend):
    run_case(case, inputs, expected, tmp_path, gpu_backend, line_parser)


def test_the_matrix_covers_every_branch_of_write_results(expected):
    """Each label / header form of moira/moira.py:842-970 occurs in the reference's outputs we compare against."""
    blob = b"".join(v for fs in expected.values() for v in fs.values())
    for needle in (b"\tlength below 200\n", b"\toverlap length below 100\n", b"\toverlap length below None\n",
                   b"\toverlap length below 300\n", b"\tcontains ambiguities\n", b"\terrors > 1.00\n", b"\terrors > 2.50\n",
                   b"\tuncert > 0.010\n", b"\tuncert > 0.020\n", b";size=2;", b">x1\n", b"@s1;ee=", b">Otu_1;ee="):
        assert needle in blob, needle
    assert len(CASES) >= 62
    assert b" 300 " in blob and b" 120 " in blob                 # scores no FASTQ file can hold, through the fasta + qual reader
    kinds = {stem for fs in expected.values() for stem in fs}
    assert {"contigs.fasta", "contigs.names", "bad.contigs.fasta", "bad.contigs.names", "bad.contigs.fastq",
            "contigs.report", "qc.good.fastq", "qc.bad.fastq", "qc.good.names"} <= kinds
