"""libmoira_io.so (include/moira_io.h): the CLI's byte-level FASTQ path must be indistinguishable from
the line-by-line path that restates the reference (moira/moira.py:1152-1204 parse_fastq,
:842-970 write_results).  CPU only: the filter itself is injected (oracle)."""
import gzip
import io
import os
import re
import types

import numpy as np
import pytest

from moira_amd import cli, fastio as F
from moira_amd.py2dict import py2_str_hash

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = os.path.join(HERE, "golden")
ROOT = os.path.dirname(HERE)


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "moira_io.h")).read()
    names = set(re.findall(r"\b(mio_[a-z0-9_]+)\s*\(", hdr))
    assert names >= {"mio_fastq_index", "mio_pack", "mio_format", "mio_py2_hash", "mio_last_error", "mio_version"}
    lib = F.load()
    for n in names:
        assert hasattr(lib, n), n
    assert set(F.PROTOTYPES) == names


def make_fastq(rng, n, quirks=True, lo=20, hi=120):
    recs = []
    for i in range(n):
        L = int(rng.integers(lo, hi))
        seq = "".join(rng.choice(list("ACGTNn"), L, p=[.24, .24, .24, .24, .03, .01]))
        qual = "".join(chr(33 + int(x)) for x in rng.integers(0, 42, L))
        hdr = "@M0:%d:x%d" % (i % 7, i)
        if quirks:
            k = i % 6
            if k == 1:
                hdr = "@@" + hdr[1:] + " extra words"
            elif k == 2:
                hdr = hdr + "\tTAB field"
            elif k == 3:
                hdr = "  " + hdr + "  "
        eol = "\r\n" if (quirks and i % 5 == 0) else "\n"
        pad = " \t" if (quirks and i % 4 == 0) else ""
        recs.append(hdr + eol + seq + pad + eol + "+" + eol + qual + pad + eol)
    return "".join(recs)


def slow_records(text, offset=33):
    fh = io.StringIO(text, newline=None)
    fh.moira_name = "x"
    return list(cli.parse_fastq(fh, None, offset, raw=True))


@pytest.mark.parametrize("final_newline", [True, False])
def test_index_matches_line_parser(final_newline):
    rng = np.random.default_rng(3)
    text = make_fastq(rng, 500)
    if not final_newline:
        text = text.rstrip("\r\n ")
    text_extra = text + ("\n@dangling\nACGT\n" if final_newline else "")   # lines that do not make a record
    want = slow_records(text_extra)
    buf = text_extra.encode()
    idx, consumed, err = F.index(buf, True, 10 ** 6)
    assert err is None and len(idx) == len(want) == 500
    for row, (h, s, q, _, _) in zip(idx, want):
        assert F.header_of(buf, row) == h
        assert buf[row[F.SEQ_OFF]:row[F.SEQ_OFF] + row[F.SEQ_LEN]].decode() == s
        assert buf[row[F.QUAL_OFF]:row[F.QUAL_OFF] + row[F.QUAL_LEN]].decode() == q.s
    # streamed in small blocks: same records, whatever the block boundaries cut through
    got = []
    for b, ix in F.FastqChunks(io.BytesIO(buf), max_records=37, block_bytes=1000):
        got += [(F.header_of(b, r), bytes(b[r[F.SEQ_OFF]:r[F.SEQ_OFF] + r[F.SEQ_LEN]]).decode()) for r in ix]
    assert got == [(h, s) for h, s, _, _, _ in want]


def test_index_reports_the_reference_errors_and_unsupported_content():
    ok = "@a\nACGT\n+\nIIII\n"
    for bad, kind in (("@b\n\n+\nIIII\n", F.REC_EMPTY_SEQ), ("@b\nACGT\n+\n\n", F.REC_EMPTY_QUAL),
                      ("@b\nACGT\n+\nIII\n", F.REC_LENGTH_MISMATCH)):
        idx, consumed, err = F.index((ok + bad + ok).encode(), True, 100)
        assert len(idx) == 1 and consumed == len(ok) and err.kind == kind and err.header == "b"
    with pytest.raises(F.Unsupported):
        F.index(b"@a\nAC\rGT\n+\nIIII\n", True, 10)              # lone CR: a line break for the text parser
    with pytest.raises(F.Unsupported):
        F.index("@a\u00a0\nACGT\n+\nIIII\n".encode(), True, 10)   # non-ASCII: Unicode strip rules


def test_pack_follows_the_packing_rules():
    rng = np.random.default_rng(5)
    text = make_fastq(rng, 300, quirks=False)
    recs = slow_records(text)
    buf = text.encode()
    idx, _, _ = F.index(buf, True, 1000)
    for T, lower in ((0, False), (50, False), (0, True)):
        sel = np.arange(len(idx))[::2]
        q, lens, has_n = F.pack(buf, idx, sel, 33, T, lower_n_is_base=lower, stride=128)
        for k, i in enumerate(sel):
            _, s, ql, _, _ = recs[i]
            s, qi = (s[:T], ql.ints()[:T]) if T else (s, ql.ints())          # ints(): ord - offset, Q0 -> 1
            want = np.array([0 if c == "N" else (255 if (c == "n" and not lower) else v) for c, v in zip(s, qi)], np.uint8)
            assert lens[k] == len(s) and np.array_equal(q[k, :len(s)], want) and not q[k, len(s):].any()
            assert has_n[k] == ("N" in s)
    with pytest.raises(ValueError, match="positive"):
        b = b"@a\nACGT\n+\nII!I\n"
        ix, _, _ = F.index(b, True, 10)
        F.pack(b, ix, None, 64, 0, stride=16)                                # '!' - 64 < 0
    with pytest.raises(ValueError, match="does not fit"):
        F.pack(buf, idx, None, 33, 0, stride=16)


def test_py2_hash_matches_the_python_restatement():
    rng = np.random.default_rng(7)
    text = make_fastq(rng, 200, quirks=False)
    buf = text.encode()
    idx, _, _ = F.index(buf, True, 1000)
    for T in (0, 33):
        hs = F.py2_hashes(buf, idx, T)
        for h, (_, s, _, _, _) in zip(hs, slow_records(text)):
            assert h == py2_str_hash((s[:T] if T else s).encode())


def test_format_matches_the_python_writer():
    rng = np.random.default_rng(9)
    text = make_fastq(rng, 120)
    buf = text.encode()
    idx, _, _ = F.index(buf, True, 1000)
    recs = slow_records(text)
    sel = np.arange(len(idx))[1::3]
    ee = rng.random(len(sel)) * 30
    lab = (np.arange(len(sel)) % 3 - 1).astype(np.int32)
    labels = ["uncert > 0.010", "contains ambiguities"]
    for T in (0, 40):
        for relabel in (None, "seq_"):
            for use_ee in (False, True):
                want = {F.FMT_FASTA: "", F.FMT_QUAL: "", F.FMT_FASTQ: ""}
                for k, i in enumerate(sel):
                    h, s, ql, _, _ = recs[i]
                    if T:
                        s, ql = s[:T], ql[:T]
                    if relabel:
                        h = "%s%d" % (relabel, 1000 + k)
                    if use_ee:
                        h += ";ee=%.2f;size=%d;" % (ee[k], 1)
                    if lab[k] >= 0:
                        h += "\t" + labels[lab[k]]
                    want[F.FMT_FASTA] += ">%s\n%s\n" % (h, s)
                    want[F.FMT_QUAL] += ">%s\n%s\n" % (h, " ".join(map(str, ql.ints())))
                    want[F.FMT_FASTQ] += "@%s\n%s\n+\n%s\n" % (h, s, "".join(chr(v + 33) for v in ql.ints()))
                for kind in want:
                    got = F.format_records(buf, idx, sel, kind, 33, T, relabel=relabel,
                                           relabel_index=1000 + np.arange(len(sel)), ee=ee if use_ee else None,
                                           labels=labels, label_id=lab)
                    assert bytes(got).decode() == want[kind]


def matrix_backend(oracle):
    """The injected filter for packed matrices (what make_gpu_backend().matrix is on a GPU box)."""
    def backend(seqs, quals, alpha, ambigs, round_):
        stride = 16 * ((max(len(s) for s in seqs) + 15) // 16)
        quals = [ql.ints() if hasattr(ql, "ints") else ql for ql in quals]
        q = np.stack([oracle.pack_read(s, ql, stride) for s, ql in zip(seqs, quals)])
        lens = np.array([len(s) for s in seqs], np.int32)
        return oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)[0]

    def matrix(q, lens, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        assert method == "poisson_binomial" and fast_discard is None
        return oracle.filter_batch(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, threads=4)[0]
    backend.matrix = matrix
    backend.methods = ("poisson_binomial",)
    return backend


def args_for(path, out, **kw):
    d = dict(alpha=0.005, match=1, gap=-2, mismatch=-1, insert=20, deltaq=6, consensus_qscore="best",
             paired=False, truncate=None, only_contig=False, error_calc="poisson_binomial",
             ambigs="treat_as_errors", round=False, silent=True, nowarnings=False, doc=False, uncert=0.01,
             maxerrors=None, processors=2, forward_fasta=None, forward_qual=None, reverse_fasta=None,
             reverse_qual=None, forward_fastq=path, reverse_fastq=None, output_format="fasta", collapse=False,
             pipeline="mothur", fastq_offset=33, relabel=None, output_compression="none", qscore_cap=40,
             min_overlap=None, trim_overlap=False, bootstrap=100, output_prefix=out, device=None,
             fast_discard=False)
    d.update(kw)
    return types.SimpleNamespace(**d)


def outputs_of(prefix):
    d = os.path.dirname(prefix)
    res = {}
    for f in sorted(os.listdir(d)):
        if f.startswith(os.path.basename(prefix) + "."):
            p = os.path.join(d, f)
            res[f.split(".", 1)[1]] = (gzip.open(p).read() if f.endswith(".gz") else open(p, "rb").read())
    return res


CASES = [
    dict(),
    dict(collapse=True),
    dict(collapse=True, pipeline="USEARCH"),
    dict(output_format="fastq"),
    dict(output_format="fastq", collapse=True, truncate=60),
    dict(truncate=60, relabel="r"),
    dict(pipeline="USEARCH", maxerrors=1.5),
    dict(ambigs="disallow", round=True),
    dict(ambigs="ignore", output_compression="gz", uncert=0.05),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_cli_byte_level_path_equals_line_path(tmp_path, oracle, monkeypatch, case):
    rng = np.random.default_rng(100 + case)
    # duplicates so that collapse has groups, a few Q0 bases, ragged lengths across two buckets
    base = make_fastq(rng, 150, quirks=True, lo=30, hi=100)
    text = base + make_fastq(np.random.default_rng(100 + case), 60, quirks=True, lo=30, hi=100)
    src = tmp_path / "in.fastq"
    src.write_bytes(text.encode())
    kw = CASES[case]
    backend = matrix_backend(oracle)
    calls = []
    real = cli._run_fast_fastq
    monkeypatch.setattr(cli, "_run_fast_fastq", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(F, "PARALLEL_MIN", 16 if case % 2 else 4096)      # odd cases: packing/formatting split over threads
    assert cli.main(args_for(str(src), str(tmp_path / "fast"), processors=3, **kw), backend=backend,
                    out=open(os.devnull, "w")) == 0
    assert calls == [1]                                                  # the byte-level path really ran
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    assert cli.main(args_for(str(src), str(tmp_path / "slow"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    assert calls == [1]
    fast, slow = outputs_of(str(tmp_path / "fast")), outputs_of(str(tmp_path / "slow"))
    assert fast.keys() == slow.keys() and len(fast) >= 2
    for k in fast:
        assert fast[k] == slow[k], k
    assert sum(len(v) for v in fast.values()) > 1000


def test_cli_falls_back_to_the_line_parser(tmp_path, oracle):
    text = make_fastq(np.random.default_rng(1), 40, quirks=False).replace("\n", "\r")      # old-Mac line ends
    src = tmp_path / "cr.fastq"
    src.write_bytes(text.encode())
    backend = matrix_backend(oracle)
    assert cli.main(args_for(str(src), str(tmp_path / "a")), backend=backend, out=open(os.devnull, "w")) == 0
    os.environ["MOIRA_NO_FASTIO"] = "1"
    try:
        assert cli.main(args_for(str(src), str(tmp_path / "b")), backend=backend, out=open(os.devnull, "w")) == 0
    finally:
        del os.environ["MOIRA_NO_FASTIO"]
    a, b = outputs_of(str(tmp_path / "a")), outputs_of(str(tmp_path / "b"))
    assert a == b and sum(len(v) for v in a.values()) > 100


def test_cli_raises_the_reference_exceptions(tmp_path, oracle):
    src = tmp_path / "bad.fastq"
    src.write_bytes(b"@a\nACGT\n+\nIIII\n@b:1\nACGT\n+\nIII\n")
    with pytest.raises(cli.LengthMismatchError) as e:
        cli.main(args_for(str(src), str(tmp_path / "o")), backend=matrix_backend(oracle), out=open(os.devnull, "w"))
    assert "b_1" in str(e.value)


# ---- paired input: contigs built in C from the two file buffers -------------------------------------
def make_pairs(rng, n, read_len=(60, 110), frag=(90, 160)):
    comp = str.maketrans("ACGTN", "TGCAN")
    f_recs, r_recs = [], []
    for i in range(n):
        L1, L2 = int(rng.integers(*read_len)), int(rng.integers(*read_len))
        F_ = max(int(rng.integers(*frag)), L1, L2)
        if i % 9 == 0:
            F_ = L1 + L2 + 5                                   # no real overlap
        fragment = "".join(rng.choice(list("ACGT"), F_))
        if i % 7 == 3:
            fragment = fragment[:F_ // 2] + fragment[:F_ - F_ // 2]     # a repeat: several equally good alignments
        fwd = list(fragment[:L1])
        rev = list(fragment[F_ - L2:][::-1].translate(comp))
        for read in (fwd, rev):
            for k in range(len(read)):
                if rng.random() < 0.03:
                    read[k] = str(rng.choice(list("ACGTN")))
        if i % 11 == 5:
            k = int(rng.integers(5, L1 - 5))
            del fwd[k]                                          # an indel
        q1 = "".join(chr(33 + int(x)) for x in np.clip(38 - rng.integers(0, 36, len(fwd)) * (np.arange(len(fwd)) / len(fwd)), 0, 41))
        q2 = "".join(chr(33 + int(x)) for x in np.clip(38 - rng.integers(0, 36, len(rev)) * (np.arange(len(rev)) / len(rev)), 0, 41))
        h = "@M1:%d:p%d" % (i % 5, i if i >= 40 else i % 20)    # a few repeated names do no harm
        f_recs.append("%s 1:N:0\n%s\n+\n%s\n" % (h, "".join(fwd), q1))
        r_recs.append("%s 2:N:0\n%s\n+\n%s\n" % (h, "".join(rev), q2))
    dup = list(range(0, n, 6))                                   # exact duplicates so that collapse has groups
    return "".join(f_recs + [f_recs[i] for i in dup]), "".join(r_recs + [r_recs[i] for i in dup])


PAIRED_CASES = [
    dict(),
    dict(collapse=True),
    dict(collapse=True, output_format="fastq", min_overlap=30),
    dict(min_overlap=40, truncate=120, relabel="c"),
    dict(collapse=True, pipeline="USEARCH", consensus_qscore="sum", qscore_cap=0),
    dict(consensus_qscore="posterior", ambigs="disallow", trim_overlap=True),
    dict(collapse=True, maxerrors=2.0, truncate=100, min_overlap=35),
    dict(only_contig=True, collapse=True),
    dict(only_contig=True, output_format="fastq", truncate=110, min_overlap=30),
    dict(only_contig=True, collapse=True, pipeline="USEARCH", relabel="ctg", min_overlap=40),
]


@pytest.mark.parametrize("case", range(len(PAIRED_CASES)))
def test_cli_byte_level_path_equals_line_path_paired(tmp_path, oracle, monkeypatch, case):
    f_text, r_text = make_pairs(np.random.default_rng(500 + case), 130)
    (tmp_path / "f.fastq").write_bytes(f_text.encode())
    with gzip.open(tmp_path / "r.fastq.gz", "wb") as fh:
        fh.write(r_text.encode())
    kw = dict(paired=True, reverse_fastq=str(tmp_path / "r.fastq.gz"), **PAIRED_CASES[case])
    backend = matrix_backend(oracle)
    calls = []
    real = cli._run_fast_fastq
    monkeypatch.setattr(cli, "_run_fast_fastq", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(cli, "PAIR_CHUNK_READS", 37)             # several chunks
    src = str(tmp_path / "f.fastq")
    assert cli.main(args_for(src, str(tmp_path / "fast"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    assert calls == [1]
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    assert cli.main(args_for(src, str(tmp_path / "slow"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    assert calls == [1]
    fast, slow = outputs_of(str(tmp_path / "fast")), outputs_of(str(tmp_path / "slow"))
    assert fast.keys() == slow.keys() and "contigs.report" in fast
    for k in fast:
        assert fast[k] == slow[k], k
    assert sum(len(v) for v in fast.values()) > 5000


def test_paired_errors_are_the_line_parsers(tmp_path, oracle):
    ok_f, ok_r = "@a\nACGTACGTAC\n+\nIIIIIIIIII\n", "@a\nGTACGTACGT\n+\nIIIIIIIIII\n"
    cases = [
        (ok_f + "@b\nACGT\n+\nIIII\n", ok_r + "@c\nACGT\n+\nIIII\n", cli.NameMismatchError, "'b'"),
        (ok_f + "@b\nACGT\n+\nIIII\n", ok_r + "@b:x\nACGT\n+\nIII\n", cli.LengthMismatchError, "b_x"),
        (ok_f + "@b\nACGT\n+\nIII\n", ok_r + "@b\nACGT\n+\nIII\n", cli.LengthMismatchError, "f.fastq"),
        (ok_f + "@bf\nACGT\n+\nIIII\n", ok_r + "@br\n\n+\nIIII\n", cli.EmptySeqError, "bf"),
    ]
    for f_text, r_text, exc, needle in cases:
        (tmp_path / "f.fastq").write_bytes(f_text.encode())
        (tmp_path / "r.fastq").write_bytes(r_text.encode())
        msgs = []
        for env in (None, "1"):
            if env:
                os.environ["MOIRA_NO_FASTIO"] = env
            try:
                with pytest.raises(exc) as e:
                    cli.main(args_for(str(tmp_path / "f.fastq"), str(tmp_path / "o"), paired=True,
                                      reverse_fastq=str(tmp_path / "r.fastq")),
                             backend=matrix_backend(oracle), out=open(os.devnull, "w"))
                msgs.append(str(e.value))
            finally:
                os.environ.pop("MOIRA_NO_FASTIO", None)
        assert msgs[0] == msgs[1] and needle in msgs[0], msgs
    # the shorter file ends the run, as zip() does
    (tmp_path / "f.fastq").write_bytes((ok_f * 3).encode())
    (tmp_path / "r.fastq").write_bytes((ok_r * 2).encode())
    assert cli.main(args_for(str(tmp_path / "f.fastq"), str(tmp_path / "o"), paired=True,
                             reverse_fastq=str(tmp_path / "r.fastq")), backend=matrix_backend(oracle),
                    out=open(os.devnull, "w")) == 0
    assert open(str(tmp_path / "o") + ".contigs.report").read().count("\n") == 3     # header + 2 pairs


def test_random_text_noise_fast_equals_line_path(tmp_path, oracle):
    """Randomly mangled FASTQ text (blank lines, stray whitespace, CRLF, '@'/'+' in odd places, Q0 bases,
    truncated tails, headers made of marks only): both paths must write the same files or raise the
    same exception."""
    backend = matrix_backend(oracle)
    rng = np.random.default_rng(2024)
    ws = [" ", "\t", " \t ", "\x0b", "\x0c", "\x1c", ""]
    outcomes = {"same_files": 0, "same_error": 0}
    for trial in range(60):
        lines = []
        for i in range(int(rng.integers(1, 30))):
            L = int(rng.integers(1, 70))
            seq = "".join(rng.choice(list("ACGTNn"), L, p=[.23, .23, .23, .23, .05, .03]))
            qual = "".join(chr(33 + int(x)) for x in rng.integers(0, 42, L))
            hdr = str(rng.choice(["@r%d" % i, "@@r:%d x" % i, "@", "@@@", "r%d" % i, "@a:b:c\td", "  @sp%d  " % i]))
            rec = [hdr, seq, str(rng.choice(["+", "+" + hdr[1:], "+ junk"])), qual]
            r = rng.random()
            if r < 0.04:
                rec[1] = ""                          # empty sequence line
            elif r < 0.08:
                rec[3] = ""                          # empty quality line
            elif r < 0.12:
                rec[3] = rec[3][:-1]                 # length mismatch
            elif r < 0.16:
                rec.insert(int(rng.integers(0, 4)), "")      # a blank line shifts the 4-line framing
            lines += rec
        eol = str(rng.choice(["\n", "\r\n"]))
        text = "".join(str(rng.choice(ws)) + ln + str(rng.choice(ws)) + eol for ln in lines)
        if rng.random() < 0.3:
            text = text.rstrip("\r\n")
        if rng.random() < 0.2:
            text = text[:int(len(text) * rng.random())]     # cut anywhere
        src = tmp_path / ("t%d.fastq" % trial)
        src.write_bytes(text.encode())
        kw = dict(collapse=bool(trial % 2), output_format="fastq" if trial % 3 == 0 else "fasta")
        res = []
        for mode in ("fast", "slow"):
            if mode == "slow":
                os.environ["MOIRA_NO_FASTIO"] = "1"
            try:
                try:
                    rc = cli.main(args_for(str(src), str(tmp_path / ("%s%d" % (mode, trial))), **kw), backend=backend,
                                  out=open(os.devnull, "w"))
                    res.append(("ok", rc, outputs_of(str(tmp_path / ("%s%d" % (mode, trial))))))
                except (cli.EmptySeqError, cli.EmptyQualError, cli.LengthMismatchError, ValueError) as e:
                    res.append(("err", type(e).__name__, str(e)))
            finally:
                os.environ.pop("MOIRA_NO_FASTIO", None)
        assert res[0] == res[1], (trial, text[:300], res[0][:2], res[1][:2])
        outcomes["same_files" if res[0][0] == "ok" else "same_error"] += 1
    assert outcomes["same_files"] >= 10 and outcomes["same_error"] >= 10, outcomes


@pytest.mark.gpu
@pytest.mark.parametrize("kw", [dict(), dict(collapse=True), dict(error_calc="poisson"), dict(fast_discard=True),
                                dict(paired=True, collapse=True, min_overlap=30)])
def test_byte_level_path_equals_line_path_on_the_gpu(tmp_path, kw):
    """Product path (HIP library, no injected backend): both CLI paths write the same files."""
    rng = np.random.default_rng(77)
    kw = dict(kw)
    if kw.get("paired"):
        f_text, r_text = make_pairs(rng, 3000)
        (tmp_path / "r.fastq").write_bytes(r_text.encode())
        kw["reverse_fastq"] = str(tmp_path / "r.fastq")
    else:
        f_text = make_fastq(rng, 6000, quirks=True, lo=30, hi=320)
    (tmp_path / "f.fastq").write_bytes(f_text.encode())
    src = str(tmp_path / "f.fastq")
    assert cli.main(args_for(src, str(tmp_path / "fast"), processors=4, **kw), out=open(os.devnull, "w")) == 0
    os.environ["MOIRA_NO_FASTIO"] = "1"
    try:
        assert cli.main(args_for(src, str(tmp_path / "slow"), processors=4, **kw), out=open(os.devnull, "w")) == 0
    finally:
        del os.environ["MOIRA_NO_FASTIO"]
    fast, slow = outputs_of(str(tmp_path / "fast")), outputs_of(str(tmp_path / "slow"))
    assert fast.keys() == slow.keys()
    for k in fast:
        assert fast[k] == slow[k], k
    assert sum(len(v) for v in fast.values()) > 100000


# ---- fasta + qual input --------------------------------------------------------------------------------
def fastq_to_fasta_qual(text, offset=33):
    fa, qu = [], []
    lines = text.split("\n")
    for i in range(0, len(lines) - 3, 4):
        h, s, _, q = lines[i:i + 4]
        fa.append(">%s\n%s\n" % (h[1:], s))
        qu.append(">%s\n%s\n" % (h[1:], " ".join(str(ord(c) - offset) for c in q)))
    return "".join(fa), "".join(qu)


FQ_CASES = [dict(), dict(collapse=True, output_format="fastq"), dict(output_format="fastq", fastq_offset=64, truncate=70),
            dict(collapse=True, pipeline="USEARCH", ambigs="disallow")]


@pytest.mark.parametrize("case", range(len(FQ_CASES)))
def test_fasta_qual_byte_level_path_equals_line_path(tmp_path, oracle, monkeypatch, case):
    text = make_fastq(np.random.default_rng(900 + case), 400, quirks=False, lo=30, hi=150)
    text += make_fastq(np.random.default_rng(900 + case), 80, quirks=False, lo=30, hi=150)       # duplicates
    fa, qu = fastq_to_fasta_qual(text)
    (tmp_path / "in.fasta").write_bytes(fa.encode())
    (tmp_path / "in.qual").write_bytes(qu.replace(" ", "\t", 3).encode())                       # a few tabs
    kw = dict(forward_fasta=str(tmp_path / "in.fasta"), forward_qual=str(tmp_path / "in.qual"), **FQ_CASES[case])
    backend = matrix_backend(oracle)
    calls = []
    real = cli._run_fast_fastq
    monkeypatch.setattr(cli, "_run_fast_fastq", lambda *a, **k: (calls.append(1), real(*a, **k))[1])
    monkeypatch.setattr(cli, "CHUNK_READS", 97)
    assert cli.main(args_for(None, str(tmp_path / "fast"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    assert calls == [1]
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    assert cli.main(args_for(None, str(tmp_path / "slow"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    fast, slow = outputs_of(str(tmp_path / "fast")), outputs_of(str(tmp_path / "slow"))
    assert fast.keys() == slow.keys() and len(fast) >= 2
    for k in fast:
        assert fast[k] == slow[k], k
    assert sum(len(v) for v in fast.values()) > 20000


def test_fasta_qual_paired_and_fallbacks(tmp_path, oracle, monkeypatch):
    f_text, r_text = make_pairs(np.random.default_rng(31), 120)
    names = {}
    for tag, text in (("f", f_text), ("r", r_text)):
        fa, qu = fastq_to_fasta_qual(text)
        (tmp_path / (tag + ".fasta")).write_bytes(fa.encode())
        (tmp_path / (tag + ".qual")).write_bytes(qu.encode())
        names[tag] = (str(tmp_path / (tag + ".fasta")), str(tmp_path / (tag + ".qual")))
    kw = dict(paired=True, collapse=True, forward_fasta=names["f"][0], forward_qual=names["f"][1],
              reverse_fasta=names["r"][0], reverse_qual=names["r"][1], min_overlap=25)
    backend = matrix_backend(oracle)
    monkeypatch.setattr(cli, "PAIR_CHUNK_READS", 41)
    assert cli.main(args_for(None, str(tmp_path / "fast"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    monkeypatch.setenv("MOIRA_NO_FASTIO", "1")
    assert cli.main(args_for(None, str(tmp_path / "slow"), **kw), backend=backend, out=open(os.devnull, "w")) == 0
    monkeypatch.delenv("MOIRA_NO_FASTIO")
    fast, slow = outputs_of(str(tmp_path / "fast")), outputs_of(str(tmp_path / "slow"))
    assert fast == slow and "contigs.report" in fast and sum(len(v) for v in fast.values()) > 10000
    # what the byte-level parser declines goes to the line parser, which raises the reference's exceptions
    (tmp_path / "bad.fasta").write_bytes(b">a\nACGT\n>b\nACGT\n")
    for qual, exc in ((b">a\n30 30 30 30\n>c\n30 30 30 30\n", cli.NameMismatchError),
                      (b">a\n30 30 30 30\n>b\n30 30 30\n", cli.LengthMismatchError),
                      (b">a\n30 30 30 30\n>b\n30 x 30 30\n", ValueError)):
        (tmp_path / "bad.qual").write_bytes(qual)
        with pytest.raises(exc):
            cli.main(args_for(None, str(tmp_path / "o"), forward_fasta=str(tmp_path / "bad.fasta"),
                              forward_qual=str(tmp_path / "bad.qual")), backend=backend, out=open(os.devnull, "w"))


def test_parallel_indexer_equals_the_sequential_one():
    """mio_fastq_index_mt (newline counts per byte slice -> record starts -> one sequential indexer per range) against
    mio_fastq_index on quirky buffers: quality lines that start with '@' or '+', CRLF, a cut last record, no final
    newline, records that fail the reference's checks, lone CR / non-ASCII bytes (-> Unsupported), every kind of
    max_records cap.  Rows, consumed, the offending record and the refusal must be identical."""
    from moira_amd import fastio as F
    rng = np.random.default_rng(11)

    def make(nrec, crlf, quirk):
        out = []
        for i in range(nrec):
            L = int(rng.integers(1, 120))
            seq = bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L))
            qual = bytes(rng.integers(33, 75, L).astype(np.uint8))
            if rng.random() < 0.1:
                qual = b"@" + qual[1:]
            if rng.random() < 0.05:
                qual = b"+" + qual[1:]
            lines = [b"@r%d some:text\tmore" % i, seq, b"+" if rng.random() < 0.8 else b"+r%d" % i, qual]
            if rng.random() < quirk:
                k = int(rng.integers(0, 4))
                if k == 0:
                    lines[1] = b""
                elif k == 1:
                    lines[3] = b"  "
                elif k == 2:
                    lines[3] = qual + b"I"
                else:
                    lines[0] = b"@@weird "
            nl = b"\r\n" if crlf else b"\n"
            out.append(nl.join(lines) + nl)
        return b"".join(out)

    def run(buf, final, maxr, th):
        try:
            idx, cons, err = F.index(buf, final, maxr, th)
            return ("ok", idx.tobytes(), cons, None if err is None else (err.kind, err.header))
        except F.Unsupported:
            return ("unsupported",)

    engaged = 0
    for it in range(40):
        nrec = int(rng.integers(3000, 9000))
        buf = make(nrec, rng.random() < 0.3, rng.choice([0, 0, 0.0002, 0.001]))
        mode = it % 5
        if mode == 1:
            buf = buf[:-int(rng.integers(1, 300))]
        if mode == 2:
            buf = buf.rstrip(b"\r\n")
        if mode == 3:
            p = int(rng.integers(0, len(buf)))
            buf = buf[:p] + b"\xc3" + buf[p + 1:]
        if mode == 4 and it % 2:
            p = int(rng.integers(0, len(buf)))
            buf = buf[:p] + b"\r" + buf[p + 1:]
        final = bool(it & 1)
        maxr = int(rng.choice([1, 100, nrec // 2, nrec, nrec + 10, 10 ** 6]))
        want = run(buf, final, maxr, 1)
        for th in (2, 3, 5, 8):
            engaged += len(buf) >= th * (64 << 10)
            assert run(buf, final, maxr, th) == want, (it, th, mode, final, maxr)
    assert engaged > 100                     # the buffers were large enough for the threaded path to run


def test_plain_file_reader_with_threads_equals_the_stream_reader(tmp_path):
    """FastqChunks(threads > 1) on an uncompressed file (concurrent preads into one fresh buffer per block, the
    partial record carried over) yields the same records as the one-thread stream reader."""
    from moira_amd import fastio as F
    rng = np.random.default_rng(12)
    path = tmp_path / "r.fastq"
    with open(path, "wb") as f:
        for i in range(60000):
            L = int(rng.integers(30, 200))
            f.write(b"@read%d\n" % i + bytes(rng.choice(np.frombuffer(b"ACGT", np.uint8), L)) + b"\n+\n"
                    + bytes(rng.integers(33, 74, L).astype(np.uint8)) + b"\n")
        f.write(b"@dangling\nACGT\n")                       # trailing lines that do not make a record: dropped

    def records(threads, block):
        out = []
        with open(path, "rb") as fh:
            for buf, idx in F.FastqChunks(fh, 7000, block_bytes=block, threads=threads):
                assert len(idx) <= 7000
                for r in idx:
                    out.append((F.header_of(buf, r), bytes(buf[r[F.SEQ_OFF]:r[F.SEQ_OFF] + r[F.SEQ_LEN]]),
                                bytes(buf[r[F.QUAL_OFF]:r[F.QUAL_OFF] + r[F.QUAL_LEN]])))
        return out
    want = records(1, 1 << 20)
    assert len(want) == 60000 and want[-1][0] == "read59999"
    assert records(4, 1 << 20) == want


def test_sharded_collapse_is_the_python2_dict(tmp_path):
    """The collapse keeps its groups in 64 shards filled by several threads and rebuilds the Python-2 dict ORDER at
    export time from the order in which the distinct sequences first appeared (moira/moira.py:459-475, :492).  40,000
    reads with heavy duplication, ties in the expected errors and in the abundances: (1) one thread and eight threads
    give identical exports and identical formatted files; (2) the group order equals what moira_amd/py2dict.py (the
    Python restatement of CPython 2.7's dict) yields for the same insertion sequence, sorted by abundance."""
    from moira_amd.py2dict import Py2Dict
    rng = np.random.default_rng(21)
    pool = ["".join(rng.choice(list("ACGT"), int(rng.integers(20, 60)))) for _ in range(9000)]
    n = 40000
    pick = np.minimum((rng.random(n) ** 3 * len(pool)).astype(int), len(pool) - 1)      # a few very abundant, many singletons
    recs = []
    for i in range(n):
        sq = pool[pick[i]]
        recs.append("@r:%d\n%s\n+\n%s\n" % (i, sq, "".join(chr(33 + int(v)) for v in rng.integers(2, 41, len(sq)))))
    buf = "".join(recs).encode()
    idx, consumed, err = F.index(buf, True, n)
    assert len(idx) == n and err is None
    ee = np.round(rng.random(n) * 3, 1)                                              # one decimal: plenty of ties
    flags = (rng.random(n) < 0.05).astype(np.uint8)
    outs = []
    for threads in (1, 8):
        g = F.Collapse(threads)
        for lo in range(0, n, 15000):                                               # three chunks: 15000 > the 8192 threshold
            g.add(buf, idx[lo:lo + 15000], ee[lo:lo + 15000], flags[lo:lo + 15000])
        gee, glen, gsize, gfl, gaux = g.export()
        sel = np.arange(len(gee))
        files = [bytes(g.format(sel, kind)) for kind in (F.FMT_FASTA, F.FMT_QUAL, F.FMT_NAMES, F.FMT_FASTQ)]
        outs.append((gee.tobytes(), glen.tobytes(), gsize.tobytes(), gfl.tobytes(), files))
        if threads == 1:
            seqs_in_order = [l for l in files[0].decode().split("\n")[1::2]]
            sizes = gsize.copy()
        g.close()
    assert outs[0] == outs[1]
    d = Py2Dict()
    count = {}
    for i in range(n):
        sq = pool[pick[i]]
        if sq not in d:
            d[sq] = True
        count[sq] = count.get(sq, 0) + 1
    want = sorted(list(d), key=lambda k: count[k], reverse=True)                    # stable, on dict order: moira.py:492
    assert seqs_in_order == want
    assert [count[k] for k in want] == sizes.tolist()


@pytest.mark.parametrize("kind", ["gz", "bz2"])
def test_block_compressed_outputs_hold_the_same_text(tmp_path, oracle, monkeypatch, kind):
    """Compressed outputs of the byte-level path are written as independently compressed blocks (gzip members / bzip2
    streams) so that several threads compress them; read back with the standard modules they must hold exactly the
    text of an uncompressed run.  The block size is shrunk so that every file has many members."""
    import bz2
    from test_cli_golden import oracle_backend, reference_args
    monkeypatch.setattr(cli._BlockCompressedWriter, "BLOCK", 700)
    monkeypatch.setattr(cli._BlockCompressedWriter, "WINDOW", 3)
    rng = np.random.default_rng(31)
    fq = tmp_path / "r.fastq"
    fq.write_text(make_fastq(rng, 900, quirks=False))
    outs = {}
    for comp in ("none", kind):
        pre = str(tmp_path / ("o_" + comp))
        a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=pre, collapse=True, output_compression=comp,
                           processors=4)
        assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
        opener = {"none": open, "gz": gzip.open, "bz2": bz2.open}[comp]
        outs[comp] = {f.split(".", 1)[1].replace("." + kind, ""): opener(tmp_path / f, "rb").read()
                      for f in sorted(os.listdir(tmp_path)) if f.startswith("o_" + comp + ".")}
    assert outs["none"].keys() == outs[kind].keys() and len(outs["none"]) >= 6
    assert outs["none"] == outs[kind]
    assert sum(len(v) for v in outs["none"].values()) > 50_000          # many 700-byte members per file
    if kind == "bz2":
        # ONE bzip2 stream per file: Python 2's BZ2File -- the reference's reader, moira/moira.py:1083-1084 --
        # stops after the first stream, so a one-shot decompressor must consume the whole file
        for f in sorted(os.listdir(tmp_path)):
            if f.startswith("o_bz2."):
                d = bz2.BZ2Decompressor()
                raw = open(tmp_path / f, "rb").read()
                text = d.decompress(raw)
                assert d.eof and d.unused_data == b"" and text == outs["none"][f.split(".", 1)[1].replace(".bz2", "")]


def test_collapse_takes_lines_longer_than_an_arena_block():
    """a header or sequence longer than the collapse store's 1 MiB block gets a block of its own (it used to
    fail as 'out of memory while collapsing')."""
    big = b"ACGT" * 300_000                                    # 1.2 MB sequence
    recs = [(b"long_read " + b"x" * 10, big), (b"r2", b"ACGTACGT"), (b"long_again", big), (b"r3", b"ACGTACGT")]
    buf = b"".join(b"@" + h + b"\n" + sq + b"\n+\n" + b"I" * len(sq) + b"\n" for h, sq in recs)
    idx, consumed, bad = F.index(buf, True, 16)
    assert len(idx) == 4 and bad is None
    col = F.Collapse(threads=2)
    col.add(buf, idx, np.array([1.0, 2.0, 0.5, 3.0]), np.zeros(4, np.uint8))
    assert len(col) == 2
    groups = col.export()
    col.close()
    assert groups is not None
