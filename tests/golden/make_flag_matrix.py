#!/usr/bin/env python3
"""Generate tests/golden/flag_matrix/ from the REAL reference: its output FILES for every non-default branch of
write_results / process_data.

Run in the build container only (needs /root/reference):
    python tests/golden/make_flag_matrix.py

What runs here
  * moira/moira.py's own parse_fastq -> process_data -> write_results (a lib2to3 copy in a temp dir outside the repo,
    imported, used, deleted -- as make_golden.py does), with the module's two optional accelerators bound to the
    reference's own code: `bernoulli` = oracle/_ref/bernoulli.so (moira/bernoullimodule.c, unmodified) and `nw` =
    moira/nw_align.pyx cythonized into the temp dir.  These are what moira.py imports when they are installed
    (moira/moira.py:241-257).
  * moira.py's main() itself does not run under Python 3 (open_input mixes bytes and str), so the loop
    around those functions -- which output files exist (moira/moira.py:296-376), the collapse rule (:459-475), the
    abundance sort (:491-504) -- is restated below, statement for statement.
  * Python 2 dicts iterate in slot order and moira's sort by abundance is stable, so groups of equal size come out in
    the slot order of a CPython-2.7 dict.  `Py27KeyOrder` below replays that table (Objects/dictobject.c and
    stringobject.c of CPython 2.7: 8 slots, 5 i + perturb + 1 probing, resize at 2/3 full to 4 x used).  It is written
    here independently of the product's moira_amd/py2dict.py, and THE GENERATOR CHECKS ITSELF before it writes anything:
    the two default-flag cases must reproduce the reference's own golden files (moira/test/test_results/*, made by a
    real Python 2) byte for byte.
Only data is written into the repo: manifest.json (cases, flags, file list, SHA-256, sizes) and outputs.tar.xz (the
files themselves, so that a failing test can show a diff).
"""
import hashlib
import io
import json
import os
import shutil
import sys
import tarfile
import types

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, HERE)
import golden_io as G  # noqa: E402
import make_golden as MG  # noqa: E402
import pb_oracle as O  # noqa: E402

REF = "/root/reference/moira"
OUT = os.path.join(HERE, "flag_matrix")
_M = (1 << 64) - 1


class Py27KeyOrder:
    """Insertion-only replay of a CPython-2.7 dict keyed by str: iteration order = slot order."""

    def __init__(self):
        self.size = 8
        self.table = [None] * 8
        self.n = 0

    @staticmethod
    def strhash(s):
        b = s.encode("latin-1")
        if not b:
            return 0
        x = (b[0] << 7) & _M
        for c in b:
            x = ((1000003 * x) ^ c) & _M
        x ^= len(b)
        return _M - 1 if x == _M else x

    def _put(self, h, key):
        mask = self.size - 1
        i = h & mask
        perturb = h
        while self.table[i & mask] is not None:
            i = (5 * i + perturb + 1) & _M
            perturb >>= 5
        self.table[i & mask] = (h, key)

    def add(self, key):
        self._put(self.strhash(key), key)
        self.n += 1
        if self.n * 3 >= self.size * 2:
            want = (2 if self.n > 50000 else 4) * self.n
            size = 8
            while size <= want:
                size <<= 1
            old, self.size, self.table = self.table, size, [None] * size
            for e in old:
                if e is not None:
                    self._put(*e)

    def keys(self):
        return [e[1] for e in self.table if e is not None]


class DefinedBernoulli:
    """moira/bernoullimodule.c, except where its result is undefined.  When the first row of the table already exceeds
    1 - alpha the C code interpolates with accumulated_probs[-1] (moira/bernoullimodule.c:254): stack garbage -- seen here as
    0.0 on one call and 0.176 on the next for the same read (contigs of summed qualities, `--qscore_cap 0`).  moira.py itself
    falls back to its Python twin when that garbage happens to be NaN (moira/moira.py:818-819); the twin defines the missing
    term as 0 (moira/moira.py:1611).  For exactly those reads -- first row = prod(1 - p_i) > 1 - alpha, evaluated with the C
    code's own expressions and order -- this shim returns the twin's value, as make_golden.py's run_set does.  `ub_reads`
    counts them."""

    def __init__(self, ref, pyref):
        self.ref, self.pyref, self.ub_reads, self.calls = ref, pyref, 0, 0

    def calculate_errors_PB(self, contig, quals, alpha):
        import math
        self.calls += 1
        row0, seen = 1.0, False
        for base, q in zip(contig, quals):
            if base in "Nn":                                        # moira/bernoullimodule.c:196
                continue
            a = math.pow(1 - math.pow(10, (q if q else 1) / -10.0), 1)   # :202, :104-107, :134-137
            row0 = row0 * a if seen else a                          # :223-229 with j = 0
            seen = True
        if seen and row0 > 1 - alpha:                               # the loop of :219-251 stops at j = 0
            self.ub_reads += 1
            ee, ns = self.pyref.calculate_errors_PB(contig.replace("n", "N"), [q if q else 1 for q in quals], alpha)
            return ee, ns
        return self.ref.calculate_errors_PB(contig, quals, alpha)


def reference_args(**kw):
    """moira/test/test_moira.py:130-135 (the reference's own test namespace), processors 1."""
    d = dict(alpha=0.005, match=1, gap=-2, mismatch=-1, insert=20, deltaq=6, consensus_qscore="best",
             paired=False, truncate=None, only_contig=False, error_calc="poisson_binomial",
             ambigs="treat_as_errors", round=False, silent=True, nowarnings=False, doc=False, uncert=0.01,
             maxerrors=None, processors=1, forward_fasta=None, forward_qual=None, reverse_fasta=None,
             reverse_qual=None, forward_fastq=None, reverse_fastq=None, output_format="fasta", collapse=True,
             pipeline="mothur", fastq_offset=33, relabel=None, output_compression="none", qscore_cap=40,
             min_overlap=None, trim_overlap=False, bootstrap=100, output_prefix=None)
    d.update(kw)
    return types.SimpleNamespace(**d)


# name, input ("shipped" = test1/test2.fastq as they are, "derived" = golden_io.derive_flag_inputs), flags
CASES = [
    ("se_shipped_default", "shipped", {}),
    ("pe_shipped_default", "shipped", {"paired": True}),
    # ---- single-end, derived input ----
    ("se_default", "derived", {}),
    ("se_truncate200", "derived", {"truncate": 200}),
    ("se_truncate200_fastq", "derived", {"truncate": 200, "output_format": "fastq"}),
    ("se_truncate150_nocollapse", "derived", {"truncate": 150, "collapse": False}),
    ("se_maxerrors1", "derived", {"maxerrors": 1.0}),
    ("se_maxerrors2p5_fastq", "derived", {"maxerrors": 2.5, "output_format": "fastq"}),
    ("se_maxerrors1_nocollapse", "derived", {"maxerrors": 1.0, "collapse": False}),
    ("se_disallow", "derived", {"ambigs": "disallow"}),
    ("se_disallow_fastq_nocollapse", "derived", {"ambigs": "disallow", "output_format": "fastq", "collapse": False}),
    ("se_ignore", "derived", {"ambigs": "ignore"}),
    ("se_round", "derived", {"round": True}),
    ("se_round_maxerrors2_usearch", "derived", {"round": True, "maxerrors": 2.0, "pipeline": "USEARCH"}),
    ("se_usearch", "derived", {"pipeline": "USEARCH"}),
    ("se_usearch_nocollapse_fastq", "derived", {"pipeline": "USEARCH", "collapse": False, "output_format": "fastq"}),
    ("se_relabel", "derived", {"relabel": "x"}),
    ("se_relabel_usearch_truncate", "derived", {"relabel": "Otu_", "pipeline": "USEARCH", "truncate": 220}),
    ("se_relabel_nocollapse", "derived", {"relabel": "r", "collapse": False}),
    ("se_fastq", "derived", {"output_format": "fastq"}),
    ("se_nocollapse", "derived", {"collapse": False}),
    ("se_alpha05_uncert02", "derived", {"alpha": 0.05, "uncert": 0.02}),
    ("se_alpha1e-4_uncert005", "derived", {"alpha": 1e-4, "uncert": 0.005}),
    ("se_poisson", "derived", {"error_calc": "poisson"}),
    ("se_poisson_ignore_round", "derived", {"error_calc": "poisson", "ambigs": "ignore", "round": True, "collapse": False}),
    ("se_min_overlap_without_paired", "derived", {"min_overlap": 30}),
    ("se_all_in_one", "derived", {"truncate": 180, "maxerrors": 1.5, "ambigs": "disallow", "round": True,
                                  "pipeline": "USEARCH", "relabel": "s", "output_format": "fastq"}),
    # ---- paired, derived input ----
    ("pe_default", "derived", {"paired": True}),
    ("pe_min_overlap100", "derived", {"paired": True, "min_overlap": 100}),
    ("pe_min_overlap100_fastq", "derived", {"paired": True, "min_overlap": 100, "output_format": "fastq"}),
    ("pe_min_overlap100_truncate300_fastq", "derived", {"paired": True, "min_overlap": 100, "truncate": 300,
                                                         "output_format": "fastq"}),
    ("pe_min_overlap150_nocollapse", "derived", {"paired": True, "min_overlap": 150, "collapse": False}),
    ("pe_only_contig", "derived", {"paired": True, "only_contig": True}),
    ("pe_only_contig_truncate", "derived", {"paired": True, "only_contig": True, "truncate": 300}),
    ("pe_only_contig_fastq_min_overlap", "derived", {"paired": True, "only_contig": True, "output_format": "fastq",
                                                      "min_overlap": 120}),
    ("pe_only_contig_nocollapse", "derived", {"paired": True, "only_contig": True, "collapse": False}),
    ("pe_only_contig_usearch", "derived", {"paired": True, "only_contig": True, "pipeline": "USEARCH"}),
    ("pe_trim_overlap", "derived", {"paired": True, "trim_overlap": True}),
    ("pe_sum", "derived", {"paired": True, "consensus_qscore": "sum"}),
    ("pe_sum_cap0_fastq", "derived", {"paired": True, "consensus_qscore": "sum", "qscore_cap": 0, "output_format": "fastq"}),
    ("pe_posterior", "derived", {"paired": True, "consensus_qscore": "posterior"}),
    ("pe_posterior_cap0_trim", "derived", {"paired": True, "consensus_qscore": "posterior", "qscore_cap": 0,
                                            "trim_overlap": True}),
    ("pe_truncate250", "derived", {"paired": True, "truncate": 250}),
    ("pe_disallow", "derived", {"paired": True, "ambigs": "disallow"}),
    ("pe_round_maxerrors1", "derived", {"paired": True, "round": True, "maxerrors": 1.0}),
    ("pe_usearch_nocollapse_fastq", "derived", {"paired": True, "pipeline": "USEARCH", "collapse": False,
                                                 "output_format": "fastq"}),
    ("pe_relabel_usearch", "derived", {"paired": True, "relabel": "c", "pipeline": "USEARCH"}),
    ("pe_scores_2_-3_-1", "derived", {"paired": True, "match": 2, "mismatch": -3, "gap": -1}),
    ("pe_insert30_deltaq3", "derived", {"paired": True, "insert": 30, "deltaq": 3}),
    ("pe_poisson", "derived", {"paired": True, "error_calc": "poisson"}),
    # ---- --error_calc poisson_binomial_py: the reference's Python twin (moira/moira.py:820-821), for which a lower-case
    #      'n' is a scored base (moira/moira.py:1605 counts only 'N'), unlike the C extension (bernoullimodule.c:196)
    ("se_pbpy_default", "derived", {"error_calc": "poisson_binomial_py"}),
    ("se_pbpy_nocollapse_ignore_fastq", "derived", {"error_calc": "poisson_binomial_py", "collapse": False, "ambigs": "ignore",
                                                     "output_format": "fastq"}),
    # ---- other encodings of the same reads: --fastq_offset 64, and a file full of things strip() forgives ----
    ("se_offset64_fastq_out", "derived_offset64", {"fastq_offset": 64, "output_format": "fastq"}),
    ("pe_offset64", "derived_offset64", {"fastq_offset": 64, "paired": True}),
    ("se_quirks_default", "derived_quirks", {}),
    ("pe_quirks_nocollapse", "derived_quirks", {"paired": True, "collapse": False}),
    ("pe_best_cap30", "derived", {"paired": True, "qscore_cap": 30}),
    # ---- the fasta + qual reader (moira/moira.py:1093-1149) on the derived reads, incl. scores of 120 and 300 ----
    ("fa_se_default", "derived_fasta_qual", {}),
    ("fa_se_nocollapse_usearch_maxerrors", "derived_fasta_qual", {"collapse": False, "pipeline": "USEARCH", "maxerrors": 1.0}),
    ("fa_se_truncate200_ignore", "derived_fasta_qual", {"truncate": 200, "ambigs": "ignore"}),
    ("fa_pe_default", "derived_fasta_qual", {"paired": True}),
    ("fa_pe_sum_cap0_min_overlap", "derived_fasta_qual", {"paired": True, "consensus_qscore": "sum", "qscore_cap": 0, "min_overlap": 100}),
]


class _Sink(io.StringIO):
    def close(self):            # keep the text readable after the driver "closes" it
        pass


def open_outputs(args):
    """Which files moira opens, by name after the prefix (moira/moira.py:296-376)."""
    files = {}

    def mk(stem):
        files[stem] = _Sink()
        return files[stem]
    o = types.SimpleNamespace(contig=None, qual=None, names=None, bad_contig=None, bad_qual=None, bad_names=None, report=None)
    fq = args.output_format == "fastq"
    if args.only_contig:
        if fq:
            o.contig = mk("contigs.fastq")
        else:
            o.contig, o.qual = mk("contigs.fasta"), mk("contigs.qual")
        if args.collapse and args.pipeline == "mothur":
            o.names = mk("contigs.names")
        if args.truncate or args.min_overlap:
            if fq:
                o.bad_contig = mk("bad.contigs.fastq")
            else:
                o.bad_contig, o.bad_qual = mk("bad.contigs.fasta"), mk("bad.contigs.qual")
            if args.collapse and args.pipeline == "mothur":
                o.bad_names = mk("bad.contigs.names")
    else:
        if fq:
            o.contig, o.bad_contig = mk("qc.good.fastq"), mk("qc.bad.fastq")
        else:
            o.contig, o.qual = mk("qc.good.fasta"), mk("qc.good.qual")
            o.bad_contig, o.bad_qual = mk("qc.bad.fasta"), mk("qc.bad.qual")
        if args.collapse and args.pipeline == "mothur":
            o.names, o.bad_names = mk("qc.good.names"), mk("qc.bad.names")
    if args.paired:
        o.report = mk("contigs.report")
        o.report.write("header\tn_seqs\toverlap_length\tgaps\tmismatches\n")
    return o, files


def drive(M, args, fwd_path, rev_path):
    """The body of moira.py's main() around the reference's own functions (moira/moira.py:400-504).  fwd_path / rev_path: a
    FASTQ file each, or a (fasta, qual) pair each (then the reference's parse_fasta_and_qual reads them)."""
    if args.only_contig:                          # check_arguments, moira/moira.py:698-699
        args.paired = True
    o, files = open_outputs(args)
    fasta_qual = isinstance(fwd_path, tuple)
    if fasta_qual:
        handles = [open(fwd_path[0]), open(fwd_path[1])] + ([open(rev_path[0]), open(rev_path[1])] if args.paired else [None, None])
        records = M.parse_fasta_and_qual(*handles)
        fh, rh = None, None
    else:
        fh = open(fwd_path, newline="")           # no newline translation: the reference iterates the raw lines and strip()s them
        rh = open(rev_path, newline="") if args.paired else None
        records = M.parse_fastq(fh, rh, args.fastq_offset)
    uniques, order = {}, Py27KeyOrder()
    totals = [0.0, 0.0, 0.0]
    processed = 0

    def write(index, header, seq, quals, ee, names_info, ov, gaps, mism):
        r = M.write_results(index, header, seq, quals, ee, names_info, ov, gaps, mism, args,
                            o.contig, o.qual, o.names, o.bad_contig, o.bad_qual, o.bad_names, o.report)
        for k in range(3):
            totals[k] += r[k]
    for header, fs, fq_, rs, rq in records:
        header, contig, cq, ee, ov, gaps, mism = M.process_data(header, fs, fq_, rs, rq, args)
        assert ee == ee, "NaN from the reference for " + header
        if args.collapse:                         # moira/moira.py:459-475
            if contig not in uniques:
                uniques[contig] = {"rep_header": header, "rep_errors": ee, "rep_quals": cq, "names_info": [header],
                                   "overlap_length": ov, "gaps": gaps, "mismatches": mism}
                order.add(contig)
            else:
                u = uniques[contig]
                if ee < u["rep_errors"]:
                    u["rep_header"], u["rep_errors"], u["rep_quals"] = header, ee, cq
                    u["names_info"].insert(0, header)
                    u["overlap_length"], u["gaps"], u["mismatches"] = ov, gaps, mism
                else:
                    u["names_info"].append(header)
        else:                                     # moira/moira.py:478-485
            write(processed, header, contig, cq, ee, None, ov, gaps, mism)
        processed += 1
    if args.collapse:                             # moira/moira.py:491-504 (stable sort over Python 2's dict order)
        keys = order.keys()
        assert len(keys) == len(uniques)
        for index, seq in enumerate(sorted(keys, key=lambda s: len(uniques[s]["names_info"]), reverse=True), start=1):
            v = uniques[seq]
            write(index, v["rep_header"], seq, v["rep_quals"], v["rep_errors"], v["names_info"],
                  v["overlap_length"], v["gaps"], v["mismatches"])
    for f in ([fh, rh] if not fasta_qual else handles):
        if f:
            f.close()
    return {k: f.getvalue().encode("latin-1") for k, f in files.items()}, processed, totals


def main():
    O.build()
    ref = O.reference_module()
    assert ref is not None, "oracle/_ref/bernoulli.so missing: make -C oracle ref"
    M, tmp = MG.load_python_reference()
    try:
        M.bernoulli, M.Cbernoulli = DefinedBernoulli(ref, M), True  # what `import bernoulli` binds (moira/moira.py:247-251)
        M.nw, M.Cy_nw_align = MG.load_cython_nw(tmp), True         # `import nw_align as nw` (moira/moira.py:241-245)
        shipped = (os.path.join(REF, "test", "test1.fastq"), os.path.join(REF, "test", "test2.fastq"))
        d1, d2 = G.derive_flag_inputs(G.read_fastq_records(shipped[0]), G.read_fastq_records(shipped[1]))
        derived = (os.path.join(tmp, "derived1.fastq"), os.path.join(tmp, "derived2.fastq"))
        G.write_fastq(derived[0], d1)
        G.write_fastq(derived[1], d2)
        fq_paths = ((os.path.join(tmp, "derived1.fasta"), os.path.join(tmp, "derived1.qual")),
                    (os.path.join(tmp, "derived2.fasta"), os.path.join(tmp, "derived2.qual")))
        G.write_fasta_qual(fq_paths[0][0], fq_paths[0][1], d1)
        G.write_fasta_qual(fq_paths[1][0], fq_paths[1][1], d2)
        off64 = (os.path.join(tmp, "off64_1.fastq"), os.path.join(tmp, "off64_2.fastq"))
        quirks = (os.path.join(tmp, "quirks1.fastq"), os.path.join(tmp, "quirks2.fastq"))
        for k, recs in enumerate((d1, d2)):
            G.write_fastq_offset64(off64[k], recs)
            G.write_fastq_quirks(quirks[k], recs)
        sha = lambda b: hashlib.sha256(b).hexdigest()
        manifest = {"source": "moira/moira.py parse_fastq -> process_data -> write_results (lib2to3 copy, bernoulli = "
                              "moira/bernoullimodule.c unmodified, nw = moira/nw_align.pyx), driven by "
                              "tests/golden/make_flag_matrix.py; Python-2 dict order replayed and self-checked against "
                              "moira/test/test_results/*",
                    "inputs": {"shipped": ["test1.fastq", "test2.fastq"],
                               "derived": {"by": "tests/golden_io.py:derive_flag_inputs",
                                           "sha256": [sha(open(p, "rb").read()) for p in derived],
                                           "records": len(d1)},
                               "derived_offset64": {"by": "tests/golden_io.py:write_fastq_offset64",
                                                    "sha256": [sha(open(p, "rb").read()) for p in off64]},
                               "derived_quirks": {"by": "tests/golden_io.py:write_fastq_quirks",
                                                  "sha256": [sha(open(p, "rb").read()) for p in quirks]},
                               "derived_fasta_qual": {"by": "tests/golden_io.py:write_fasta_qual on the derived records",
                                                      "sha256": [sha(open(p, "rb").read()) for pair in fq_paths for p in pair]}},
                    "cases": {}}
        results = {}
        for name, which, flags in CASES:
            args = reference_args(**flags)
            ub0 = M.bernoulli.ub_reads
            f, r = {"shipped": shipped, "derived": derived, "derived_fasta_qual": fq_paths, "derived_offset64": off64,
                    "derived_quirks": quirks}[which]
            files, processed, totals = drive(M, args, f, r)
            results[name] = files
            manifest["cases"][name] = {"input": which, "flags": flags, "processed": processed, "reads_scored_by_the_python_twin": M.bernoulli.ub_reads - ub0,
                                       "discarded_errors_minlength_minoverlap": totals,
                                       "files": {k: {"sha256": sha(v), "bytes": len(v)} for k, v in sorted(files.items())}}
            print("%-40s %5d reads (ub %3d)  discarded %s  %s" % (name, processed, M.bernoulli.ub_reads - ub0, [int(t) for t in totals],
                                                        " ".join("%s:%d" % (k, len(v)) for k, v in sorted(files.items()))))
        # ---- self-check: the two default-flag runs on the shipped files ARE the reference's golden files ----
        for case, gold in (("se_shipped_default", "forward"), ("pe_shipped_default", "paired")):
            for kind in ("good.fasta", "good.qual", "good.names", "bad.fasta", "bad.qual", "bad.names"):
                want = open(os.path.join(REF, "test", "test_results", "%s.qc.%s" % (gold, kind)), "rb").read()
                assert results[case]["qc." + kind] == want, (case, kind)
        print("self-check: default-flag runs reproduce moira/test/test_results/* byte for byte (Python-2 dict order included)")
        # every label / branch the matrix is meant to pin must actually occur
        blob = b"".join(v for fs in results.values() for v in fs.values())
        for needle in (b"\tlength below 200\n", b"\toverlap length below 100\n", b"\toverlap length below None\n",
                       b"\toverlap length below 300\n", b"\tcontains ambiguities\n", b"\terrors > 1.00\n",
                       b"\tuncert > 0.010\n", b"\tuncert > 0.020\n", b";ee=", b";size=2;", b">x1\n", b"@s1;ee="):
            assert needle in blob, needle
        os.makedirs(OUT, exist_ok=True)
        json.dump(manifest, open(os.path.join(OUT, "manifest.json"), "w"), indent=1, sort_keys=True)
        tar_path = os.path.join(OUT, "outputs.tar.xz")
        import lzma
        with lzma.open(tar_path, "wb", preset=9 | lzma.PRESET_EXTREME) as xz, tarfile.open(fileobj=xz, mode="w") as tf:
            # files of one kind next to each other: the cases share most of their text
            for stem in sorted({k for fs in results.values() for k in fs}):
                for name, _, _ in CASES:
                    if stem in results[name]:
                        ti = tarfile.TarInfo("%s/%s" % (name, stem))
                        ti.size = len(results[name][stem])
                        tf.addfile(ti, io.BytesIO(results[name][stem]))
        print("%d cases, %d files, %.1f MB of text -> %s (%.2f MB)" % (
            len(CASES), sum(len(f) for f in results.values()), len(blob) / 1e6, os.path.relpath(tar_path, ROOT),
            os.path.getsize(tar_path) / 1e6))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
