#!/usr/bin/env python3
"""moira-compatible command line on top of the MI355X filter.

    python -m moira_amd.cli --forward_fastq reads.fastq [--paired --reverse_fastq mates.fastq] ...

Keeps moira.py's flags, defaults, validation messages, input handling and output files
(moira/moira.py:581-675 parse_arguments, :678-781 check_arguments, :1058-1204 readers,
:842-970 write_results, :264-578 main), so it is a drop-in for the script.  What changes is the
shape of the hot loop: instead of one `process_data` task per read through a multiprocessing
Pool (moira/moira.py:431-454), records are read in chunks, contigs for paired reads are built on
the CPU by libmoira_contig.so (all host cores), the chunk is packed into per-length-bucket
quality matrices and filtered on the GPU by libmoira_pb.so, and the per-read results flow into the
same collapse / write logic.

`--error_calc poisson_binomial` and `poisson_binomial_py` both run the HIP path (they are the same
arithmetic; the reference's two names select its C or Python implementation -- which differ in one thing that is
kept: the C extension counts a lower-case 'n' as ambiguous, the Python twin scores it as a base).  `poisson` sums the
per-read lambda on the GPU and finishes the scalar CDF on the host with the reference's libm calls
(moira/moira.py:1637-1679).  `bootstrap` (deprecated, random) is evaluated on the host as moira.py
does (moira/moira.py:1682-1720).  `--processors` sets the CPU threads used for contig construction.

Output order: with --collapse moira writes groups sorted by abundance and, inside one abundance,
in Python-2 dict order.  That order is reproduced (moira_amd/py2dict.py) so the output files are
byte-identical to the reference's.
"""
import argparse
import bz2
import gzip
import io
import math
import os
import struct
import sys
import time
import zlib

import numpy as np

from .buckets import QualStr
from .py2dict import Py2Dict

__version__ = "1.3.2-mi355x"
CHUNK_READS = 262144


# ---------------------------------------------------------------------------------------------
# exceptions (names and messages of moira/moira.py:973-1055)
# ---------------------------------------------------------------------------------------------
class UnsupportedReadError(Exception):
    """A read this build cannot score the way the reference would (never scored differently instead): raised BEFORE
    the chunk is filtered; main() removes the partial output files, prints the message and returns 1."""


class QualityTooHighError(UnsupportedReadError):
    """Not a reference exception: the packed quality matrix holds one byte per base (0 = 'N', 255 = 'n'), so Phred
    scores above 254 cannot be encoded.  FASTQ cannot produce them (the largest character minus the smallest offset is
    222) and neither can contig construction from FASTQ input (`--consensus_qscore sum` without a cap: 186); only a
    .qual file with such integers can (the reference takes any int there, moira/bernoullimodule.c:92-108).  Such a read
    is scored exactly through the per-read entries (process_chunk); this error is what a backend without them raises."""
    def __init__(self, header, quality):
        self.header, self.quality = header, quality

    def __str__(self):
        return ("Sequence %s has a quality score of %d; this backend encodes scores up to 254 (an error probability of "
                "4e-26). Cap the scores in the .qual file (e.g. at 93, the FASTQ maximum) and run again: a capped score "
                "changes an error probability that is below 1e-25 either way." % (self.header, self.quality))


class ReadTooLongError(UnsupportedReadError):
    """Not a reference exception: the Poisson-binomial kernels cover reads of up to 65535 bases (the reference's
    own C path overruns its stack near 1000; its Python twin has no limit but needs O(J^2 L) Python
    steps).  Raised BEFORE the chunk is filtered; main() removes the partial output files and explains the
    alternatives."""
    def __init__(self, header, length):
        self.header, self.length = header, length

    def __str__(self):
        return ("Sequence %s has %d bases after contig construction / truncation; the poisson_binomial error "
                "calculation of this build supports at most %d. Use --error_calc poisson (recommended by moira "
                "for reads > 500 nt) or --truncate." % (self.header, self.length, MAX_PB_LEN))


MAX_PB_LEN = 65535       # MPB_MAX_LEN of the HIP library (a read may NEED at most 16384 DP rows: ~16,000 expected errors)


class ReturnedNaNError(Exception):
    def __init__(self, header, length=None):
        self.header, self.length = header, length

    def __str__(self):
        msg = ("Error calculation returned NaN for sequence %s. If using a C implementation, "
               "try switching to the python one instead." % self.header)          # the reference's words (moira/moira.py:973-979)
        if self.length is not None and self.length > 16383:
            # the one way this build produces a NaN under the Poisson-binomial methods: more rows than 16 waves hold
            msg += (" (This build: the read has %d bases and needs more than 16384 rows of the error table, i.e. about 16,000 "
                    "expected errors; --error_calc poisson has no such limit.)" % self.length)
        return msg


class UnpairedFilesError(Exception):
    def __init__(self, lf, lq, rf=None, rq=None):
        self.rf, self.rq = rf, rq

    def __str__(self):
        if not self.rf and not self.rq:
            return "You must provide at least a forward fasta and quality file"
        return ("If reading from paired-end files, you must provide fasta and quality files for both "
                "the forward and the reverse reads")


class NameMismatchError(Exception):
    def __init__(self, lfheader, lqheader, rfheader=None, rqheader=None):
        self.h = (lfheader, lqheader, rfheader, rqheader)

    def __str__(self):
        lf, lq, rf, rq = self.h
        if not rf:
            return "Fasta header does not match Qfile header. Offending headers were: FASTA: %s   QFILE: %s" % (repr(lf), repr(lq))
        if not lq:
            return "Header mismatch. Offending headers were: forward_fastq: %s,   reverse_fastq %s" % (repr(lf), repr(rf))
        return ("Header mismatch. Offending headers were: forward_fasta: %s   forward_qual %s   "
                "reverse_fasta %s   reverse_qual %s" % (repr(lf), repr(lq), repr(rf), repr(rq)))


class LengthMismatchError(Exception):
    def __init__(self, fheader=None, ffilename=None, qfilename=None):
        self.a = (fheader, ffilename, qfilename)

    def __str__(self):
        h, f, q = self.a
        if None not in (h, f, q):
            return ("Error reading sequence %s in files %s and %s. Sequence length and quality length "
                    "do not match" % (repr(h), repr(f), repr(q)))
        if not q:
            return "Error reading sequence %s in file %s. Sequence length and quality length do not match" % (repr(h), repr(f))
        return "Sequence and qualities are of different lengths."


class EmptySeqError(Exception):
    def __init__(self, fheader, filename):
        self.a = (fheader, filename)

    def __str__(self):
        return "Error reading file %s. Sequence %s was empty" % (repr(self.a[1]), repr(self.a[0]))


class EmptyQualError(Exception):
    def __init__(self, qheader, filename):
        self.a = (qheader, filename)

    def __str__(self):
        return "Error reading file %s. Quality %s was empty" % (repr(self.a[1]), repr(self.a[0]))


# ---------------------------------------------------------------------------------------------
# arguments
# ---------------------------------------------------------------------------------------------
def build_parser():
    def str2bool(value):
        return value.lower() in ("yes", "true", "t", "1")

    p = argparse.ArgumentParser(
        description="Perform quality filtering on a set of sequences.",
        epilog="Limits of this build (a run that meets one stops with a message and leaves no partial output): the "
               "poisson_binomial methods score reads of up to 65535 bases that need at most 16384 rows of the error table, i.e. about "
               "16,000 expected errors (--error_calc poisson: any length).")
    g = p.add_argument_group("General options")
    g.add_argument("-ff", "--forward_fasta", type=str, help="Forward fasta file (can be gzip or bzip2 compressed).")
    g.add_argument("-fq", "--forward_qual", type=str, help="Forward qual file (can be gzip or bzip2 compressed).")
    g.add_argument("-rf", "--reverse_fasta", type=str, help="Reverse fasta file (can be gzip or bzip2 compressed).")
    g.add_argument("-rq", "--reverse_qual", type=str, help="Reverse qual file (can be gzip or bzip2 compressed).")
    g.add_argument("-ffq", "--forward_fastq", type=str, help="Forward fastq file (can be gzip or bzip2 compressed).")
    g.add_argument("-rfq", "--reverse_fastq", type=str, help="Reverse fastq file (can be gzip or bzip2 compressed).")
    g.add_argument("-l", "--relabel", type=str,
                   help="Generate sequential labels for the ordered sequences, with the specified string at the beginning.")
    g.add_argument("-o", "--output_format", type=str, default="fasta", choices=("fasta", "fastq"),
                   help="Output format: fasta with qual (and mothur name file if --collapse) or fastq.")
    g.add_argument("-pi", "--pipeline", type=str, default="mothur", choices=("mothur", "USEARCH"),
                   help="Make the output format compatible with the indicated analysis pipeline.")
    g.add_argument("-op", "--output_prefix", type=str, help="Prefix for the output files")
    g.add_argument("-oc", "--output_compression", type=str, default="none", choices=("none", "gz", "bz2"),
                   help="Compression of the output files")
    g.add_argument("-p", "--processors", type=int, default=1,
                   help="Number of CPU threads for contig construction (the filter itself runs on the GPU).")
    g.add_argument("--paired", action="store_true",
                   help="Assemble paired-end reads and perform quality control on the resulting contig.")
    g.add_argument("-fo", "--fastq_offset", type=int, default=33)
    g.add_argument("--only_contig", action="store_true", help="Assemble contigs but don't perform quality control.")
    g.add_argument("--silent", action="store_true",
                   help="Do not print welcome, progress and goodbye messages. Warnings will still be printed.")
    g.add_argument("--nowarnings", action="store_true", help="Do not print warning messages.")
    g.add_argument("--doc", action="store_true", help="Print full documentation.")

    c = p.add_argument_group("Contig construction options")
    c.add_argument("-m", "--match", type=int, default=1, help="Needleman-Wunsch aligner match score.")
    c.add_argument("-x", "--mismatch", type=int, default=-1, help="Needleman-Wunsch aligner mismatch penalty.")
    c.add_argument("-g", "--gap", type=int, default=-2, help="Needleman-Wunsch aligner gap penalty.")
    c.add_argument("--trim_overlap", action="store_true", help="Trim the contig to the overlapping region.")
    c.add_argument("-i", "--insert", type=int, default=20, help="Contig constructor insert threshold.")
    c.add_argument("-d", "--deltaq", type=int, default=6, help="Contig constructor mismatch correction deltaq threshold.")
    c.add_argument("-q", "--consensus_qscore", type=str, default="best", choices=("best", "sum", "posterior"),
                   help="Contig constructor consensus qscore.")
    c.add_argument("-z", "--qscore_cap", type=int, default=40,
                   help="Maximum consensus quality score reported by the contig constructor. Use 0 for no cap.")

    f = p.add_argument_group("Sequence filtering options")
    f.add_argument("-c", "--collapse", type=str2bool, default="True",
                   help="Collapse identical sequences before quality control.")
    f.add_argument("-t", "--truncate", type=int,
                   help="Truncate sequences to a fixed length before quality control. Discard smaller sequences.")
    f.add_argument("-mo", "--min_overlap", type=int, help="Discard contigs with less than the specified overlap length.")
    f.add_argument("-e", "--error_calc", type=str, default="poisson_binomial",
                   choices=("poisson_binomial", "poisson_binomial_py", "poisson", "bootstrap"),
                   help="Error calculation method.")
    f.add_argument("-n", "--ambigs", type=str, default="treat_as_errors",
                   choices=("disallow", "ignore", "treat_as_errors"), help="Treatment of ambiguities.")
    f.add_argument("-r", "--round", action="store_true",
                   help="Round down the predicted number of errors to their nearest integer prior to filtering.")
    eu = f.add_mutually_exclusive_group()
    eu.add_argument("-u", "--uncert", type=float, default=0.01,
                    help="Maximum allowed uncertainty (errors / sequence length).")
    eu.add_argument("-me", "--maxerrors", type=float, help="Maximum allowed errors per sequence.")
    f.add_argument("-a", "--alpha", type=float, default=0.005, help="Alpha cutoff value for the error distributions.")
    f.add_argument("-b", "--bootstrap", type=int, default=100, help="Number of replicates to use with the bootstrap method")
    f.add_argument("--device", type=str, default=None,
                   help="(not in moira.py) GPU index (default: LOCAL_RANK or 0), a comma-separated list of indices, or "
                        "'all': with more than one GPU every chunk is split over them in read order from this one "
                        "process (one host thread and one PCIe link per GPU) -- the role --processors had for "
                        "moira.py's per-read pool.")
    f.add_argument("--fast_discard", action="store_true",
                   help="(not in moira.py) skip the exact error calculation for reads that provably exceed the "
                        "threshold; only with --collapse false and the mothur pipeline, where the expected errors "
                        "of a discarded read are never used. Kept/discarded sets are unchanged.")
    return p


def parse_arguments(argv=None):
    args = build_parser().parse_args(argv)
    if isinstance(args.collapse, str):            # argparse does not pass a str default through `type`... it does; be safe
        args.collapse = args.collapse.lower() in ("yes", "true", "t", "1")
    return args


def check_arguments(args, out=None):
    """ref: moira/moira.py:678-781 (same checks, same messages)."""
    def say(msg=""):
        print(msg, file=out or sys.stdout)

    say()
    if args.doc:
        say(__doc__)
        return False
    ok = True
    warn = not args.nowarnings
    if args.only_contig:
        args.paired = True
    else:
        if not args.forward_fastq and (not args.forward_fasta or not args.forward_qual):
            if warn:
                say("- You must at least provide one fastq file, or a fasta and quality files.")
            ok = False
        if args.paired:
            if not args.reverse_fastq and (not args.reverse_fasta or not args.reverse_qual):
                if warn:
                    say("- You must provide one reverse fastq file, or reverse fasta and quality files.")
                ok = False
    checks = (
        (args.match < 0, "- Needleman-Wunsch match score must be a non-negative integer."),
        (args.mismatch > 0, "- Needleman-Wunsch mismatch penalty must be a non-positive integer."),
        (args.gap > 0, "- Needleman-Wunsch gap penalty must be a non-positive integer."),
        (args.insert < 1, "- The contig constructor insert parameter must be a positive integer."),
        (args.deltaq < 1, "- The contig constructor deltaq parameter must be a positive integer."),
        (not 0 < args.uncert <= 1, "- The uncert parameter must be between 0 (not included) and 1."),
        (args.maxerrors is not None and args.maxerrors <= 0, "- The maxerrors parameter must be greater than 0."),
        (not 0 < args.alpha < 1, "- The alpha parameter must be between 0 (not included) and 1."),
        (bool(args.truncate) and args.truncate <= 0, "- The truncate parameter must be greater than 0."),
        (args.min_overlap is not None and args.min_overlap <= 0, "- The min_overlap parameter must be greater than 0."),
    )
    for bad, msg in checks:
        if bad:
            if warn:
                say(msg)
            ok = False
    if not ok:
        if warn:
            say("\nFor more info type moira.py -h or moira.py --doc.\n")
        return False
    if args.processors < 1:
        if warn:
            say("- Processors must be a non-zero positive integer. The default value of 1 will be used.")
        args.processors = 1
    if (args.reverse_fasta or args.reverse_fastq) and not args.paired and warn:
        say("You provided a reverse sequence file, but not the --paired flag. Note that only the forward file will be processed.")
        say()
    if args.min_overlap and not args.paired and warn:
        say("You specified a value for --min_overlap, but not the --paired flag. Note that contigs will not be assembled.")
        say()
    if args.error_calc == "bootstrap" and warn:
        say("The bootstrap method is only included for testing and nostalgia. Mainly the second, at this point.")
        say('If your purpose falls outside of these two categories, please consider switching to "-e poisson_binomial" or "-e poisson".')
        say()
    return True


# ---------------------------------------------------------------------------------------------
# input (ref: moira/moira.py:1058-1204)
# ---------------------------------------------------------------------------------------------
def open_input(filename):
    """Sniff gzip / bzip2 by magic bytes; text mode, one line at a time."""
    with io.open(filename, "rb") as fh:
        start = fh.read(3)
    if start.startswith(b"\x42\x5a\x68"):
        f = bz2.open(filename, "rt", newline=None)
    elif start.startswith(b"\x1f\x8b\x08"):
        f = gzip.open(filename, "rt", newline=None)
    else:
        f = io.open(filename, "rt", buffering=1 << 20)
    f.moira_name = filename
    return f


class _ReadAhead:
    """A compressed input read on a thread of its own: decompression (zlib / bz2 release the GIL) of the next blocks runs
    while the current one is indexed, packed and filtered -- and, for paired input, the two files inflate side by side
    instead of one after the other.  read(n) hands out the blocks as they come (at most n bytes, b"" at the end)."""

    def __init__(self, fh, block=1 << 25, depth=3):
        import queue
        import threading
        self.fh, self.block = fh, block
        self.q = queue.Queue(maxsize=depth)
        self.left = b""
        self.done = False
        self.stop = threading.Event()
        self.t = threading.Thread(target=self._work, daemon=True)
        self.t.start()

    def _work(self):
        import queue
        item = None
        try:
            while not self.stop.is_set():
                if item is None:
                    data = self.fh.read(self.block)
                    item = (data, None)
                try:
                    self.q.put(item, timeout=0.1)
                except queue.Full:
                    continue
                if not item[0]:
                    return
                item = None
        except BaseException as e:          # noqa: B902 -- re-raised by read()
            while not self.stop.is_set():
                try:
                    self.q.put((b"", e), timeout=0.1)
                    return
                except queue.Full:
                    continue

    def read(self, n=-1):
        if not self.left and not self.done:
            data, err = self.q.get()
            if err is not None:
                self.done = True
                raise err
            if not data:
                self.done = True
            self.left = data
        if n is None or n < 0 or n >= len(self.left):
            out, self.left = self.left, b""
        else:
            out, self.left = self.left[:n], self.left[n:]
        return out

    def close(self):
        self.stop.set()
        self.t.join()
        self.fh.close()


def open_input_binary(filename, threads=1):
    """Same sniffing, bytes out: the C parser (moira_amd/fastio.py) takes whole blocks.  `threads`: what a BGZF-blocked
    gzip file (bgzip, Illumina's writers, this package's own .gz outputs) is inflated on; any other gzip file is one
    stream on one thread."""
    with io.open(filename, "rb") as fh:
        start = fh.read(3)
    if start.startswith(b"\x42\x5a\x68"):
        return _ReadAhead(bz2.open(filename, "rb"))
    if start.startswith(b"\x1f\x8b\x08"):
        # libmoira_io's own inflate (csrc/inflate.cpp): about twice zlib's text rate, on a thread of its own
        if os.environ.get("MOIRA_ZLIB_INPUT"):
            return _ReadAhead(gzip.open(filename, "rb"))
        from . import fastio as F
        return _ReadAhead(F.GzipReader(io.open(filename, "rb", buffering=0), threads=threads))
    return io.open(filename, "rb", buffering=0)


def _norm(header, mark):
    # strip, tabs -> spaces, first token, drop the leading mark(s), ':' -> '_'   (moira.py:1121,1175)
    return header.strip().replace("\t", " ").split(" ")[0].lstrip(mark).replace(":", "_")


def parse_fastq(fwd, rev=None, fastq_offset=33, raw=False):
    """raw=False: the reference's generator contract (int lists).  raw=True: qualities stay
    FASTQ strings (QualStr) so the chunk can be packed by one C loop instead of per-base Python."""
    fb, rb = [], []
    it = zip(fwd, rev) if rev is not None else ((l, None) for l in fwd)
    for fl, rl in it:
        fb.append(fl.strip())
        if rev is not None:
            rb.append(rl.strip())
        if len(fb) == 4:
            fh = _norm(fb[0], "@")
            fs = fb[1]
            fq = QualStr(fb[3], fastq_offset) if raw else [ord(x) - fastq_offset for x in fb[3]]
            if not fs:
                raise EmptySeqError(fh, fwd.moira_name)
            if not fq:
                raise EmptyQualError(fh, fwd.moira_name)
            if len(fs) != len(fq):
                raise LengthMismatchError(fh, fwd.moira_name)
            fb = []
            if rev is not None:
                rh = _norm(rb[0], "@")
                rs = rb[1]
                rq = QualStr(rb[3], fastq_offset) if raw else [ord(x) - fastq_offset for x in rb[3]]
                if not rs:
                    raise EmptySeqError(fh, fwd.moira_name)
                if not rq:
                    raise EmptyQualError(fh, fwd.moira_name)
                if len(rs) != len(rq):
                    raise LengthMismatchError(rh, rev.moira_name)
                if fh != rh:
                    raise NameMismatchError(fh, None, rh, None)
                rb = []
                yield fh, fs, fq, rs, rq
            else:
                yield fh, fs, fq, None, None


def parse_fasta_and_qual(ff, fq, rf=None, rq=None):
    if not ff or not fq:
        raise UnpairedFilesError(ff, fq, rf, rq)
    if (not rf and rq) or (rf and not rq):
        raise UnpairedFilesError(ff, fq, rf, rq)

    def quals_of(line):
        toks = line.strip().replace("\t", " ").split(" ")
        return [int(t) for t in toks] if toks != [""] else []

    while True:
        fh, qh = ff.readline(), fq.readline()
        rh = rf.readline() if rf else ""
        rqh = rq.readline() if rq else ""
        if rf and rq:
            if not fh and not rh and not qh and not rqh:
                break
        elif not fh and not qh:
            break
        fh, qh = _norm(fh, ">"), _norm(qh, ">")
        fseq = ff.readline().strip()
        fquals = quals_of(fq.readline())
        if rf:
            rh, rseq = _norm(rh, ">"), rf.readline().strip()
            rqh, rquals = _norm(rqh, ">"), quals_of(rq.readline())
            if len({fh, rh, qh, rqh}) != 1:
                raise NameMismatchError(fh, qh, rh, rqh)
        elif len({fh, qh}) != 1:
            raise NameMismatchError(fh, qh)
        if not fseq:
            raise EmptySeqError(fh, ff.moira_name)
        if not fquals:
            raise EmptyQualError(qh, fq.moira_name)
        if rf and not rseq:
            raise EmptySeqError(rh, rf.moira_name)
        if rq and not rquals:
            raise EmptyQualError(rqh, rq.moira_name)
        if len(fseq) != len(fquals):
            raise LengthMismatchError(fh, ff.moira_name, fq.moira_name)
        if rf and len(rseq) != len(rquals):
            raise LengthMismatchError(rh, ff.moira_name, fq.moira_name)
        yield (fh, fseq, fquals, rseq, rquals) if rf else (fh, fseq, fquals, None, None)


# ---------------------------------------------------------------------------------------------
# the bootstrap calculator stays a host loop: deprecated in the reference and random by construction
# (ref: moira/moira.py:1682-1733)
# ---------------------------------------------------------------------------------------------
def interpolate(e1, p1, e2, p2, alpha):
    r = e1 + ((e2 - e1) * ((1 - alpha) - p1) / (p2 - p1))
    return 0 if r < 0 else r


def calculate_errors_bootstrap(sequence, quals, alpha, bootstrap):
    from numpy.random import random
    results, ns = [], 0
    for _ in range(int(bootstrap)):
        errors, ns = 0, 0
        for base, q in zip(sequence, quals):
            if base == "N":
                ns += 1
            elif random() <= 10 ** (q / -10.0):
                errors += 1
        results.append(errors)
    return float(np.percentile(results, (1 - alpha) * 100)), ns


# ---------------------------------------------------------------------------------------------
# the filter half of process_data for a chunk (ref: moira/moira.py:784-833)
# ---------------------------------------------------------------------------------------------
def parse_devices(device):
    """--device value -> [int, ...] or ["all"] (None: LOCAL_RANK or 0)."""
    if device is None:
        return [int(os.environ.get("LOCAL_RANK", "0"))]
    if isinstance(device, int):
        return [device]
    text = str(device).strip()
    if text.lower() == "all":
        return ["all"]
    try:
        devs = [int(x) for x in text.split(",") if x.strip() != ""]
    except ValueError:
        devs = []
    if not devs or min(devs) < 0:
        raise ValueError("--device wants a GPU index, a comma-separated list of indices, or 'all' (got %r)" % (device,))
    return devs


def make_gpu_backend(device=None):
    """Default (and only product) backend: the HIP library.  Fails loudly without a GPU."""
    from .buckets import filter_bucketed
    from .engine import Engine
    devs = parse_devices(device)
    if len(devs) == 1:
        eng = Engine(devs[0])
    else:
        from .shard import MultiEngine
        eng = MultiEngine(None if devs == ["all"] else devs)

    def backend(seqs, quals, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        if fast_discard is not None and method == "poisson_binomial":
            uncert, maxerrors = fast_discard
            ee, ns, _ = filter_bucketed(eng, seqs, quals, alpha=alpha, ambigs=ambigs, round_=round_,
                                        uncert=uncert, maxerrors=maxerrors, decision_only=True)
            return ee      # reads settled by the bound carry ee = +inf ("certainly above the threshold"); NaN stays an error
        if method == "poisson":
            # the Python reference scores a lower-case n as a normal base (moira.py:1660)
            seqs = [s.replace("n", "A") if "n" in s else s for s in seqs]
        ee, ns, _ = filter_bucketed(eng, seqs, quals, method=method, alpha=alpha, ambigs=ambigs,
                                    round_=round_, uncert=1.0)
        return ee
    def matrix(q, lens, alpha, ambigs, round_, method="poisson_binomial", fast_discard=None):
        """The same for reads that are already packed (the byte-level FASTQ path): ee per read."""
        if fast_discard is not None and method == "poisson_binomial":
            uncert, maxerrors = fast_discard
            r = eng.filter(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, uncert=uncert,
                           maxerrors=maxerrors, decision_only=True)
            return r.ee    # settled reads: +inf from the kernel; a NaN is a genuine failure and raises ReturnedNaNError
        run = eng.filter_poisson if method == "poisson" else eng.filter
        return run(q, lens=lens, alpha=alpha, ambigs=ambigs, round_=round_, uncert=1.0).ee
    backend.engine = eng
    backend.matrix = matrix
    backend.per_read = eng.calculate_errors_PB         # any int quality (a read whose scores do not fit the byte matrix)
    backend.per_read_poisson = eng.calculate_errors_poisson
    backend.methods = ("poisson_binomial", "poisson")
    return backend


def process_chunk(records, args, backend):
    """records: list of (header, fseq, fquals, rseq, rquals) -> list of
    (header, contig, contig_quals, expected_errors, overlap_length, gaps, mismatches)."""
    n = len(records)
    if args.paired:
        from . import contig as CT
        seqs, cq, clen, ov, gaps, mism = CT.contigs_batch(
            [r[1] for r in records], [r[2] for r in records], [r[3] for r in records], [r[4] for r in records],
            args.match, args.mismatch, args.gap, args.insert, args.deltaq, args.consensus_qscore,
            args.qscore_cap, args.trim_overlap, threads=args.processors)
        quals = [cq[i, :clen[i]] for i in range(n)]                  # int32 views; lists only when written
        ov, gaps, mism = ov.tolist(), gaps.tolist(), mism.tolist()
    else:
        seqs = [r[1] for r in records]
        quals = [r[2] for r in records]
        ov = gaps = mism = [0] * n
    if args.truncate:
        seqs = [s[:args.truncate] for s in seqs]
        quals = [q[:args.truncate] for q in quals]
    if args.only_contig:
        ee = [0] * n
    else:
        quals = [ql if isinstance(ql, QualStr) else                                     # moira.py:814 (Q0 -> 1);
                 (np.maximum(ql, 1) if isinstance(ql, np.ndarray) else [q if q > 0 else 1 for q in ql])
                 for ql in quals]                                                       # QualStr clamps in ints()
        big = []                                             # reads with a score the byte matrix cannot hold (> 254)
        if args.error_calc in ("poisson_binomial", "poisson_binomial_py", "poisson") and getattr(backend, "methods", None):
            for i, ql in enumerate(quals):                   # the packed matrix holds one byte per base
                if not isinstance(ql, QualStr) and len(ql) and int(max(ql)) > 254:
                    if not getattr(backend, "per_read_poisson" if args.error_calc == "poisson" else "per_read", None):
                        raise QualityTooHighError(records[i][0], int(max(ql)))
                    big.append(i)
        bset = set(big)
        bq = [[min(int(v), 254) for v in ql] if i in bset else ql for i, ql in enumerate(quals)] if big else quals

        def per_read(fn, seqs=seqs):
            # a .qual file may hold any integer and the reference takes it (moira/bernoullimodule.c:92-108,
            # moira/moira.py:1637-1679): such a read goes through the per-read entry (its own code table for that
            # call); the batch saw a capped stand-in
            for i in big:
                try:
                    e, ns = fn(seqs[i], [int(v) for v in quals[i]], args.alpha)
                except OverflowError:                       # the Poisson function's own failure: NaN, as in the batch
                    e, ns = float("nan"), 0
                if args.ambigs == "treat_as_errors":                                    # moira.py:826-827
                    e = e + ns
                if args.round:                                                          # moira.py:828-829
                    e = math.floor(e)
                ee[i] = float(e)
        if args.error_calc in ("poisson_binomial", "poisson_binomial_py"):
            for i, sq in enumerate(seqs):
                if len(sq) > MAX_PB_LEN:
                    raise ReadTooLongError(records[i][0], len(sq))
            scored = seqs
            if args.error_calc == "poisson_binomial_py":        # the Python twin counts only 'N' (moira/moira.py:1605): for it
                scored = [s.replace("n", "A") if "n" in s else s for s in seqs]      # a lower-case n is a base like any other
            if getattr(args, "fast_discard", False) and not args.collapse and args.pipeline == "mothur" \
                    and "poisson" in getattr(backend, "methods", ()):
                ee = backend(scored, bq, args.alpha, args.ambigs, args.round,
                             fast_discard=(args.uncert, args.maxerrors))
            else:
                ee = backend(scored, bq, args.alpha, args.ambigs, args.round)          # includes +Ns / floor
            ee = [float(x) for x in ee]
            per_read(getattr(backend, "per_read", None), scored)
        elif args.error_calc == "poisson" and "poisson" in getattr(backend, "methods", ()):
            ee = [float(x) for x in backend(seqs, bq, args.alpha, args.ambigs, args.round, method="poisson")]
            per_read(getattr(backend, "per_read_poisson", None))
        elif args.error_calc == "poisson":
            raise RuntimeError("this backend has no Poisson method (the HIP library provides it; there is no CPU path)")
        else:
            ee = []
            for s, ql in zip(seqs, quals):
                ql = ql.ints() if isinstance(ql, QualStr) else ql
                e, ns = calculate_errors_bootstrap(s, ql, args.alpha, args.bootstrap)
                if args.ambigs == "treat_as_errors":
                    e = e + ns
                if args.round:
                    e = math.floor(e)
                ee.append(e)
    return [(records[i][0], seqs[i], quals[i], ee[i], ov[i], gaps[i], mism[i]) for i in range(n)]


# ---------------------------------------------------------------------------------------------
# output (ref: moira/moira.py:842-970)
# ---------------------------------------------------------------------------------------------
class Outputs:
    def __init__(self):
        self.contig = self.qual = self.names = None
        self.bad_contig = self.bad_qual = self.bad_names = None
        self.report = None
        self.files = []


def write_results(index, header, sequence, quals, expected_errors, names_info, overlap_length, gaps,
                  mismatches, args, o):
    """Returns (discarded_errors, discarded_minlength, discarded_minoverlap)."""
    if args.relabel:
        header = "%s%d" % (args.relabel, index)
    if args.pipeline == "USEARCH":
        size = len(names_info) if names_info else 1
        header = header + ";ee=%.2f;size=%d;" % (expected_errors, size)
    if args.paired:
        o.report.write("%s\t%s\t%s\t%s\t%s\n" % (header, len(names_info) if args.collapse else 1,
                                                 overlap_length, gaps, mismatches))
    fq = args.output_format == "fastq"

    def qstr():
        if isinstance(quals, QualStr) and quals.offset == args.fastq_offset:
            return quals.fastq_line()
        return "".join([chr(q + args.fastq_offset) for q in (quals.ints() if isinstance(quals, QualStr) else quals)])

    def qline():
        return quals.qual_line() if isinstance(quals, QualStr) else " ".join(map(str, quals))

    def bad(label, names_header, counts):
        if fq:
            o.bad_contig.write("@%s\t%s\n%s\n+\n%s\n" % (header, label, sequence, qstr()))
        else:
            o.bad_contig.write(">%s\t%s\n%s\n" % (header, label, sequence))
            o.bad_qual.write(">%s\t%s\n%s\n" % (header, label, qline()))
        if args.collapse:
            if args.pipeline == "mothur":
                o.bad_names.write("%s\t%s\n" % (names_header, ",".join(names_info)))
            n = len(names_info)
            return tuple(n if c else 0 for c in counts)
        return counts

    def good():
        if fq:
            o.contig.write("@%s\n%s\n+\n%s\n" % (header, sequence, qstr()))
        else:
            o.contig.write(">%s\n%s\n" % (header, sequence))
            o.qual.write(">%s\n%s\n" % (header, qline()))
        return (0, 0, 0)

    if args.truncate and len(sequence) < args.truncate:
        return bad("length below %s" % args.truncate, header.lstrip(">"), (0, 1, 0))
    if args.min_overlap and overlap_length < args.min_overlap:
        # the reference's fastq branch prints args.truncate here (moira.py:890); kept
        return bad("overlap length below %s" % (args.truncate if fq else args.min_overlap),
                   header.lstrip(">"), (0, 0, 1))
    if args.only_contig:
        r = good()
        if args.collapse and args.pipeline == "mothur":
            o.names.write("%s\t%s\n" % (header.lstrip(">"), ",".join(names_info)))
        return r
    if "N" in sequence and args.ambigs == "disallow":
        return bad("contains ambiguities", header, (1, 0, 0))
    if args.maxerrors:
        keep, label, nh = expected_errors <= args.maxerrors, "errors > %.2f" % args.maxerrors, header.lstrip(">")
    else:
        keep, label, nh = expected_errors <= len(sequence) * args.uncert, "uncert > %.3f" % args.uncert, header
    if keep:
        r = good()
        if args.collapse and args.pipeline == "mothur":
            o.names.write("%s\t%s\n" % (header, ",".join(names_info)))
        return r
    return bad(label, nh, (1, 0, 0))


BGZF_DATA = 0xff00            # text bytes per BGZF member: whatever deflate makes of it fits the 64 KiB the size field can state
BGZF_EOF = bytes.fromhex("1f8b08040000000000ff0600424302001b0003000000000000000000")     # the format's end marker: an empty member


def bgzf_compress(data, level=4):
    """`data` as BGZF members (SAM/BAM specification section 4.1): ordinary gzip members -- gzip, zcat and Python's gzip
    read the file as usual -- of at most 64 KiB each that state their compressed size in a 'B','C' extra subfield, so a
    reader that knows the convention (bgzip, htslib, this package's GzipReader) inflates them on several threads."""
    mv = memoryview(data).cast("B")
    out = []
    for pos in range(0, len(mv), BGZF_DATA):
        piece = mv[pos:pos + BGZF_DATA]
        c = zlib.compressobj(level, zlib.DEFLATED, -15)
        z = c.compress(piece) + c.flush()
        if len(z) + 26 > 65536:                       # cannot happen at this piece size (deflate's bound); stored if it did
            c = zlib.compressobj(0, zlib.DEFLATED, -15)
            z = c.compress(piece) + c.flush()
        out.append(b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\0BC\x02\0" + struct.pack("<H", len(z) + 25))
        out.append(z)
        out.append(struct.pack("<II", zlib.crc32(piece), len(piece)))
    return b"".join(out)


class _BlockCompressedWriter:
    """A .gz output written as a sequence of independently compressed blocks (gzip members: the format defines the
    concatenation of valid files as a valid file, and gzip / zcat / Python 2 and 3 read it as one), so that the blocks are
    compressed on several threads (zlib releases the GIL) while the caller goes on formatting; a .bz2 output is ONE bzip2
    stream compressed in order by a thread of its own (Python 2's BZ2File, the reference's reader, stops after the first
    stream of a multi-stream file):
    write() cuts its data into 1 MiB blocks, hands them to a pool and writes out, in order, whatever has finished; at
    most `WINDOW` blocks are in flight.  One compressor thread per file is what bounds a run with compressed output
    otherwise: ~60 MB/s of text per file at gzip level 4.  A .gz output is BGZF (bgzf_compress: each 1 MiB block becomes
    17 members, the file ends with the format's end marker), so it can also be READ on several threads."""
    BLOCK = 1 << 20
    WINDOW = 64
    _pool = None

    @classmethod
    def pool(cls):
        if cls._pool is None:
            from concurrent.futures import ThreadPoolExecutor
            from .contig import usable_cpus
            cls._pool = ThreadPoolExecutor(max(2, min(usable_cpus(), 32)))
        return cls._pool

    def __init__(self, path, kind):
        import collections
        self.f = open(path, "wb")
        self.kind = kind
        self.carry = b""
        self.inflight = collections.deque()
        self.wrote = False
        # .bz2: ONE bzip2 stream, compressed in order by one thread of its own.  Python 2's bz2.BZ2File -- what the
        # reference reads its inputs with (moira/moira.py:1083-1084) -- stops after the first stream of a file, so a
        # multi-stream .bz2 written here and fed back to moira.py would be cut short without a word
        self.bz = bz2.BZ2Compressor() if kind == "bz2" else None
        self.own = None
        if self.bz is not None:
            from concurrent.futures import ThreadPoolExecutor
            self.own = ThreadPoolExecutor(1)

    def _pack(self, block):
        if self.kind == "gz":
            return bgzf_compress(block, 4)                      # level 4: same content, ~4x the speed of level 9
        return self.bz.compress(block)                          # (tasks of the one-thread pool run in submission order)

    def _drain(self, keep):
        while len(self.inflight) > keep:
            self.f.write(self.inflight.popleft().result())      # in order; waits for the oldest block only
            self.wrote = True

    def _submit(self, block):
        self.inflight.append((self.own or self.pool()).submit(self._pack, block))
        self._drain(self.WINDOW)

    def write(self, data):
        mv = memoryview(data).cast("B")
        n, pos = len(mv), 0
        if self.carry:
            need = self.BLOCK - len(self.carry)
            if n < need:
                self.carry += bytes(mv)
                return n
            self._submit(self.carry + bytes(mv[:need]))
            self.carry, pos = b"", need
        while n - pos >= self.BLOCK:
            self._submit(bytes(mv[pos:pos + self.BLOCK]))        # a copy: the caller's buffer is reused after write() returns
            pos += self.BLOCK
        if pos < n:
            self.carry = bytes(mv[pos:])
        while self.inflight and self.inflight[0].done():         # whatever has finished, without waiting
            self.f.write(self.inflight.popleft().result())
            self.wrote = True
        return n

    def close(self):
        if self.f is not None:
            if self.carry:
                self._submit(self.carry)
                self.carry = b""
            self._drain(0)
            if self.kind == "gz":
                self.f.write(BGZF_EOF)                          # (also what makes an empty output a valid archive)
            else:
                self.f.write(self.bz.flush())                   # the end of the one stream (an empty file is a valid archive too)
                self.own.shutdown()
            self.f.close()
            self.f = None


def _open_outputs(args, output_name, binary=False):
    mode = "wb" if binary else "wt"
    if args.output_compression == "gz":
        opener, suffix = (lambda p: _BlockCompressedWriter(p, "gz") if binary else gzip.open(p, mode, compresslevel=4)), ".gz"
    elif args.output_compression == "bz2":
        opener, suffix = (lambda p: _BlockCompressedWriter(p, "bz2") if binary else bz2.open(p, mode)), ".bz2"
    else:
        opener, suffix = (lambda p: open(p, mode[:2] if binary else "w")), ""
    o = Outputs()

    def mk(stem):
        path = "%s.%s%s" % (output_name, stem, suffix)
        o.files.append(path)
        return opener(path)
    fq = args.output_format == "fastq"
    names = args.collapse and args.pipeline == "mothur"
    if args.only_contig:
        o.contig = mk("contigs.fastq" if fq else "contigs.fasta")
        o.qual = None if fq else mk("contigs.qual")
        o.names = mk("contigs.names") if names else None
        if args.truncate or args.min_overlap:
            o.bad_contig = mk("bad.contigs.fastq" if fq else "bad.contigs.fasta")
            o.bad_qual = None if fq else mk("bad.contigs.qual")
            o.bad_names = mk("bad.contigs.names") if names else None
    else:
        o.contig = mk("qc.good.fastq" if fq else "qc.good.fasta")
        o.qual = None if fq else mk("qc.good.qual")
        o.bad_contig = mk("qc.bad.fastq" if fq else "qc.bad.fasta")
        o.bad_qual = None if fq else mk("qc.bad.qual")
        if names:
            o.names, o.bad_names = mk("qc.good.names"), mk("qc.bad.names")
    if args.paired:
        o.report = mk("contigs.report")
        head = "header\tn_seqs\toverlap_length\tgaps\tmismatches\n"
        o.report.write(head.encode() if binary else head)
    return o


def _close(o):
    for f in (o.contig, o.qual, o.names, o.bad_contig, o.bad_qual, o.bad_names, o.report):
        if f is not None:
            try:
                f.close()
            except Exception:
                pass


# ---------------------------------------------------------------------------------------------
# byte-level path for FASTQ input (include/moira_io.h): same results, no per-read Python text work
# ---------------------------------------------------------------------------------------------
PAIR_CHUNK_READS = 65536      # a contig slot is header + 2 x (l1 + l2) bytes


def _fast_eligible(args, backend):
    if os.environ.get("MOIRA_NO_FASTIO") or not (args.forward_fastq or (args.forward_fasta and args.forward_qual)):
        return False
    if not args.paired and args.min_overlap:
        return False
    if args.only_contig:
        return True                                   # no error calculation at all: contigs straight to the writers
    return bool(args.error_calc in ("poisson_binomial", "poisson_binomial_py", "poisson")
                and getattr(backend, "matrix", None) is not None
                and (args.error_calc != "poisson" or "poisson" in getattr(backend, "methods", ())))


def _record_error(which, e, args):
    """The exception the line parser raises for a record that fails its checks (moira.py:1178-1195;
    the reverse file's empty-line errors name the forward header and file, as there)."""
    from . import fastio as F
    if which == 0:
        name = args.forward_fastq
        if e.kind == F.REC_EMPTY_SEQ:
            return EmptySeqError(e.header, name)
        if e.kind == F.REC_EMPTY_QUAL:
            return EmptyQualError(e.header, name)
        return LengthMismatchError(e.header, name)
    if e.kind == F.REC_EMPTY_SEQ:
        return EmptySeqError(e.forward_header, args.forward_fastq)
    if e.kind == F.REC_EMPTY_QUAL:
        return EmptyQualError(e.forward_header, args.forward_fastq)
    return LengthMismatchError(e.header, args.reverse_fastq)


def _text_threads(args):
    """Threads the text side of a run may use for ONE call (reading, indexing, packing or formatting a chunk):
    --processors, at most the CPUs this process is granted."""
    from .contig import usable_cpus
    return max(1, min(int(args.processors or 1), usable_cpus(), 64))


def _fast_chunks(args):
    """(buf, idx, aux) per chunk: reads as they are in the file, or contigs built from the two files."""
    from . import fastio as F
    if not args.forward_fastq:
        yield from _fast_chunks_fasta_qual(args)
        return
    if not args.paired:
        fh = open_input_binary(args.forward_fastq, _text_threads(args))
        try:
            for buf, idx in F.FastqChunks(fh, CHUNK_READS, threads=_text_threads(args)):
                yield buf, idx, None
        except F.RecordError as e:
            raise _record_error(0, e, args)
        finally:
            fh.close()
        return
    from . import contig as CT
    half = max(1, _text_threads(args) // 2)
    ffh, rfh = open_input_binary(args.forward_fastq, half), open_input_binary(args.reverse_fastq, half)
    try:
        # reading + indexing the two files runs one chunk ahead of contig construction, on a thread of its own
        for fbuf, fidx, rbuf, ridx in _prefetched(iter(F.PairedFastqChunks(ffh, rfh, PAIR_CHUNK_READS, threads=_text_threads(args))), depth=1):
            # forward_header != reverse_header (moira.py:1197-1198); ':' -> '_' on both sides cannot change equality
            bad = F.first_header_mismatch(fbuf, fidx, rbuf, ridx)
            n = len(fidx) if bad < 0 else bad
            if n:
                try:
                    cbuf, cidx, aux = CT.contigs_from_fastq(
                        fbuf, fidx[:n], rbuf, ridx[:n], args.fastq_offset, args.match, args.mismatch, args.gap,
                        args.insert, args.deltaq, args.consensus_qscore, args.qscore_cap, args.trim_overlap,
                        threads=args.processors)
                except CT.QualityRange as e:
                    raise F.Unsupported(str(e))
                yield cbuf, cidx, aux
            if bad >= 0:
                raise NameMismatchError(F.header_of(fbuf, fidx[bad]), None, F.header_of(rbuf, ridx[bad]), None)
    except F.PairedRecordError as e:
        raise _record_error(e.which, e.err, args)
    finally:
        ffh.close()
        rfh.close()


def _prefetched(gen, depth=2):
    """Run a generator in a background thread, `depth` items ahead.  The chunk producers spend their time
    in C calls that release the GIL (decompression, indexing, contig construction), so reading chunk
    k+1 overlaps packing, filtering and writing chunk k.  Exceptions travel with the items."""
    import queue
    import threading
    q = queue.Queue(maxsize=depth)
    stop = threading.Event()
    END = object()

    def work():
        try:
            for item in gen:
                while not stop.is_set():
                    try:
                        q.put((item, None), timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    break
            else:
                q.put((END, None))
                return
        except BaseException as e:          # noqa: B902 -- re-raised in the consumer
            q.put((END, e))
            return
        finally:
            gen.close()

    t = threading.Thread(target=work, daemon=True)
    t.start()
    try:
        while True:
            item, err = q.get()
            if err is not None:
                raise err
            if item is END:
                return
            yield item
    finally:
        stop.set()
        while t.is_alive():                  # drain so that a producer blocked on put() can finish
            try:
                q.get_nowait()
            except queue.Empty:
                t.join(0.05)


class _InOrder:
    """Calls run one after the other in a background thread (formatting + writing chunk k while chunk
    k+1 is packed and filtered; the C formatters and file writes release the GIL).  An exception of a
    call is raised by the next submit() or by close()."""

    def __init__(self, depth=2):
        import queue
        import threading
        self.q = queue.Queue(maxsize=depth)
        self.err = None
        self.t = threading.Thread(target=self._work, daemon=True)
        self.t.start()

    def _work(self):
        while True:
            fn = self.q.get()
            if fn is None:
                return
            if self.err is None:
                try:
                    fn()
                except BaseException as e:      # noqa: B902 -- re-raised in the submitting thread
                    self.err = e

    def submit(self, fn):
        if self.err is not None:
            raise self.err
        self.q.put(fn)

    def close(self, raise_errors=True):
        self.q.put(None)
        self.t.join()
        if raise_errors and self.err is not None:
            raise self.err


def _fast_chunks_fasta_qual(args):
    """fasta + qual input: records rebuilt as header | sequence | quality bytes (offset 0) by
    mio_fasta_qual_index, so that the rest of the path is the FASTQ one."""
    from . import fastio as F
    half = max(1, _text_threads(args) // 2)
    files = [open_input_binary(args.forward_fasta, half), open_input_binary(args.forward_qual, half)]
    try:
        if not args.paired:
            for buf, idx in F.FastaQualChunks(files[0], files[1], CHUNK_READS):
                yield buf, idx, None
            return
        from . import contig as CT
        files += [open_input_binary(args.reverse_fasta, half), open_input_binary(args.reverse_qual, half)]
        fwd = iter(F.FastaQualChunks(files[0], files[1], PAIR_CHUNK_READS))
        rev = iter(F.FastaQualChunks(files[2], files[3], PAIR_CHUNK_READS))
        while True:
            a, b = next(fwd, None), next(rev, None)
            if a is None and b is None:
                return
            # a file that ends early, or names that differ, are NameMismatchErrors of the line parser
            if a is None or b is None or len(a[1]) != len(b[1]) or F.first_header_mismatch(a[0], a[1], b[0], b[1]) >= 0:
                raise F.Unsupported("forward and reverse records do not pair up")
            try:
                cbuf, cidx, aux = CT.contigs_from_fastq(
                    a[0], a[1], b[0], b[1], 0, args.match, args.mismatch, args.gap, args.insert, args.deltaq,
                    args.consensus_qscore, args.qscore_cap, args.trim_overlap, threads=args.processors)
            except CT.QualityRange as e:
                raise F.Unsupported(str(e))
            yield cbuf, cidx, aux
    finally:
        for f in files:
            f.close()


def _run_fast_fastq(args, backend, o, say, t0):
    """Chunks of the input as (buffer, record index); contig construction, packing, collapse and record
    formatting in C (moira_amd/fastio.py, moira_amd/contig.py).  Decisions are write_results'
    (ref: moira/moira.py:842-970), vectorised.
    Returns (processed, discarded_errors, discarded_minlength, discarded_minoverlap)."""
    from . import fastio as F
    from .buckets import bucket_of
    T = args.truncate or 0
    only = bool(args.only_contig)                                # moira.py:809-810, :898-908: contigs, no quality control
    in_off = args.fastq_offset if args.forward_fastq else 0      # fasta+qual records carry the integers themselves
    method = "poisson" if args.error_calc == "poisson" else "poisson_binomial"
    # a lower-case 'n' is an ambiguous base for the C extension only (moira/bernoullimodule.c:196); the reference's Python
    # functions -- the Poisson approximation and the twin that --error_calc poisson_binomial_py selects -- score it as a base
    # (moira/moira.py:1605, :1660 test for 'N' alone)
    n_is_base = args.error_calc in ("poisson", "poisson_binomial_py")
    fd = None
    if getattr(args, "fast_discard", False) and not args.collapse and args.pipeline == "mothur" \
            and method == "poisson_binomial":
        fd = (args.uncert, args.maxerrors)
    fq = args.output_format == "fastq"
    usearch = args.pipeline == "USEARCH"
    thr_label = ("errors > %.2f" % args.maxerrors) if args.maxerrors else ("uncert > %.3f" % args.uncert)
    # the fastq branch of the reference prints args.truncate in the overlap label (moira.py:888); kept
    labels = ["length below %s" % args.truncate, "contains ambiguities", thr_label,
              "overlap length below %s" % (args.truncate if fq else args.min_overlap)]
    min_ov = args.min_overlap if args.paired and args.min_overlap else 0

    def decide(ee, length, has_n, ov):
        """write_results' branch per record or group: -1 good, else the index of its label."""
        label = np.full(len(ee), -1, np.int32)
        if not only:
            keep = (ee <= args.maxerrors) if args.maxerrors else (ee <= length * args.uncert)   # len(sequence) * uncert
            label[~keep] = 2
            if args.ambigs == "disallow":
                label[has_n] = 1
        if min_ov:
            label[ov < min_ov] = 3
        if T:
            label[length < T] = 0
        return label

    processed = 0
    disc_err = disc_len = disc_ov = 0.0
    threads = _text_threads(args)                        # --processors: packing / formatting calls in flight
    groups = F.Collapse(threads) if args.collapse else None
    pool = None
    if threads > 1:
        from concurrent.futures import ThreadPoolExecutor
        pool = ThreadPoolExecutor(threads)
    emit = _InOrder()
    # one writer lane per output file: the files of a chunk are formatted (by the pool) and written side by side, each
    # file's chunks in order.  Buffered writes to ONE file are serialised by the kernel anyway; different files are not.
    lanes = {}

    def lane(f):
        if threads <= 1:
            return emit
        k = id(f)
        if k not in lanes:
            lanes[k] = _InOrder()
        return lanes[k]
    ok = False
    try:
        for buf, idx, aux in _prefetched(_fast_chunks(args)):
            n = len(idx)
            lens = np.minimum(idx[:, F.SEQ_LEN], T) if T else idx[:, F.SEQ_LEN].copy()
            if not only and method == "poisson_binomial" and n and int(lens.max()) > MAX_PB_LEN:
                k = int(np.argmax(lens))
                raise ReadTooLongError(F.header_of(buf, idx[k]), int(lens[k]))
            strides = bucket_of(lens, 64)
            ee = np.zeros(n, np.float64)                 # --only_contig: process_data returns 0 (moira.py:809-810)
            has_n = np.zeros(n, bool)
            for stride in (np.unique(strides) if not only else ()):
                sel = np.nonzero(strides == stride)[0]
                q, ln, fl = F.pack_parallel(pool, threads, buf, idx, sel, in_off, T, n_is_base, int(stride))
                ee[sel] = backend.matrix(q, ln, args.alpha, args.ambigs, args.round, method=method, fast_discard=fd)
                has_n[sel] = fl
            nan = np.isnan(ee)
            if nan.any():
                raise ReturnedNaNError(F.header_of(buf, idx[int(np.argmax(nan))]), int(lens[int(np.argmax(nan))]))
            if args.collapse:
                emit.submit(lambda buf=buf, idx=idx, ee=ee, has_n=has_n, aux=aux: groups.add(buf, idx, ee, has_n, T, aux))
            else:
                label = decide(ee, lens, has_n, aux[:, 0] if aux is not None else None)
                disc_len += int((label == 0).sum())
                disc_ov += int((label == 3).sum())
                disc_err += int(((label == 1) | (label == 2)).sum())

                hdr = dict(relabel=args.relabel or None)
                if args.paired:
                    def write_report(buf=buf, idx=idx, aux=aux, ee=ee, first=processed, n=n, hdr=hdr):
                        everything = np.arange(n)
                        o.report.write(F.format_report(buf, idx, everything, aux, relabel_index=first + everything,
                                                       ee=ee if usearch else None, **hdr))
                    lane(o.report).submit(write_report)
                good, bad = np.nonzero(label < 0)[0], np.nonzero(label >= 0)[0]
                for sel, main_f, qual_f, lab in ((good, o.contig, o.qual, None), (bad, o.bad_contig, o.bad_qual, label)):
                    if not len(sel):
                        continue
                    kw = dict(fastq_offset=in_off, out_offset=args.fastq_offset, clamp_q0=not only, max_len=T,
                              relabel_index=processed + sel,
                              ee=ee[sel] if usearch else None, labels=labels if lab is not None else None,
                              label_id=lab[sel] if lab is not None else None, **hdr)
                    for kind, f in (((F.FMT_FASTQ, main_f),) if fq else ((F.FMT_FASTA, main_f), (F.FMT_QUAL, qual_f))):
                        def write_file(buf=buf, idx=idx, sel=sel, kind=kind, f=f, kw=kw):
                            for piece in F.format_parallel(pool, threads, buf, idx, sel, kind, scratch="fmt%x_" % id(f), **kw):
                                f.write(piece)
                        lane(f).submit(write_file)
            processed += n
            if not args.silent:
                say("%d sequences processed in %.1f seconds." % (processed, time.time() - t0))
        emit.close()
        for ln in lanes.values():
            ln.close()
        ok = True
        if args.collapse:
            # the groups by decreasing abundance (moira.py:490-493), then write_results' decisions per group
            gee, glen, gsize, g_n, gaux = groups.export()
            label = decide(gee, glen, g_n, gaux[:, 0])
            disc_len += int(gsize[label == 0].sum())
            disc_ov += int(gsize[label == 3].sum())
            disc_err += int(gsize[(label == 1) | (label == 2)].sum())
            names = args.pipeline == "mothur"
            hdr = dict(fastq_offset=in_off, out_offset=args.fastq_offset, clamp_q0=not only, relabel=args.relabel or None,
                       usearch=usearch)
            if args.paired:
                o.report.write(groups.format(np.arange(len(gee)), F.FMT_REPORT, **hdr))
            # header.lstrip('>') on the names line of three kinds of bad groups, and of every group with
            # --only_contig (moira.py:880,894,907,943)
            strip = (label == 0) | (label == 3) | ((label == 2) & bool(args.maxerrors)) | ((label < 0) & only)
            jobs = []
            for sel, main_f, qual_f, names_f, lab in ((np.nonzero(label < 0)[0], o.contig, o.qual, o.names, None),
                                                      (np.nonzero(label >= 0)[0], o.bad_contig, o.bad_qual, o.bad_names, label)):
                if not len(sel):
                    continue
                kw = dict(labels=labels if lab is not None else None, label_id=lab[sel] if lab is not None else None, **hdr)
                jobs += [(sel, F.FMT_FASTQ, main_f, kw)] if fq else [(sel, F.FMT_FASTA, main_f, kw), (sel, F.FMT_QUAL, qual_f, kw)]
                if names:
                    jobs.append((sel, F.FMT_NAMES, names_f, dict(lstrip_gt=strip[sel], **hdr)))

            def write_file(sel, kind, f, k, tag):
                for piece in F.collapse_format_parallel(pool, threads, groups, sel, kind, scratch="cf%d_" % tag, **k):
                    f.write(piece)
            if threads > 1 and len(jobs) > 1:
                # every output file on its own thread: formatting goes to the pool, and writes to different files
                # do not wait for each other
                lanes_end = [_InOrder() for _ in jobs]
                for tag, (ln, (sel, kind, f, k)) in enumerate(zip(lanes_end, jobs)):
                    ln.submit(lambda sel=sel, kind=kind, f=f, k=k, tag=tag: write_file(sel, kind, f, k, tag))
                errs = []
                for ln in lanes_end:
                    try:
                        ln.close()
                    except BaseException as e:      # noqa: B902 -- every lane is joined before anything is raised
                        errs.append(e)
                if errs:
                    raise errs[0]
            else:
                for tag, (sel, kind, f, k) in enumerate(jobs):
                    write_file(sel, kind, f, k, tag)
    finally:
        if not ok:
            emit.close(raise_errors=False)         # an exception is already on its way
            for ln in lanes.values():
                ln.close(raise_errors=False)
        if pool is not None:
            pool.shutdown(wait=True)
        if groups is not None:
            groups.close()
    return processed, disc_err, disc_len, disc_ov


# ---------------------------------------------------------------------------------------------
# main (ref: moira/moira.py:264-578)
# ---------------------------------------------------------------------------------------------
def main(args, backend=None, out=None, _no_fastio=False):
    """`backend(seqs, quals, alpha, ambigs, round) -> ee` defaults to the HIP library; the parameter
    exists so the host logic can be unit-tested on machines without a GPU."""
    def say(msg=""):
        print(msg, file=out or sys.stdout)

    if not args.silent:
        say()
        say("-" * 79)
        say()
        say("moira (MI355X build) v%s" % __version__)
        say("Poisson-binomial read filtering after Puente-Sanchez F, Aguirre J, Parro V (2016), NAR 44(4): e40.")
        say()
        say("-" * 79)
        say()
    if not check_arguments(args, out):
        return 1
    if args.output_prefix:
        output_name = args.output_prefix
    elif args.forward_fastq:
        output_name = ".".join(args.forward_fastq.split(".")[:-1])
    else:
        output_name = ".".join(args.forward_fasta.split(".")[:-1])
    try:
        if args.forward_fastq:
            fwd = open_input(args.forward_fastq)
            rev = open_input(args.reverse_fastq) if args.paired else None
            parse = parse_fastq(fwd, rev, args.fastq_offset, raw=True)
        else:
            ff, fqf = open_input(args.forward_fasta), open_input(args.forward_qual)
            rf = open_input(args.reverse_fasta) if args.paired else None
            rq = open_input(args.reverse_qual) if args.paired else None
            parse = parse_fasta_and_qual(ff, fqf, rf, rq)
    except IOError as e:
        say(str(e))
        say()
        return 1
    needs_gpu = (not args.only_contig) and args.error_calc in ("poisson_binomial", "poisson_binomial_py", "poisson")
    if backend is None and needs_gpu:
        backend = make_gpu_backend(getattr(args, "device", None))
    fast = _fast_eligible(args, backend) and not _no_fastio
    try:
        o = _open_outputs(args, output_name, binary=fast)
    except IOError as e:
        say(str(e))
        say()
        return 1
    try:
        processed = 0
        disc_err = disc_len = disc_ov = 0.0
        uniques = Py2Dict() if args.collapse else None
        t0 = time.time()
        chunk = []

        def flush(chunk):
            nonlocal processed, disc_err, disc_len, disc_ov
            for header, contig, cquals, ee, ov, gaps, mism in process_chunk(chunk, args, backend):
                if isinstance(ee, float) and math.isnan(ee):
                    raise ReturnedNaNError(header, len(contig))
                if args.collapse:
                    if contig not in uniques:
                        uniques[contig] = {"rep_header": header, "rep_errors": ee, "rep_quals": cquals,
                                           "names_info": [header], "overlap_length": ov, "gaps": gaps,
                                           "mismatches": mism}
                    else:
                        u = uniques[contig]
                        if ee < u["rep_errors"]:                                # strict: first seen wins ties
                            u.update(rep_header=header, rep_errors=ee, rep_quals=cquals, overlap_length=ov,
                                     gaps=gaps, mismatches=mism)
                            u["names_info"].insert(0, header)
                        else:
                            u["names_info"].append(header)
                else:
                    r = write_results(processed, header, contig, cquals, ee, None, ov, gaps, mism, args, o)
                    disc_err += r[0]; disc_len += r[1]; disc_ov += r[2]
                processed += 1
            if not args.silent:
                say("%d sequences processed in %.1f seconds." % (processed, time.time() - t0))

        if fast:
            from . import fastio
            try:
                processed, disc_err, disc_len, disc_ov = _run_fast_fastq(args, backend, o, say, t0)
            except fastio.Unsupported:
                # content the byte-level parser does not reproduce: start over with the line parser
                _close(o)
                return main(args, backend=backend, out=out, _no_fastio=True)
        else:
            for rec in parse:
                chunk.append(rec)
                if len(chunk) >= CHUNK_READS:
                    flush(chunk)
                    chunk = []
            if chunk:
                flush(chunk)
        if args.collapse:
            order = sorted(uniques, key=lambda s: len(uniques[s]["names_info"]), reverse=True)   # stable
            for index, sequence in enumerate(order, start=1):
                v = uniques[sequence]
                r = write_results(index, v["rep_header"], sequence, v["rep_quals"], v["rep_errors"],
                                  v["names_info"], v["overlap_length"], v["gaps"], v["mismatches"], args, o)
                disc_err += r[0]; disc_len += r[1]; disc_ov += r[2]
        if not args.silent and processed:
            remaining = processed - disc_err - disc_len - disc_ov
            say("- Kept %d (%.2f%%) of the original sequences." % (remaining, remaining / processed * 100))
            if args.truncate:
                say("- %d (%.2f%%) of the original sequences were discarded due to length < %s."
                    % (disc_len, disc_len / processed * 100, args.truncate))
            if args.paired and args.min_overlap:
                say("- %d (%.2f%%) of the original sequences were discarded due to paired-end reads having "
                    "an overlap length < %s." % (disc_ov, disc_ov / processed * 100, args.min_overlap))
            say("- %d (%.2f%%) of the original sequences were discarded due to low quality.\n"
                % (disc_err, disc_err / processed * 100))
            say("The following output files were generated:")
            for p in o.files:
                say(p)
            say()
    except UnsupportedReadError as e:
        _close(o)
        for p in o.files:                       # nothing half-written is left behind
            try:
                os.remove(p)
            except OSError:
                pass
        say(str(e))
        say()
        return 1
    finally:
        _close(o)
    return 0


def cli(argv=None):
    return main(parse_arguments(argv))


if __name__ == "__main__":
    sys.exit(cli())
