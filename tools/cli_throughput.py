#!/usr/bin/env python3
"""End-to-end rate of the moira-compatible CLI on a synthetic FASTQ (parser-bound; reported
separately from bench.py's kernel-resident number)."""
import os
import sys
import tempfile
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import pb_oracle as O  # noqa: E402
from moira_amd import cli  # noqa: E402
from moira_amd.contig import usable_cpus  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
q, _ = O.synth_fill(n, 256, fixed_len=250, seed=1)
rng = np.random.default_rng(1)
tmp = tempfile.mkdtemp(dir=os.environ.get("CLI_TMP") or None)
path = os.path.join(tmp, "synth.fastq")


def write_fastq(pth, prefix, bases, quals):
    """n records of equal length as one byte matrix (fixed-width decimal ids), written in one go."""
    m, L = bases.shape
    w = len(str(m - 1))
    ids = np.char.zfill(np.arange(m).astype("U%d" % w), w).astype("S%d" % w).view(np.uint8).reshape(m, w)
    head = np.frombuffer(("@" + prefix).encode(), np.uint8)
    rec = np.empty((m, len(head) + w + 1 + L + 3 + L + 1), np.uint8)
    c = 0
    for part in (head, ids, b"\n", bases, b"\n+\n", quals, b"\n"):
        part = np.frombuffer(part, np.uint8) if isinstance(part, bytes) else part
        k = part.shape[-1]
        rec[:, c:c + k] = part
        c += k
    rec.tofile(pth)


bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 250))]
bases[q[:, :250] == 0] = ord("N")
write_fastq(path, "r", bases, (np.maximum(q[:, :250], 1) + 33).astype(np.uint8))
# the same reads drawn from a pool of n / 40 distinct sequences: what an amplicon run looks like to the collapse step
dup_path = os.path.join(tmp, "synth_dup.fastq")
pool_n = max(n // 40, 1)
pick = np.minimum((rng.random(n) ** 2 * pool_n).astype(np.int64), pool_n - 1)
write_fastq(dup_path, "r", bases[:pool_n][pick], (np.maximum(q[:, :250], 1) + 33).astype(np.uint8))
del bases, pick


def run(label, extra, env=None):
    old = os.environ.pop("MOIRA_NO_FASTIO", None)
    if env:
        os.environ.update(env)
    try:
        out = os.path.join(tmp, "out_" + "".join(ch if ch.isalnum() else "_" for ch in label))
        args = cli.parse_arguments(["-ffq", path, "-op", out, "--silent"] + extra)
        t = time.perf_counter()
        rc = cli.main(args, out=open(os.devnull, "w"))
        dt = time.perf_counter() - t
        print("CLI end to end [%s]: %d reads (250 bp) in %.2f s = %.0f reads/s; rc=%d" % (label, n, dt, n / dt, rc), flush=True)
        # a run leaves up to 10 GB of output in the page cache: remove it and let the write-back finish (untimed), or
        # the NEXT run is measured under the kernel's dirty-page throttling
        for f in os.listdir(tmp):
            if f.startswith("out_"):
                os.remove(os.path.join(tmp, f))
        os.sync()
    finally:
        os.environ.pop("MOIRA_NO_FASTIO", None)
        if old is not None:
            os.environ["MOIRA_NO_FASTIO"] = old



def paired_files(m, L=250, frag=380):
    """m pairs of 2 x L bp reads off random fragments (overlap L*2-frag), ~1 % substitutions."""
    rng = np.random.default_rng(7)
    B = np.frombuffer(b"ACGT", np.uint8)
    lut = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        lut[a] = b
    frags = B[rng.integers(0, 4, (m, frag))]
    fwd, rev = frags[:, :L].copy(), lut[frags[:, frag - L:][:, ::-1]]
    for a in (fwd, rev):
        pos = np.minimum((rng.random((m, 3)) ** 0.4 * L).astype(int), L - 1)
        a[np.arange(m)[:, None], pos] = B[rng.integers(0, 4, (m, 3))]
    qv = (np.maximum(q[:m, :L], 1) + 33).astype(np.uint8)
    paths = []
    for tag, arr in (("R1", fwd), ("R2", rev)):
        pth = os.path.join(tmp, "synth_%s.fastq" % tag)
        write_fastq(pth, "p", arr, qv)
        paths.append(pth)
    return paths


P = str(usable_cpus())
print("CPUs granted: %s; files under %s" % (P, tmp), flush=True)
run("warm-up", ["-c", "false"])
if os.environ.get("CLI_SKIP_SINGLE"):
    n_keep, n = n, 0
else:
    n_keep = n
if n:
    run("fastq in, fastq out, no collapse, -p 1", ["-c", "false", "-o", "fastq"])
    run("fastq in, fastq out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-p", P])
    run("fastq in, fasta+qual out, no collapse, -p 1", ["-c", "false"])
    run("fastq in, fasta+qual out, no collapse, -p " + P, ["-c", "false", "-p", P])
    run("fastq in, fasta+qual out, collapse (every read distinct: worst case), -p " + P, ["-c", "true", "-p", P])
    _keep = path
    path = dup_path
    run("fastq in, fasta+qual out, collapse (reads drawn from n/40 distinct sequences), -p " + P, ["-c", "true", "-p", P])
    run("fastq in, fasta+qual out, collapse (reads drawn from n/40 distinct sequences), -p 1", ["-c", "true"])
    path = _keep
    if n <= 1_000_000:
        run("line parser: fasta+qual out, no collapse", ["-c", "false"], {"MOIRA_NO_FASTIO": "1"})
n = n_keep
if os.environ.get("CLI_GZ"):
    # compressed output (block-parallel members) and compressed input (one inflating stream per file) on the first 2 M reads
    import subprocess
    m_gz = min(n, 2_000_000)
    small = os.path.join(tmp, "small.fastq")
    rec_bytes = os.path.getsize(path) // n
    with open(path, "rb") as f, open(small, "wb") as g:
        g.write(f.read(rec_bytes * m_gz))
    _keep, _n, path, n = path, n, small, m_gz
    run("fastq in, fastq.gz out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-oc", "gz", "-p", P])
    run("fastq in, fastq.gz out, no collapse, -p 1", ["-c", "false", "-o", "fastq", "-oc", "gz"])
    subprocess.check_call(["gzip", "-1", "-k", "-f", small])
    path = small + ".gz"
    run("fastq.gz in, fastq out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-p", P])
    run("fastq.gz in through zlib (MOIRA_ZLIB_INPUT=1), fastq out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-p", P],
        {"MOIRA_ZLIB_INPUT": "1"})
    os.environ.pop("MOIRA_ZLIB_INPUT", None)
    # the same text as BGZF members (what bgzip / Illumina's writers / this CLI's own .gz outputs are): inflated on -p threads
    from moira_amd.cli import BGZF_EOF, bgzf_compress
    path = os.path.join(tmp, "small_bgzf.fastq.gz")
    with open(small, "rb") as f, open(path, "wb") as g:
        while True:
            blk = f.read(64 * 0xff00)
            if not blk:
                break
            g.write(bgzf_compress(blk, 1))
        g.write(BGZF_EOF)
    run("fastq.gz (BGZF) in, fastq out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-p", P])
    run("fastq.gz (BGZF) in, fastq out, no collapse, -p 1", ["-c", "false", "-o", "fastq"])
    run("fastq.gz (BGZF) in, fastq.gz (BGZF) out, no collapse, -p " + P, ["-c", "false", "-o", "fastq", "-oc", "gz", "-p", P])
    path, n = _keep, _n

m = min(n, int(os.environ.get("CLI_PAIRS", "200000")))
r1, r2 = paired_files(m)


def run_paired(label, extra, env=None):
    global path, n
    keep = (path, n)
    path, n = r1, m
    try:
        run(label, ["-rfq", r2, "--paired", "-p", str(usable_cpus())] + extra, env)
    finally:
        path, n = keep


run_paired("paired 2x250: contigs + filter, collapse", ["-c", "true"])
run_paired("paired 2x250: contigs + filter, no collapse", ["-c", "false"])
run_paired("paired 2x250: --only_contig, no collapse", ["-c", "false", "--only_contig"])
if m <= 200_000:
    run_paired("line parser: paired, collapse", ["-c", "true"], {"MOIRA_NO_FASTIO": "1"})
import shutil  # noqa: E402
shutil.rmtree(tmp, ignore_errors=True)
