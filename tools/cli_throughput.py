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

n = int(sys.argv[1]) if len(sys.argv) > 1 else 300_000
q, _ = O.synth_fill(n, 256, fixed_len=250, seed=1)
rng = np.random.default_rng(1)
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 250))]
bases[q[:, :250] == 0] = ord("N")
qa = (np.maximum(q[:, :250], 1) + 33).astype(np.uint8)
tmp = tempfile.mkdtemp()
path = os.path.join(tmp, "synth.fastq")
with open(path, "w") as f:
    for i in range(n):
        f.write("@r%d\n%s\n+\n%s\n" % (i, bases[i].tobytes().decode(), qa[i].tobytes().decode()))
args = cli.parse_arguments(["-ffq", path, "-op", os.path.join(tmp, "out"), "--silent", "-c", "false"])
t = time.perf_counter()
rc = cli.main(args, out=open(os.devnull, "w"))
dt = time.perf_counter() - t
good = sum(1 for l in open(os.path.join(tmp, "out.qc.good.fasta")) if l.startswith(">"))
print("CLI end to end: %d reads (250 bp, fastq in, fasta+qual out, no collapse) in %.2f s = %.0f reads/s; kept %d; rc=%d"
      % (n, dt, n / dt, good, rc))
