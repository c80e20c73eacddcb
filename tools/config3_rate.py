#!/usr/bin/env python3
"""BASELINE config 3 in miniature: paired 2 x 300 bp reads -> CPU contig construction
(libmoira_contig.so, all host cores) -> GPU filter.  Reports the two stages separately: the pipeline
is NW-bound by design (north_star keeps contig construction on the CPU)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd import contig as CT  # noqa: E402
from moira_amd.buckets import filter_bucketed  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
L, frag = 300, 450                      # 150 bp overlap
rng = np.random.default_rng(3)
B = np.frombuffer(b"ACGT", np.uint8)
comp = {ord("A"): ord("T"), ord("C"): ord("G"), ord("G"): ord("C"), ord("T"): ord("A")}
lut = np.zeros(256, np.uint8)
for k, v in comp.items():
    lut[k] = v
frags = B[rng.integers(0, 4, (n, frag))]
fwd = frags[:, :L].copy()
rev = lut[frags[:, frag - L:][:, ::-1]]
for a in (fwd, rev):                    # ~0.7 % substitutions, concentrated towards the 3' end
    pos = np.minimum((rng.random((n, 2)) ** 0.4 * L).astype(int), L - 1)
    a[np.arange(n)[:, None], pos] = B[rng.integers(0, 4, (n, 2))]
qual = np.clip(38 - (np.arange(L) / L) ** 3 * rng.integers(4, 30, (n, 1)) - rng.integers(0, 6, (n, L)), 2, 40).astype(np.int32)
fs = [r.tobytes().decode() for r in fwd]
rs = [r.tobytes().decode() for r in rev]
fq = [r for r in qual]
rq = [r[::1] for r in qual]
threads = os.cpu_count()
t = time.perf_counter()
seqs, cq, clen, ov, gaps, mism = CT.contigs_batch(fs, fq, rs, rq, threads=threads)
t_contig = time.perf_counter() - t
quals = [cq[i, :clen[i]] for i in range(n)]
with Engine(0) as eng:
    filter_bucketed(eng, seqs[:1000], quals[:1000])
    t = time.perf_counter()
    ee, ns, passed = filter_bucketed(eng, seqs, quals)
    t_filter = time.perf_counter() - t
print("config 3 (miniature): %d pairs 2x%d bp; contig len %d..%d, overlap median %d"
      % (n, L, clen.min(), clen.max(), int(np.median(ov))))
print("  contig construction (CPU, %d threads, incl. Python marshalling): %.2f s = %.0f pairs/s" % (threads, t_contig, n / t_contig))
print("  pack + GPU filter + gather (host lists in, PCIe):                %.2f s = %.0f contigs/s; kept %d" % (t_filter, n / t_filter, int(passed.sum())))
