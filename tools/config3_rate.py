#!/usr/bin/env python3
"""BASELINE config 3 in miniature: paired 2 x 300 bp reads (two FASTQ files) -> CPU contig construction
(libmoira_contig.so, all host cores, straight from the file buffers) -> pack -> GPU filter.  The stages are
timed separately: the pipeline is contig/text-bound by design (north_star keeps contig construction on
the CPU); the filter itself is a rounding error."""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd import contig as CT, fastio as F  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400_000
L, frag = 300, 450                      # 150 bp overlap
rng = np.random.default_rng(3)
B = np.frombuffer(b"ACGT", np.uint8)
lut = np.zeros(256, np.uint8)
for a, b in zip(b"ACGT", b"TGCA"):
    lut[a] = b
frags = B[rng.integers(0, 4, (n, frag))]
fwd, rev = frags[:, :L].copy(), lut[frags[:, frag - L:][:, ::-1]]
for a in (fwd, rev):                    # ~0.7 % substitutions, concentrated towards the 3' end
    pos = np.minimum((rng.random((n, 2)) ** 0.4 * L).astype(int), L - 1)
    a[np.arange(n)[:, None], pos] = B[rng.integers(0, 4, (n, 2))]
qual = (np.clip(38 - (np.arange(L) / L) ** 3 * rng.integers(4, 30, (n, 1)) - rng.integers(0, 6, (n, L)), 2, 40) + 33).astype(np.uint8)
tmp = tempfile.mkdtemp()
paths = []
for tag, arr in (("R1", fwd), ("R2", rev)):
    p = os.path.join(tmp, tag + ".fastq")
    with open(p, "wb") as f:
        for i in range(n):
            f.write(b"@p%d\n" % i + arr[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n")
    paths.append(p)
threads = int(sys.argv[3]) if len(sys.argv) > 3 else CT.usable_cpus()
t_index = t_contig = t_pack = t_filter = 0.0
kept = total = 0
with Engine(0) as eng:
    eng.filter(np.full((8, 608), 30, np.uint8), fixed_len=600)          # warm-up
    t0 = time.perf_counter()
    chunk_pairs = int(sys.argv[2]) if len(sys.argv) > 2 else 65536
    it = iter(F.PairedFastqChunks(open(paths[0], "rb"), open(paths[1], "rb"), chunk_pairs, block_bytes=1 << 27))
    while True:
        t = time.perf_counter()
        chunk = next(it, None)
        t_index += time.perf_counter() - t
        if chunk is None:
            break
        fbuf, fidx, rbuf, ridx = chunk
        t = time.perf_counter()
        cbuf, cidx, aux = CT.contigs_from_fastq(fbuf, fidx, rbuf, ridx, 33, threads=threads)
        t_contig += time.perf_counter() - t
        t = time.perf_counter()
        q, lens, has_n = F.pack(cbuf, cidx, None, 33, 0, stride=608, reuse=True)
        t_pack += time.perf_counter() - t
        t = time.perf_counter()
        r = eng.filter(q, lens=lens)
        t_filter += time.perf_counter() - t
        kept += r.n_pass
        total += len(lens)
    wall = time.perf_counter() - t0
print("config 3 (miniature): %d pairs 2x%d bp, %d host threads; contigs kept %d" % (total, L, threads, kept))
for name, t in (("read + index both files", t_index), ("contig construction (NW + consensus)", t_contig),
                ("pack into the quality matrix", t_pack), ("GPU filter incl. PCIe both ways", t_filter)):
    print("  %-40s %.3f s = %.3e pairs/s" % (name, t, total / t))
print("  %-40s %.3f s = %.3e pairs/s" % ("all stages, one after the other", wall, total / wall))
