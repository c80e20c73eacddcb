"""BASELINE configs[2]: "paired 2 x 300 bp reads, NW contig on CPU then GPU filter" at a size the oracle covers:
120 k synthetic pairs (150 bp overlap, substitutions towards the 3' end, a few Ns) written as two FASTQ files ->
record index -> contig construction on the CPU straight from the file buffers (libmoira_contig.so, pinned to the
reference's aligner by tests/golden/nw_pairs.npz) -> packed ragged matrix (contigs of 300..600 bases, stride 608)
-> GPU filter in chunks.  Every contig's ee / Ns / decision is compared with the oracle run on the same packed
rows, and a sample of the contigs with the list-based contig entry (the path the reference-fixture tests pin)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_pairs(tmp, n, L=300, frag=450, seed=3):
    rng = np.random.default_rng(seed)
    B = np.frombuffer(b"ACGT", np.uint8)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGTN", b"TGCAN"):
        comp[a] = b
    frags = B[rng.integers(0, 4, (n, frag))]
    fwd, rev = frags[:, :L].copy(), comp[frags[:, frag - L:][:, ::-1]]
    for a in (fwd, rev):
        pos = np.minimum((rng.random((n, 2)) ** 0.4 * L).astype(int), L - 1)
        a[np.arange(n)[:, None], pos] = B[rng.integers(0, 4, (n, 2))]
        amb = rng.random((n, L)) < 0.0005
        a[amb] = ord("N")
    qual = (np.clip(38 - (np.arange(L) / L) ** 3 * rng.integers(4, 30, (n, 1)) - rng.integers(0, 6, (n, L)), 2, 40) + 33).astype(np.uint8)
    paths = []
    for tag, arr in (("R1", fwd), ("R2", rev)):
        p = os.path.join(tmp, tag + ".fastq")
        with open(p, "wb") as f:
            f.write(b"".join(b"@p%d\n" % i + arr[i].tobytes() + b"\n+\n" + qual[i].tobytes() + b"\n" for i in range(n)))
        paths.append(p)
    return paths, fwd, rev, qual


def test_config3_contigs_on_cpu_then_gpu_filter(tmp_path, oracle):
    from moira_amd import contig as CT, fastio as F
    from moira_amd.engine import Engine
    n = 120_000
    paths, fwd, rev, qual = _write_pairs(str(tmp_path), n)
    total = kept = 0
    first = None
    with Engine(0) as eng:
        for fbuf, fidx, rbuf, ridx in F.PairedFastqChunks(open(paths[0], "rb"), open(paths[1], "rb"), 32768):
            cbuf, cidx, aux = CT.contigs_from_fastq(fbuf, fidx, rbuf, ridx, 33)
            q, lens, has_n = F.pack(cbuf, cidx, None, 33, 0, stride=608)
            assert lens.min() >= 300 and lens.max() <= 600
            r = eng.filter(q, lens=lens)
            ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=oracle.lib().pbo_max_threads())
            assert np.array_equal(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
            if first is None:
                first = (q[:200].copy(), lens[:200].copy(), aux[:200].copy())
            total += len(lens)
            kept += r.n_pass
    assert total == n and 0.5 * n < kept < n
    # the same 200 contigs through the list-based entry (strings and integer lists, as the reference's make_contig has them)
    q200, l200, aux200 = first
    seqs, cq, clen, ov, gaps, mism = CT.contigs_batch(
        [fwd[i].tobytes().decode() for i in range(200)], [[int(v) - 33 for v in qual[i]] for i in range(200)],
        [rev[i].tobytes().decode() for i in range(200)], [[int(v) - 33 for v in qual[i]] for i in range(200)])
    assert np.array_equal(clen, l200) and np.array_equal(np.stack([ov, gaps, mism], 1), aux200)
    for i in range(200):
        want = oracle.pack_read(seqs[i], [int(v) for v in cq[i, :clen[i]]], 608)
        assert np.array_equal(q200[i], want)
