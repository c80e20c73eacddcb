#!/usr/bin/env python3
"""Generate tests/golden/*.npz and kat.json from the REAL reference.

Run in the build container only (needs /root/reference):
    python tests/golden/make_golden.py

What runs here
  * oracle/_ref/bernoulli.so -- moira/bernoullimodule.c compiled unmodified
    from where it lies (oracle/Makefile `ref`).  Every `ee_ref`/`ns_ref` below is
    its output.
  * moira/moira.py's Python twin `calculate_errors_PB` (Python-2 source): a
    lib2to3-converted copy is written to a temp dir OUTSIDE the repo, imported,
    used, and deleted.  It supplies `ee_py` for the cases where the C reference
    has undefined behaviour (first CDF row already above 1-alpha,
    moira/bernoullimodule.c:254) and for a cross-check subset.
  * moira/nw_align.pyx (the reference's Cython Needleman-Wunsch): cythonized (`cython -2`) and
    compiled with gcc INTO A TEMP DIR outside the repo, imported, used on >= 2,000 read pairs
    (overlapping, low-complexity / tie-heavy, unrelated, contained, unequal lengths, five
    match/mismatch/gap settings) and deleted -> nw_pairs.npz (inputs + aligned strings + score),
    plus moira.py's make_contig on those alignments in its three consensus modes -> nw_contigs.npz.
  * moira.py's calculate_errors_poisson on >= 2,000 reads incl. the region where it raises
    OverflowError -> poisson.npz.
  * `--long` (round 3): reads of 1024 .. 4096 bases -> long_reads.npz.  The C reference keeps its whole table on the
    stack (moira/bernoullimodule.c:214: 8 (L+1) L' bytes, 134 MB at 4096 bases), so this mode re-runs itself in a
    child process whose stack limit is raised first; the arithmetic is the reference's own, unmodified.  The Python
    twin is run on the reads it can finish (it makes J^2 L / 2 Python calls).
  * `--lut` (round 3): the Phred -> probability table and a set of libm probes (pow, exp) are added to kat.json as
    hex-float strings, so that a libm difference between the build container and a GPU box shows up under its own
    test name (last bullet).
  * `--bigq` (round 3): reads that carry quality scores above 254 (a .qual file or a Python caller can hold them; the
    byte matrix cannot) through the real extension and the Python twin -> bigq.json (sequence, integer scores, alpha,
    the reference's (ee, Ns) as hex floats).
Only data (inputs + the reference's outputs) is written into the repo.

The inputs are stored in the packed-qscore encoding of include/moira_pb.h
(0 = 'N', 255 = 'n', else Q).  The reference is called with the sequence
'A'/'N'/'n' per base and the integer scores.
"""
import json
import os
import shutil
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import pb_oracle as O  # noqa: E402

REF = "/root/reference/moira"


def load_python_reference():
    tmp = tempfile.mkdtemp(prefix="moira_py3_")
    dst = os.path.join(tmp, "moira_ref_py3.py")
    shutil.copy(os.path.join(REF, "moira.py"), dst)
    os.chmod(dst, 0o644)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", dst],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, tmp)
    import moira_ref_py3 as M
    return M, tmp


def load_cython_nw(tmp):
    """moira/nw_align.pyx -> nw_align.c -> nw_align.so, all inside `tmp` (outside the repo)."""
    import sysconfig
    c_file = os.path.join(tmp, "nw_align.c")
    subprocess.check_call([sys.executable, "-m", "cython", "-2", os.path.join(REF, "nw_align.pyx"), "-o", c_file],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    so = os.path.join(tmp, "nw_align" + sysconfig.get_config_var("EXT_SUFFIX"))
    subprocess.check_call(["gcc", "-O2", "-fPIC", "-shared", "-w", "-I", sysconfig.get_paths()["include"],
                           c_file, "-o", so])
    import importlib.util
    spec = importlib.util.spec_from_file_location("nw_align", so)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cat_strings(strs):
    """list[str] -> (uint8 buffer, int64 offsets[n+1])"""
    off = np.zeros(len(strs) + 1, np.int64)
    off[1:] = np.cumsum([len(x) for x in strs])
    return np.frombuffer("".join(strs).encode("ascii"), np.uint8).copy(), off


NW_PARAMS = [(1, -1, -2), (2, -3, -1), (1, 0, 0), (5, -4, -10), (1, -2, -1)]


def nw_pairs(rng):
    """(seq_1, seq_2) pairs: what nw_align sees in moira (forward read, reverse-complemented mate) and the
    shapes where a restatement goes wrong (ties, no overlap, containment, 1-base reads)."""
    def rnd(n, alphabet="ACGT"):
        return "".join(rng.choice(list(alphabet), n)) if n else ""

    def mutate(s, rate, alphabet="ACGTN"):
        s = list(s)
        for k in range(len(s)):
            if rng.random() < rate:
                s[k] = rng.choice(list(alphabet))
        for _ in range(int(rng.integers(0, 3))):
            if len(s) > 2 and rng.random() < 0.5:
                del s[int(rng.integers(0, len(s)))]
            elif rng.random() < 0.3:
                s.insert(int(rng.integers(0, len(s) + 1)), rng.choice(list("ACGT")))
        return "".join(s)

    pairs = [("A", "A"), ("A", "C"), ("ACGT", "ACGT"), ("AAAAAAAA", "AAAA"), ("ACGTACGTAC", "TTTT"),
             ("A", "ACGTACGT"), ("ACGTACGT", "T"), ("AC" * 20, "CA" * 20), ("A" * 50, "A" * 50), ("A" * 30, "A" * 47)]
    for _ in range(900):                      # overlapping pairs, as real paired reads are
        n1, n2 = int(rng.integers(20, 260)), int(rng.integers(20, 260))
        k = int(rng.integers(1, min(n1, n2) + 1))
        a = rnd(n1)
        b = a[n1 - k:] + rnd(n2 - k)
        pairs.append((mutate(a, 0.01), mutate(b, rng.choice([0.0, 0.02, 0.1]))))
    for _ in range(500):                      # low complexity: two-letter alphabets, homopolymer runs, tandem repeats
        kind = rng.integers(0, 3)
        n1, n2 = int(rng.integers(5, 200)), int(rng.integers(5, 200))
        if kind == 0:
            a, b = rnd(n1, "AC"), rnd(n2, "AC")
            if rng.random() < 0.5:
                k = int(rng.integers(1, min(n1, n2) + 1))
                b = a[n1 - k:] + rnd(n2 - k, "AC")
        elif kind == 1:
            unit = rnd(int(rng.integers(1, 5)))
            a, b = (unit * 200)[:n1], (unit * 200)[int(rng.integers(0, 4)):][:n2]
            a, b = mutate(a, 0.03, "ACGT"), mutate(b, 0.03, "ACGT")
        else:
            a = "".join(c * int(rng.integers(1, 9)) for c in rnd(40))[:n1]
            b = "".join(c * int(rng.integers(1, 9)) for c in rnd(40))[:n2]
        pairs.append((a, b))
    for _ in range(350):                      # unrelated reads (no overlap at all)
        pairs.append((rnd(int(rng.integers(1, 300))), rnd(int(rng.integers(1, 300)))))
    for _ in range(300):                      # one read contained in the other / very unequal lengths
        n1 = int(rng.integers(60, 330))
        a = rnd(n1)
        k = int(rng.integers(1, 40))
        s = int(rng.integers(0, n1 - k + 1))
        b = mutate(a[s:s + k], 0.03)
        pairs.append((a, b) if rng.random() < 0.5 else (b, a))
    return [(a, b) for a, b in pairs if a and b]


def make_nw_fixtures(pyref, tmp):
    nw = load_cython_nw(tmp)
    # the reference's own KAT first (moira/test/test_moira.py:53-55)
    rng = np.random.default_rng(20161004)
    pairs = nw_pairs(rng)
    s1, s2, a1, a2, sc, prm = [], [], [], [], [], []
    for k, (x, y) in enumerate(pairs):
        for pi in ([0] if k % 3 else range(len(NW_PARAMS))):      # every third pair under all five settings
            m, mm, g = NW_PARAMS[pi]
            r1, r2, score = nw.nw_align(x, y, m, mm, g)
            s1.append(x); s2.append(y); a1.append(r1); a2.append(r2); sc.append(score); prm.append(pi)
    b1, o1 = cat_strings(s1); b2, o2 = cat_strings(s2); c1, p1 = cat_strings(a1); c2, p2 = cat_strings(a2)
    out = os.path.join(HERE, "nw_pairs.npz")
    np.savez_compressed(out, seq1=b1, off1=o1, seq2=b2, off2=o2, aln1=c1, aoff1=p1, aln2=c2, aoff2=p2,
                        score=np.array(sc, np.int64), param=np.array(prm, np.int8),
                        params=np.array(NW_PARAMS, np.int32))
    print("nw_pairs       n=%5d alignments of %d pairs (moira/nw_align.pyx, cythonized in a temp dir) -> %s"
          % (len(sc), len(pairs), os.path.relpath(out, ROOT)))
    # make_contig (moira/moira.py:1376-1558) on the reference's alignment of the default-parameter cases
    modes = [("best", 40, False), ("sum", 0, False), ("posterior", 40, False), ("best", 40, True), ("sum", 60, True)]
    rows = []
    for k in range(0, len(sc), 4):
        if prm[k] != 0 or len(s1[k]) < 2:
            continue
        q1 = [int(v) for v in rng.integers(2, 42, len(s1[k]))]
        q2 = [int(v) for v in rng.integers(2, 42, len(s2[k]))]
        for mi, (mode, cap, trim) in enumerate(modes):
            try:
                cs, cq, ov, gaps, mism = pyref.make_contig(a1[k], list(q1), a2[k], list(q2), 20, 6, mode, cap, trim)
            except Exception:                                       # the reference raises on some degenerate alignments
                continue
            rows.append((k, mi, q1, q2, cs, [int(v) for v in cq], int(ov), int(gaps), int(mism)))
    cs_b, cs_o = cat_strings([r[4] for r in rows])
    flat = lambda idx: (np.concatenate([np.asarray(r[idx], np.int16) for r in rows]),
                        np.concatenate([[0], np.cumsum([len(r[idx]) for r in rows])]).astype(np.int64))
    q1b, q1o = flat(2); q2b, q2o = flat(3); cqb, cqo = flat(5)
    out = os.path.join(HERE, "nw_contigs.npz")
    np.savez_compressed(out, pair=np.array([r[0] for r in rows], np.int32), mode=np.array([r[1] for r in rows], np.int8),
                        q1=q1b, q1off=q1o, q2=q2b, q2off=q2o, contig=cs_b, coff=cs_o, cq=cqb, cqoff=cqo,
                        stats=np.array([r[6:9] for r in rows], np.int32),
                        modes=np.array([m[0] for m in modes]), caps=np.array([m[1] for m in modes], np.int32),
                        trims=np.array([m[2] for m in modes], np.bool_), insert=np.int32(20), deltaq=np.int32(6))
    print("nw_contigs     n=%5d contigs (moira.py make_contig, %d modes) -> %s" % (len(rows), len(modes), os.path.relpath(out, ROOT)))


def make_poisson_fixture(pyref):
    """moira/moira.py:1637-1679 on packed reads (0 = 'N'; the Python reference scores 'n' as a base, so no 255)."""
    rng = np.random.default_rng(20161005)
    q, lens = rand_mixed(rng, 1800, 352)
    q[q == 255] = 17
    # long low-quality reads: Lambda up to several hundred -> factorial(171) / float pow overflow region
    ql = np.zeros((500, 608), np.uint8)
    ll = rng.integers(100, 600, 500).astype(np.int32)
    for i in range(500):
        hi = int(rng.integers(2, 9))
        ql[i, :ll[i]] = rng.integers(1, hi + 1, ll[i])
        if rng.random() < 0.2:
            ql[i, rng.integers(0, ll[i], 5)] = 0
    alphas = (0.005, 0.05, 1e-4)
    sets = []
    for mat, ln in ((q, lens), (ql, ll)):
        n = mat.shape[0]
        ee = np.full((len(alphas), n), np.nan)
        ns = np.zeros(n, np.int32)
        ovf = np.zeros((len(alphas), n), np.uint8)
        for i in range(n):
            seq, quals = unpack(mat[i], int(ln[i]))
            for ai, a in enumerate(alphas):
                try:
                    e, s = pyref.calculate_errors_poisson(seq, quals, a)
                    ee[ai, i] = e
                    ns[i] = s
                except OverflowError:
                    ovf[ai, i] = 1
                    ns[i] = seq.count("N")
        sets.append((mat, ln, ee, ns, ovf))
    out = os.path.join(HERE, "poisson.npz")
    np.savez_compressed(out, alphas=np.array(alphas), q_a=sets[0][0], lens_a=sets[0][1], ee_a=sets[0][2], ns_a=sets[0][3],
                        ovf_a=sets[0][4], q_b=sets[1][0], lens_b=sets[1][1], ee_b=sets[1][2], ns_b=sets[1][3], ovf_b=sets[1][4])
    print("poisson        n=%5d reads x %d alphas (overflow cases: %d) -> %s"
          % (len(lens) + len(ll), len(alphas), int(sets[0][4].sum() + sets[1][4].sum()), os.path.relpath(out, ROOT)))


def unpack(row, n):
    """packed bytes -> (seq, quals) as the reference wants them."""
    seq = "".join("N" if v == 0 else "n" if v == 255 else "A" for v in row[:n])
    quals = [20 if v in (0, 255) else int(v) for v in row[:n]]
    return seq, quals


def run_set(name, q, lens, alpha, ref, pyref, py_every=0, py_mask=None):
    n = q.shape[0]
    ee_ref = np.full(n, np.nan)
    ns_ref = np.zeros(n, np.int32)
    ee_py = np.full(n, np.nan)
    ub = np.zeros(n, np.uint8)
    # rows==1 (CDF crosses on the first row) or no scored base: the C result is undefined / trivial
    _, _, _, rows = O.filter_batch(q, lens=lens, alpha=alpha, threads=8)
    for i in range(n):
        seq, quals = unpack(q[i], int(lens[i]))
        e, s = ref.calculate_errors_PB(seq, quals, alpha)
        ns_ref[i] = s
        if rows[i] == 1:
            ub[i] = 1            # C reads accumulated_probs[-1]; value is garbage, do not pin it
        else:
            ee_ref[i] = e
        if ub[i] or (py_every and i % py_every == 0) or (py_mask is not None and py_mask[i]):
            # Python twin counts only 'N' (moira.py:1605); feed it the sequence with n->N so both
            # references see the same ambiguous set (the C semantics, which the build follows)
            ep, sp = pyref.calculate_errors_PB(seq.replace("n", "N"), quals, alpha)
            ee_py[i] = ep
            assert sp == s, (name, i, sp, s)
    out = os.path.join(HERE, name + ".npz")
    np.savez_compressed(out, q=q, lens=lens.astype(np.int32), alpha=np.float64(alpha),
                        ee_ref=ee_ref, ns_ref=ns_ref, ee_py=ee_py, ub=ub)
    both = ~np.isnan(ee_ref) & ~np.isnan(ee_py)
    assert np.array_equal(ee_ref[both], ee_py[both]), "C and Python references disagree in " + name
    print("%-14s n=%5d  ub=%4d  py-checked=%5d  ee range %.3g..%.3g  -> %s"
          % (name, n, ub.sum(), (~np.isnan(ee_py)).sum(), np.nanmin(ee_ref), np.nanmax(ee_ref),
             os.path.relpath(out, ROOT)))


def rand_mixed(rng, n, stride):
    q = np.zeros((n, stride), np.uint8)
    lens = rng.integers(1, stride - 1, n).astype(np.int32)
    for i in range(n):
        L = lens[i]
        kind = rng.integers(0, 5)
        if kind == 0:      # uniform junk
            row = rng.integers(1, 42, L)
        elif kind == 1:    # high quality with a decaying tail
            row = np.clip(38 - (np.arange(L) / max(L, 1)) ** 2 * rng.integers(0, 36) - rng.integers(0, 6, L), 1, 41)
        elif kind == 2:    # very high quality (short ones hit the UB case)
            row = rng.integers(30, 42, L)
        elif kind == 3:    # wide range incl. > 41 (fasta+qual inputs can exceed 41)
            row = rng.integers(1, 94, L)
        else:              # low quality
            row = rng.integers(1, 12, L)
        row = row.astype(np.uint8)
        amb = rng.random(L) < 0.02
        row[amb] = np.where(rng.random(amb.sum()) < 0.7, 0, 255)
        q[i, :L] = row
    return q, lens


def edge_cases(kat_q):
    rows, alphas = [], []

    def add(vals, alpha=0.005):
        rows.append(np.asarray(vals, np.uint8)); alphas.append(alpha)
    for L in (1, 2, 5, 10, 20, 40, 50):
        add([40] * L)
    add([0] * 30)                       # all N
    add([255] * 7)                      # all n
    add([0, 255, 30, 30, 0, 12, 255, 40, 2, 2])
    add([1] * 4)                        # 'ACGT',[0]*4 after the Q0->1 clamp
    add([1] * 50)
    add([2] * 300)
    add([93] * 300)
    add([41] * 120)
    add([254] * 10)
    add([3] * 600)
    add(list(range(1, 255)))
    for a in (0.001, 0.005, 0.05, 0.5, 0.9, 1e-6):
        add(kat_q, a)
    for L in (1, 2, 3):
        for qv in (1, 2, 3, 10, 20, 23, 24, 30):
            add([qv] * L)
    return rows, alphas


LONG_SPECS = [
    # (length, q_lo, q_hi_exclusive, copies, also through the Python twin)
    (1024, 30, 41, 2, True), (1025, 30, 41, 2, True), (1500, 25, 41, 2, True), (2500, 20, 41, 2, True),
    (4096, 30, 41, 2, True), (1500, 8, 20, 1, True), (1100, 12, 30, 1, True),
    (1100, 1, 3, 1, False), (1500, 1, 3, 2, False), (2500, 1, 4, 1, False), (2000, 2, 6, 1, False),
    (4096, 2, 5, 1, False), (3000, 10, 25, 1, False),
]


def make_long_fixture(ref, pyref):
    """Reads longer than the 1023 bases rounds 1-2 covered, through the REAL reference: rows needed from 3 to > 2000
    (more than 1024 rows = more than one wave of the HIP path)."""
    rng = np.random.default_rng(20161006)
    rows, use_py = [], []
    for L, lo, hi, copies, py in LONG_SPECS:
        for c in range(copies):
            r = rng.integers(lo, hi, L).astype(np.uint8)
            if c == 1:                                   # ambiguous bases, also beyond column 960
                r[rng.integers(0, L, 6)] = 0
                r[rng.integers(960, L, 2)] = 255
            rows.append(r)
            use_py.append(py)
    stride = 4096
    q = np.zeros((len(rows), stride), np.uint8)
    lens = np.zeros(len(rows), np.int32)
    for i, r in enumerate(rows):
        q[i, :len(r)] = r
        lens[i] = len(r)
    run_set("long_reads", q, lens, 0.005, ref, pyref, py_mask=np.array(use_py))
    _, _, _, need = O.filter_batch(q, lens=lens, alpha=0.005, threads=8)
    print("long_reads     rows needed: min %d, max %d, > 1024 rows: %d reads" % (need.min(), need.max(), int((need > 1024).sum())))


def make_bigq_fixture(ref, pyref):
    """Quality scores above 254: the reference takes any int (moira/bernoullimodule.c:92-108, moira/moira.py:1561-1634)."""
    rng = np.random.default_rng(20161008)
    big = [255, 256, 300, 999, 1000, 3000, 3239, 3240, 3241, 5000, 65535, 2 ** 31 - 1]
    reads = []

    def add(seq, quals, alpha):
        quals = [int(v) for v in quals]
        e, s = ref.calculate_errors_PB(seq, quals, alpha)
        ep, sp = pyref.calculate_errors_PB(seq.replace("n", "N"), [v if v else 1 for v in quals], alpha)
        assert sp == s
        _, _, rows = O.ee_rowwise(seq, quals, alpha)
        ub = int(rows == 1)                      # the C reference reads accumulated_probs[-1] there: the twin's value is pinned
        if not ub:
            assert e == ep, (seq, quals, alpha, e, ep)
        # the Poisson approximation of the same read (moira/moira.py:1637-1679; only 'N' is skipped there)
        try:
            pe, pn = pyref.calculate_errors_poisson(seq, quals, alpha)
            pe = float(pe).hex()
        except OverflowError:
            pe, pn = None, seq.count("N")
        reads.append({"seq": seq, "quals": quals, "alpha": alpha, "ee": float(ep).hex(), "ns": int(s), "ub": ub,
                      "poisson_ee": pe, "poisson_ns": int(pn)})

    for k in range(90):
        L = int(rng.integers(1, 220))
        lo, hi = [(2, 42), (1, 8), (25, 42), (1, 255)][k % 4]
        quals = rng.integers(lo, hi, L).tolist()
        for pos in rng.integers(0, L, int(rng.integers(1, max(2, L // 3)))):
            quals[int(pos)] = int(rng.choice(big))
        seq = ["A"] * L
        if k % 3 == 0:
            for pos in rng.integers(0, L, 1 + L // 40):
                seq[int(pos)] = "N" if rng.random() < 0.7 else "n"
        if k % 7 == 0:
            quals[int(rng.integers(0, L))] = 0
        add("".join(seq), quals, float([0.005, 0.05, 1e-5, 0.3][k % 4]))
    add("A" * 40, [300] * 40, 0.005)                                            # nothing but big scores
    add("A" * 300, [255 + 3 * i for i in range(150)] + [7] * 150, 0.005)       # 150 distinct big values
    add("A" * 500, list(range(1, 201)) + [1000 + i for i in range(54)] + [3] * 246, 0.005)    # 254 distinct values in all
    add("ANnA", [5000, 5000, 5000, 2], 0.005)
    add("A", [2 ** 31 - 1], 0.5)
    add("ACGTNACGT", [0, 300, 0, 12, 0, 0, 40, 1000, 7], 0.05)                 # Q0: clamped by the extension, p = 1 for the Poisson function
    add("A" * 300, [1] * 280 + [400] * 20, 0.005)                               # Lambda ~ 222: the Poisson function overflows
    out = os.path.join(HERE, "bigq.json")
    json.dump({"source": "oracle/_ref/bernoulli.so (moira/bernoullimodule.c unmodified) and the Python twin of moira/moira.py:1561-1634; "
                         "ub = 1: the C reference's value is undefined (bernoullimodule.c:254), the twin's is stored",
               "reads": reads}, open(out, "w"), indent=0)
    print("bigq.json: %d reads, %d of them ub" % (len(reads), sum(r["ub"] for r in reads)))


def add_lut_to_kat():
    """Phred -> {p, 1-p, p'} exactly as moira/bernoullimodule.c:202,140-145 evaluate them with THIS container's libm,
    plus libm probes of the Poisson tail (moira/moira.py:1671: exp(-Lambda) * Lambda**j / factorial(j)), as hex floats."""
    import math
    path = os.path.join(HERE, "kat.json")
    kat = json.load(open(path))
    a, b = O.lut()
    lut = {}
    for qv in range(1, 255):
        p = math.pow(10, qv / -10.0)
        assert math.pow(1 - p, 1) == a[qv]
        lut[str(qv)] = [float(p).hex(), float(a[qv]).hex(), float(b[qv]).hex()]
    rng = np.random.default_rng(20161007)
    probes = []
    for lam in [0.01, 0.5, 1.0, 2.4, 6.932519986616133, 17.25, 40.0, 88.8, 150.0, 300.5, 700.0] + \
            [float(x) for x in rng.uniform(0.001, 400.0, 29)]:
        for j in (1, 2, 7, 33, 120, 170):
            try:
                pw = math.pow(lam, j)
            except OverflowError:
                pw = math.inf
            probes.append([float(lam).hex(), j, float(math.exp(-lam)).hex(), float(pw).hex()])
    kat["lut"] = {"source": "pow(10, q / -10.0), pow(1 - p, 1), ((1-1+1)/(1.0*1)) * (p/(1-p)) * pow(1-p, 1): "
                            "moira/bernoullimodule.c:202,140-145; glibc of the build container", "q": lut}
    kat["libm_probes"] = {"source": "exp(-lam), pow(lam, j) as moira/moira.py:1671 calls them", "rows": probes}
    json.dump(kat, open(path, "w"), indent=0)
    print("kat.json: lut (%d scores) and %d libm probes added" % (len(lut), len(probes)))


def main():
    O.build()
    if "--lut" in sys.argv:
        add_lut_to_kat()
        return
    if "--long" in sys.argv and not os.environ.get("MAKE_GOLDEN_BIG_STACK"):
        import resource

        def big_stack():
            resource.setrlimit(resource.RLIMIT_STACK, (resource.RLIM_INFINITY, resource.RLIM_INFINITY))
        sys.exit(subprocess.call([sys.executable, os.path.abspath(__file__), "--long"], preexec_fn=big_stack,
                                 env=dict(os.environ, MAKE_GOLDEN_BIG_STACK="1")))
    ref = O.reference_module()
    assert ref is not None, "oracle/_ref/bernoulli.so missing: make -C oracle ref"
    pyref, tmp = load_python_reference()
    if "--bigq" in sys.argv:
        try:
            make_bigq_fixture(ref, pyref)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    if "--long" in sys.argv:
        try:
            make_long_fixture(ref, pyref)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    if "--only-new" in sys.argv:          # nw_*.npz and poisson.npz only (the PB sets are already committed)
        try:
            make_nw_fixtures(pyref, tmp)
            make_poisson_fixture(pyref)
        finally:
            shutil.rmtree(tmp, ignore_errors=True)
        return
    try:
        # ---- known-answer tests of the reference's own test-suite (moira/test/test_moira.py) ----
        ns = {}
        src = open(os.path.join(REF, "test", "test_moira.py")).read()
        start = src.index("testSeq1 = ")
        end = src.index("args = Arguments(")
        exec(src[start:end], ns)       # data literals only (sequences, quals, expected tuples)
        fwd = ns["test_ForwardProcess"]
        prd = ns["test_PairedProcess"]
        kat = {
            "source": "moira/test/test_moira.py:40,43,45,118-128",
            "kat1": {"seq": ns["testSeq1"], "quals": ns["testQual1"], "alpha": 0.005,
                     "ee": 6.446879136706666, "ns": 0, "poisson_ee": 6.932519986616133},
            "kat2_forward_truncate200": {"seq": fwd[1], "quals": fwd[2], "alpha": 0.005,
                                         "ee_plus_ns": fwd[3]},
            "kat3_paired_truncate200": {"seq": prd[1], "quals": prd[2], "alpha": 0.005,
                                        "ee_plus_ns": prd[3], "overlap_gaps_mismatches": list(prd[4:7])},
        }
        kat["contig"] = {
            "source": "moira/test/test_moira.py:50-59,119-126",
            "seq2": ns["testSeq2"], "qual2": ns["testQual2"],
            "rc2": [ns["testRC2"][0], ns["testRC2"][1]], "aligned": list(ns["test_aligned"]),
            "contig": list(ns["test_contig"]),
            "args": {"match": 1, "mismatch": -1, "gap": -2, "insert": 20, "deltaq": 6,
                     "consensus_qscore": "best", "qscore_cap": 40, "trim_overlap": False}}
        assert fwd[3] == 0.9685179556745876 and prd[3] == 0.9643903629780557
        for k in ("kat1",):
            assert ref.calculate_errors_PB(kat[k]["seq"], kat[k]["quals"], 0.005) == (kat[k]["ee"], 0)
        json.dump(kat, open(os.path.join(HERE, "kat.json"), "w"), indent=0)
        print("kat.json written")

        rng = np.random.default_rng(20161003)
        q, lens = rand_mixed(rng, 3000, 352)
        run_set("rand_mixed", q, lens, 0.005, ref, pyref, py_every=10)
        q, lens = rand_mixed(rng, 600, 352)
        run_set("rand_alpha05", q, lens, 0.05, ref, pyref, py_every=10)

        rows, alphas = edge_cases(np.asarray(ns["testQual1"], np.uint8))
        for a in sorted(set(alphas)):
            sel = [r for r, al in zip(rows, alphas) if al == a]
            stride = 16 * ((max(len(r) for r in sel) + 15) // 16)
            qq = np.zeros((len(sel), stride), np.uint8)
            ll = np.zeros(len(sel), np.int32)
            for i, r in enumerate(sel):
                qq[i, :len(r)] = r
                ll[i] = len(r)
            run_set("edge_alpha_%g" % a, qq, ll, a, ref, pyref, py_every=1)

        q, lens = O.synth_fill(4096, 320, fixed_len=300, seed=2)
        run_set("synth300", q, lens, 0.005, ref, pyref, py_every=64)
        q, lens = O.synth_fill(1000, 256, fixed_len=250, seed=1)
        run_set("synth250", q, lens, 0.005, ref, pyref, py_every=50)
        q, lens = O.synth_fill(2048, 608, min_len=50, max_len=600, seed=5)
        run_set("synth_ragged", q, lens, 0.005, ref, pyref, py_every=64)
        make_nw_fixtures(pyref, tmp)
        make_poisson_fixture(pyref)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
