"""CPU contig construction (libmoira_contig.so) against the reference's own KATs
(moira/test/test_moira.py:50-59) and, when the real reference is importable in the build
container, against moira.py's make_contig / nw_align on random pairs."""
import numpy as np
import pytest

import golden_io as G
from moira_amd import contig as CT


@pytest.fixture(scope="module")
def kat():
    return G.load_kat()


def test_reverse_complement_kat(kat):
    c = kat["contig"]
    seq, quals = CT.reverse_complement(c["seq2"], c["qual2"])
    assert [seq, quals] == c["rc2"]
    assert CT.reverse_complement("ACGTNWSRYMKBVDH-.") == ".-DHBVMKRYSWNACGT"
    with pytest.raises(ValueError, match="IUPAC"):
        CT.reverse_complement("ACGX")


def test_nw_align_kat(kat):
    c = kat["contig"]
    a = c["args"]
    rc = CT.reverse_complement(c["seq2"])
    got = CT.nw_align(kat["kat1"]["seq"], rc, a["match"], a["mismatch"], a["gap"])
    assert list(got) == c["aligned"] and got[2] == 13431


def test_make_contig_kat(kat):
    c = kat["contig"]
    a = c["args"]
    rc, rq = CT.reverse_complement(c["seq2"], c["qual2"])
    a1, a2, _ = CT.nw_align(kat["kat1"]["seq"], rc, a["match"], a["mismatch"], a["gap"])
    got = CT.make_contig(a1, kat["kat1"]["quals"], a2, rq, a["insert"], a["deltaq"],
                         a["consensus_qscore"], a["qscore_cap"], a["trim_overlap"])
    assert list(got) == c["contig"] and got[2:] == (162, 3, 8)


def test_batch_equals_single_and_kat3(kat):
    c = kat["contig"]
    seqs, cq, clen, ov, gaps, mism = CT.contigs_batch([kat["kat1"]["seq"]] * 5, [kat["kat1"]["quals"]] * 5,
                                                      [c["seq2"]] * 5, [c["qual2"]] * 5, threads=3)
    for i in range(5):
        assert seqs[i] == c["contig"][0] and list(cq[i, :clen[i]]) == c["contig"][1]
        assert (ov[i], gaps[i], mism[i]) == (162, 3, 8)
    # KAT-3 (test_PairedProcess): contig truncated to 200 is the filter's input
    k3 = kat["kat3_paired_truncate200"]
    assert seqs[0][:200] == k3["seq"] and list(cq[0, :200]) == k3["quals"]


def _random_pair(rng):
    L = int(rng.integers(30, 120))
    frag = "".join(rng.choice(list("ACGT"), int(L * 1.5)))
    f = list(frag[:L])
    r = list(frag[-L:])
    for s in (f, r):
        for _ in range(int(rng.integers(0, 6))):
            s[int(rng.integers(0, len(s)))] = rng.choice(list("ACGTN"))
        if rng.random() < 0.3:
            del s[int(rng.integers(0, len(s)))]
    f, r = "".join(f), "".join(r)
    rrc = CT.reverse_complement(r)
    return f, [int(x) for x in rng.integers(2, 41, len(f))], rrc, [int(x) for x in rng.integers(2, 41, len(rrc))]


def test_against_python_reference_when_available(tmp_path):
    import os
    import shutil
    import subprocess
    import sys
    ref = "/root/reference/moira/moira.py"
    if not os.path.exists(ref):
        pytest.skip("reference not present (GPU box)")
    dst = tmp_path / "moira_ref_py3.py"
    shutil.copy(ref, dst)
    os.chmod(dst, 0o644)
    subprocess.check_call([sys.executable, "-m", "lib2to3", "-w", "-n", str(dst)],
                          stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    sys.path.insert(0, str(tmp_path))
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            import moira_ref_py3 as M
    finally:
        sys.path.pop(0)
    rng = np.random.default_rng(11)
    for it in range(60):
        f, fq, r, rq = _random_pair(rng)
        for mode, cap, trim in (("best", 40, False), ("sum", 0, False), ("posterior", 40, False), ("best", 40, True)):
            rrc, rrq = M.reverse_complement(r, list(rq))
            assert CT.reverse_complement(r, rq) == (rrc, rrq)
            a1, a2, sc = CT.nw_align(f, rrc, 1, -1, -2)
            # the 2to3'd pure-Python nw_align compares ints with None (moira.py:1348); make_contig
            # on OUR alignment is compared instead, and the alignment itself is pinned by the KAT
            # and by the golden paired outputs
            want = M.make_contig(a1, list(fq), a2, list(rrq), 20, 6, mode, cap, trim)
            got = CT.make_contig(a1, fq, a2, rrq, 20, 6, mode, cap, trim)
            assert got == tuple(want), (it, mode)


def test_simd_walk_equals_scalar_loop():
    """mct_nw_align (anti-diagonals, 16-bit, SIMD) against mct_nw_align_scalar (the plain restatement of
    moira/nw_align.pyx:49-201): alignment strings and score, incl. empty/1-base reads, unequal lengths,
    tie-heavy low-complexity reads, other parameters, and the automatic switch for large parameters."""
    import numpy as np
    rng = np.random.default_rng(11)
    cases = [("", ""), ("A", ""), ("", "C"), ("A", "A"), ("A", "C"), ("ACGT", "ACGT"), ("AAAAAAAA", "AAAA"),
             ("ACGTACGTAC", "TTTT")]
    for _ in range(300):
        n1, n2 = int(rng.integers(1, 330)), int(rng.integers(1, 330))
        alphabet = "ACGT" if rng.random() < 0.7 else "AC"
        a = "".join(rng.choice(list(alphabet), n1))
        if rng.random() < 0.6:          # an overlapping pair, as real paired reads are
            k = int(rng.integers(1, min(n1, n2) + 1))
            b = a[n1 - k:] + "".join(rng.choice(list(alphabet), n2 - k))
            b = "".join(c if rng.random() > 0.03 else "T" for c in b)
        else:
            b = "".join(rng.choice(list(alphabet), n2))
        cases.append((a, b))
    for a, b in cases:
        for m, mm, g in ((1, -1, -2), (2, -3, -1), (1, 0, 0)):
            assert CT.nw_align(a, b, m, mm, g) == CT.nw_align(a, b, m, mm, g, scalar=True), (a, b, m, mm, g)
    a, b = cases[20]
    assert CT.nw_align(a, b, 500, -700, -900) == CT.nw_align(a, b, 500, -700, -900, scalar=True)   # 32-bit path


def _strings(buf, off):
    b = buf.tobytes().decode("ascii")
    return [b[off[i]:off[i + 1]] for i in range(len(off) - 1)]


def test_nw_align_against_reference_fixture():
    """tests/golden/nw_pairs.npz: 4,808 alignments produced by the REFERENCE's Cython aligner
    (moira/nw_align.pyx, cythonized and run by tests/golden/make_golden.py in the build container) on
    overlapping, low-complexity / tie-heavy, unrelated, contained and unequal-length pairs under five
    match/mismatch/gap settings.  Both walks of the library (anti-diagonal SIMD and row-by-row scalar) must
    reproduce the aligned strings and the score: this pins the tie-break order (nw_align.pyx:96-113) and the
    3' overlap fix (:155-201) to the reference, not to ourselves."""
    z = G.load_set("nw_pairs")
    s1, s2 = _strings(z["seq1"], z["off1"]), _strings(z["seq2"], z["off2"])
    a1, a2 = _strings(z["aln1"], z["aoff1"]), _strings(z["aln2"], z["aoff2"])
    assert len(s1) >= 2000 * 2 and len(set(zip(s1, s2))) >= 2000
    bad = []
    for k in range(len(s1)):
        m, mm, g = (int(v) for v in z["params"][z["param"][k]])
        want = (a1[k], a2[k], int(z["score"][k]))
        if CT.nw_align(s1[k], s2[k], m, mm, g) != want:
            bad.append(("simd", k))
        if k % 2 == 0 and CT.nw_align(s1[k], s2[k], m, mm, g, scalar=True) != want:
            bad.append(("scalar", k))
    assert not bad, bad[:10]


def test_make_contig_against_reference_fixture():
    """tests/golden/nw_contigs.npz: moira.py's make_contig (moira/moira.py:1376-1558) on the reference's
    own alignments, five consensus/cap/trim settings, random qualities."""
    z, c = G.load_set("nw_pairs"), G.load_set("nw_contigs")
    a1, a2 = _strings(z["aln1"], z["aoff1"]), _strings(z["aln2"], z["aoff2"])
    contigs = _strings(c["contig"], c["coff"])
    assert len(contigs) >= 2000
    for r in range(len(contigs)):
        k, mi = int(c["pair"][r]), int(c["mode"][r])
        q1 = [int(v) for v in c["q1"][c["q1off"][r]:c["q1off"][r + 1]]]
        q2 = [int(v) for v in c["q2"][c["q2off"][r]:c["q2off"][r + 1]]]
        got = CT.make_contig(a1[k], q1, a2[k], q2, int(c["insert"]), int(c["deltaq"]), str(c["modes"][mi]),
                             int(c["caps"][mi]), bool(c["trims"][mi]))
        want = (contigs[r], [int(v) for v in c["cq"][c["cqoff"][r]:c["cqoff"][r + 1]]]) + tuple(int(v) for v in c["stats"][r])
        assert got == want, (r, k, mi)


@pytest.mark.parametrize("isa", ["generic", "avx2", "avx512"])
def test_every_instruction_set_path_reproduces_the_reference_alignments(isa):
    """The fill of the score matrix exists once per instruction set (plain loop, AVX2: 16 cells per instruction, AVX-512:
    32; picked once per process, MOIRA_CONTIG_ISA forces one).  Each path, in a process of its own, must reproduce the
    reference's Cython alignments of tests/golden/nw_pairs.npz -- strings and scores -- exactly.  A path the CPU lacks
    falls back to the best it has (still compared)."""
    import os
    import subprocess
    import sys
    code = r"""
import sys
sys.path[:0] = [%r, %r]
import numpy as np, golden_io as G
from moira_amd import contig as CT
z = G.load_set("nw_pairs")
params = z["params"]
bad = 0
n = len(z["score"])
for k in range(0, n, 2):
    s1 = bytes(z["seq1"][z["off1"][k]:z["off1"][k + 1]]).decode(); s2 = bytes(z["seq2"][z["off2"][k]:z["off2"][k + 1]]).decode()
    m, mm, g = (int(v) for v in params[z["param"][k]])
    a1, a2, sc = CT.nw_align(s1, s2, m, mm, g)
    want1 = bytes(z["aln1"][z["aoff1"][k]:z["aoff1"][k + 1]]).decode(); want2 = bytes(z["aln2"][z["aoff2"][k]:z["aoff2"][k + 1]]).decode()
    bad += (a1, a2, sc) != (want1, want2, int(z["score"][k]))
print("checked", (n + 1) // 2, "bad", bad)
""" % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MOIRA_CONTIG_ISA=isa)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.strip().endswith("bad 0") and "checked 2404" in out.stdout, out.stdout


def test_nw_align_dropin_module_is_the_references(kat):
    """moira_amd/dropin/nw_align.py: what an unchanged moira.py gets from `import nw_align as nw` (moira/moira.py:241-245,
    :794) -- the reference's own known answer and a sample of the 4,808 alignments its Cython code produced."""
    import importlib.util
    import os
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "moira_amd", "dropin", "nw_align.py")
    spec = importlib.util.spec_from_file_location("nw_align_dropin", path)
    nw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nw)
    c, a = kat["contig"], kat["contig"]["args"]
    rc = CT.reverse_complement(c["seq2"])
    got = nw.nw_align(kat["kat1"]["seq"], rc, a["match"], a["mismatch"], a["gap"])
    assert list(got) == c["aligned"] and got[2] == 13431
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "nw_pairs.npz"))
    params = z["params"]
    for k in range(0, len(z["score"]), 37):
        s1 = z["seq1"][z["off1"][k]:z["off1"][k + 1]].tobytes().decode()
        s2 = z["seq2"][z["off2"][k]:z["off2"][k + 1]].tobytes().decode()
        a1 = z["aln1"][z["aoff1"][k]:z["aoff1"][k + 1]].tobytes().decode()
        a2 = z["aln2"][z["aoff2"][k]:z["aoff2"][k + 1]].tobytes().decode()
        m, mm, g = (int(v) for v in params[z["param"][k]])
        assert nw.nw_align(s1, s2, m, mm, g) == (a1, a2, int(z["score"][k]))
