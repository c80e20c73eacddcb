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


def unpack(row, n):
    """packed bytes -> (seq, quals) as the reference wants them."""
    seq = "".join("N" if v == 0 else "n" if v == 255 else "A" for v in row[:n])
    quals = [20 if v in (0, 255) else int(v) for v in row[:n]]
    return seq, quals


def run_set(name, q, lens, alpha, ref, pyref, py_every=0):
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
        if ub[i] or (py_every and i % py_every == 0):
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


def main():
    O.build()
    ref = O.reference_module()
    assert ref is not None, "oracle/_ref/bernoulli.so missing: make -C oracle ref"
    pyref, tmp = load_python_reference()
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
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
