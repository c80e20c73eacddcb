"""Readers for the fixtures under tests/golden (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NPZ_SETS = ["rand_mixed", "rand_alpha05", "edge_alpha_1e-06", "edge_alpha_0.001", "edge_alpha_0.005",
            "edge_alpha_0.05", "edge_alpha_0.5", "edge_alpha_0.9", "synth300", "synth250", "synth_ragged",
            "long_reads"]      # long_reads: 1024 .. 4096 bases, up to 2169 DP rows, from the real reference (round 3)


def load_set(name):
    z = np.load(os.path.join(GOLDEN, name + ".npz"))
    return {k: z[k] for k in z.files}


def load_kat():
    return json.load(open(os.path.join(GOLDEN, "kat.json")))


def read_fasta_qual(prefix):
    """[(header, label, seq, quals)] from <prefix>.fasta + <prefix>.qual (moira's writer format,
    ref: moira/moira.py:953-961: '>header[\\tlabel]' then one line)."""
    fa = open(prefix + ".fasta").read().split("\n")
    qu = open(prefix + ".qual").read().split("\n")
    out = []
    for i in range(0, len(fa) - 1, 2):
        head = fa[i][1:].split("\t")
        assert qu[i] == fa[i]
        quals = [int(x) for x in qu[i + 1].split()]
        out.append((head[0], head[1] if len(head) > 1 else "", fa[i + 1], quals))
    return out


def expected_value(s):
    """The value every implementation must produce for set `s`:
    the C reference where it is defined, the Python twin where C is undefined."""
    exp = s["ee_ref"].copy()
    ub = s["ub"].astype(bool)
    exp[ub] = s["ee_py"][ub]
    return exp


def lut_fixture():
    """{q: (p, 1-p, p')} for q = 1..254 and the libm probes [(lam, j, exp(-lam), pow(lam, j))], decoded from the
    hex-float strings make_golden.py --lut wrote (values of the build container's glibc)."""
    kat = load_kat()
    lut = {int(q): tuple(float.fromhex(x) for x in v) for q, v in kat["lut"]["q"].items()}
    probes = [(float.fromhex(r[0]), int(r[1]), float.fromhex(r[2]), float.fromhex(r[3])) for r in kat["libm_probes"]["rows"]]
    return lut, probes


def bigq_fixture():
    """[(seq, quals, alpha, ee, ns, ub)]: reads with quality scores above 254 and the real reference's results
    (make_golden.py --bigq)."""
    import json
    d = json.load(open(os.path.join(GOLDEN, "bigq.json")))
    return [(r["seq"], r["quals"], r["alpha"], float.fromhex(r["ee"]), r["ns"], r["ub"]) for r in d["reads"]]


def bigq_poisson_fixture():
    """[(seq, quals, alpha, ee or None, ns)]: the same reads through the reference's calculate_errors_poisson
    (None: the function raised OverflowError)."""
    import json
    d = json.load(open(os.path.join(GOLDEN, "bigq.json")))
    return [(r["seq"], r["quals"], r["alpha"], None if r["poisson_ee"] is None else float.fromhex(r["poisson_ee"]), r["poisson_ns"])
            for r in d["reads"]]
