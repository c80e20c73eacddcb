"""Readers for the fixtures under tests/golden (data only)."""
import json
import os

import numpy as np

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
NPZ_SETS = ["rand_mixed", "rand_alpha05", "edge_alpha_1e-06", "edge_alpha_0.001", "edge_alpha_0.005",
            "edge_alpha_0.05", "edge_alpha_0.5", "edge_alpha_0.9", "synth300", "synth250", "synth_ragged"]


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
