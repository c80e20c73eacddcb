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


# ---------------------------------------------------------------------------------------------
# flag matrix (round 4): inputs derived from the reference's test1 / test2.fastq, the reference's outputs for them
# ---------------------------------------------------------------------------------------------
FLAG_DIR = os.path.join(GOLDEN, "flag_matrix")


def _lcg(state):
    """One step of a 64-bit linear congruential generator (written out so that the derived input never depends on a
    library's random stream): returns (new state, 31 random bits)."""
    state = (state * 6364136223846793005 + 1442695040888963407) & ((1 << 64) - 1)
    return state, state >> 33


def read_fastq_records(path):
    """[(header line, sequence, plus line, quality line)] of a plain / .gz / .bz2 FASTQ file."""
    import bz2
    import gzip
    opener = gzip.open if path.endswith(".gz") else bz2.open if path.endswith(".bz2") else open
    with opener(path, "rt") as f:
        lines = [l.rstrip("\n") for l in f]
    return [tuple(lines[i:i + 4]) for i in range(0, len(lines) - 3, 4)]


def derive_flag_inputs(rec1, rec2):
    """The flag matrix's input pair: the reference's test1 / test2 records with what the shipped files lack --
    ambiguous bases in both cases ('N' and 'n'), Q0 characters, short reads (so that --truncate discards and
    --min_overlap bites), header text after a blank / tab and ':' in names, and late duplicates of early reads with
    better qualities (so that a collapse group changes its representative).  Deterministic, index-driven."""
    out1, out2 = [], []
    st = 20161009
    n = min(len(rec1), len(rec2))
    for i in range(n):
        h1, s1, p1, q1 = rec1[i]
        h2, s2, p2, q2 = rec2[i]
        s1, q1, s2, q2 = list(s1), list(q1), list(s2), list(q2)
        if i % 7 == 3:
            st, k = _lcg(st)
            for _ in range(1 + k % 3):
                st, pos = _lcg(st)
                s1[pos % len(s1)] = "N"
        if i % 11 == 5:
            st, k = _lcg(st)
            for _ in range(1 + k % 2):
                st, pos = _lcg(st)
                s1[pos % len(s1)] = "n"
        if i % 9 == 4:
            st, k = _lcg(st)
            for _ in range(1 + k % 3):
                st, pos = _lcg(st)
                s2[pos % len(s2)] = "N"
        if i % 17 == 2:
            for _ in range(3):
                st, pos = _lcg(st)
                q1[pos % len(q1)] = "!"
        if i % 13 == 6:
            st, k = _lcg(st)
            cut = 100 + k % 100
            s1, q1 = s1[:cut], q1[:cut]
        if i % 19 == 8:
            st, k = _lcg(st)
            cut = 60 + k % 91
            s2, q2 = s2[:cut], q2[:cut]
        if i % 23 == 1:
            h1, h2 = h1 + " 1:N:0:7\tlane x", h2 + "\t2:N:0:7"
        if i % 29 == 9:
            h1 = h2 = "@run:7:" + h1[1:]
        out1.append((h1, "".join(s1), p1, "".join(q1)))
        out2.append((h2, "".join(s2), p2, "".join(q2)))
    for k in range(40):                         # late duplicates: same bases, better (even k) or worse (odd k) qualities
        i = (k * 37) % n
        h1, s1, p1, q1 = out1[i]
        h2, s2, p2, q2 = out2[i]
        bump = (lambda c: chr(min(ord(c) + 3, 73))) if k % 2 == 0 else (lambda c: chr(max(ord(c) - 4, 35)))
        out1.append(("@dup%d" % k, s1, "+", "".join(bump(c) for c in q1)))
        out2.append(("@dup%d" % k, s2, "+", "".join(bump(c) for c in q2)))
    return out1, out2


def write_fastq(path, records):
    with open(path, "w") as f:
        for r in records:
            f.write("%s\n%s\n%s\n%s\n" % r)


def write_fastq_offset64(path, records):
    """The same reads in the Illumina 1.3-1.7 encoding (quality characters + 31: --fastq_offset 64)."""
    with open(path, "w") as f:
        for h, s, p, q in records:
            f.write("%s\n%s\n%s\n%s\n" % (h, s, p, "".join(chr(ord(c) + 31) for c in q)))


def write_fastq_quirks(path, records):
    """The same reads as a file a tolerant parser still has to read as the reference does (moira/moira.py:1166-1175: every
    line is strip()ped, the header is cut at the first blank / tab, ALL leading '@' go, ':' becomes '_'): CRLF line ends,
    doubled '@', trailing blanks and tabs, the header repeated on the '+' line, a last line without a newline."""
    out = []
    for i, (h, s, p, q) in enumerate(records):
        if i % 5 == 1:
            h = "@" + h
        if i % 6 == 2:
            h = h + " \t"
        if i % 7 == 3:
            s = s + " "
        if i % 8 == 4:
            p = "+" + h.lstrip("@").split(" ")[0]
        if i % 9 == 5:
            q = q + "\t"
        out.append("\r\n".join((h, s, p, q)) if i % 2 else "\n".join((h, s, p, q)))
    with open(path, "w", newline="") as f:
        f.write("".join(r + ("\r\n" if k % 3 == 0 else "\n") for k, r in enumerate(out[:-1])) + out[-1])


def write_fasta_qual(fa_path, qu_path, records, offset=33):
    """The same reads as fasta + qual (the reference's second reader, moira/moira.py:1093-1149): header text after a blank /
    tab (which the reader must drop), and -- what no FASTQ file can hold -- a few scores above 93 and above 254."""
    with open(fa_path, "w") as fa, open(qu_path, "w") as qu:
        for i, (h, s, _, q) in enumerate(records):
            name = h[1:].replace("\t", " ").split(" ")[0]
            quals = [ord(c) - offset for c in q]
            if i % 37 == 5 and quals:
                quals[(i * 7) % len(quals)] = 300
            if i % 41 == 7 and quals:
                quals[(i * 11) % len(quals)] = 120
            fa.write(">%s some description %d\n%s\n" % (name, i, s))
            qu.write(">%s\tother text\n%s\n" % (name, " ".join(map(str, quals))))


def flag_manifest():
    return json.load(open(os.path.join(FLAG_DIR, "manifest.json")))


def flag_outputs():
    """{case: {file stem: bytes}} from the committed archive of the reference's outputs."""
    import io
    import tarfile
    out = {}
    with tarfile.open(os.path.join(FLAG_DIR, "outputs.tar.xz"), "r:xz") as tf:
        for m in tf.getmembers():
            if m.isfile():
                case, stem = m.name.split("/", 1)
                out.setdefault(case, {})[stem] = tf.extractfile(m).read()
    return out
