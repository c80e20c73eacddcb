"""The per-read entry without a launch per call (round 5): while mpb_calculate_errors_PB is called from a process of its own,
the context keeps k_serve resident with one mailbox entry (moira_amd/csrc/mpb_api.cpp serve_one; the broker's form:
tests/test_gpu_broker.py).  Same results as the launch per call, whatever happens in between: pauses longer than the kernel's
lifetime, batch calls that grow (free and re-allocate) the workspace, reads the server does not take."""
import os
import subprocess
import sys
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _reads(seed, count):
    rng = np.random.default_rng(seed)
    out = []
    for k in range(count):
        n = int(rng.integers(1, 500))
        lo, hi = [(2, 41), (25, 41), (1, 8), (30, 42)][int(rng.integers(0, 4))]
        q = [int(v) for v in rng.integers(lo, hi, n)]
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        amb = rng.random(n) < 0.02
        s[amb] = np.where(rng.random(int(amb.sum())) < 0.7, ord("N"), ord("n"))
        out.append((s.tobytes().decode(), q, float([0.005, 0.05, 1e-4][int(rng.integers(0, 3))])))
    return out


def test_calls_pauses_batches_and_long_reads(oracle):
    from moira_amd.engine import Engine
    reads = _reads(3, 300)
    reads += [("A" * n, [30 + (i % 9) for i in range(n)], 0.005) for n in (2047, 2048, 5000)]       # the last two: not the server's
    reads += [("ACGT" * 10, [300] * 40, 0.05), ("A" * 1500, [1 + (i % 3) for i in range(1500)], 0.005)]   # a private table; > 1024 rows
    reads += [("", [], 0.005), ("N", [20], 0.005), ("n" * 20, [30] * 20, 0.005), ("A" * 16, [40] * 16, 0.9),   # empty, all-ambiguous,
              ("ACGT" * 75, [0] * 300, 0.005), ("A" * 300, [254] * 300, 1e-6)]                              # first-row crossing, Q0, Q254
    want = [oracle.ee_rowwise(s, q, a)[:2] for s, q, a in reads]
    with Engine(0) as eng:
        for i, r in enumerate(reads):
            assert eng.calculate_errors_PB(*r) == want[i], i
        time.sleep(0.25)                                   # the kernel has left (its lifetime is 100 ms): the next call launches it again
        for i, r in enumerate(reads[:50]):
            assert eng.calculate_errors_PB(*r) == want[i], i
        # batch calls in between: each larger than the last, so the workspace is freed and re-allocated with the server resident
        for n in (1000, 50_000, 400_000):
            q, _ = oracle.synth_fill(n, 320, fixed_len=300, seed=9, profile=0)
            res = eng.filter(q, fixed_len=300)
            ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8)
            assert np.array_equal(res.ee.view(np.uint64), ee.view(np.uint64)) and np.array_equal(res.ns, ns)
            for i, r in enumerate(reads[:20]):
                assert eng.calculate_errors_PB(*r) == want[i], (n, i)
        t0 = time.perf_counter()
        k = 0
        while time.perf_counter() - t0 < 0.35:             # a steady stream across three lifetimes
            i = k % 200
            assert eng.calculate_errors_PB(*reads[i]) == want[i], i
            k += 1
        assert k > 2000                                    # (>= 6,000 calls/s even on a loaded box; 5 x 10^4 alone)
    # the context is gone (its kernel was asked to leave first); a new one starts over
    with Engine(0) as eng:
        assert eng.calculate_errors_PB(*reads[0]) == want[0]


def test_the_launch_per_call_gives_the_same(oracle):
    """MPB_SERVE=0 (read when the context first serves a per-read call): the k_small launch per call."""
    code = ("import sys, json; sys.path.insert(0, %r)\n"
            "sys.path.insert(0, %r)\n"
            "from test_gpu_per_read_server import _reads\n"
            "from moira_amd.engine import Engine\n"
            "with Engine(0) as e: print(json.dumps([e.calculate_errors_PB(*r) for r in _reads(3, 120)]))\n"
            % (ROOT, os.path.join(ROOT, "tests")))
    outs = []
    for v in ("0", "1"):
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, cwd=ROOT,
                           env=dict(os.environ, MPB_SERVE=v))
        assert r.returncode == 0, r.stderr[-1500:]
        outs.append(r.stdout.strip().splitlines()[-1])
    assert outs[0] == outs[1]
    import json
    want = [list(oracle.ee_rowwise(s, q, a)[:2]) for s, q, a in _reads(3, 120)]
    assert json.loads(outs[0]) == want


def test_reference_vectors_read_by_read(oracle):
    """The reference's own results (tests/golden/*.npz: bernoullimodule.c through the real extension, its Python twin where C
    is undefined), every read of every set, ONE calculate_errors_PB call each: the resident server (registers-only body for
    reads of <= 1024 bases and <= 64 rows, the class bodies beyond, the ordinary path for >= 2048 bases or > 1024 rows)."""
    import golden_io as G
    from moira_amd.engine import Engine
    done = 0
    with Engine(0) as eng:
        for name in G.NPZ_SETS:
            s = G.load_set(name)
            q, lens, exp, alpha = s["q"], s["lens"], G.expected_value(s), float(s["alpha"])
            step = 1 if name != "long_reads" else 4                     # (the long reads: every fourth, they take milliseconds each)
            for i in range(0, len(lens), step):
                row = q[i, :int(lens[i])]
                seq = "".join("N" if b == 0 else "n" if b == 255 else "A" for b in row.tolist())
                quals = [30 if b in (0, 255) else int(b) for b in row.tolist()]
                ee, ns = eng.calculate_errors_PB(seq, quals, alpha)
                want = float(exp[i])
                assert (ee == want or (ee != ee and want != want)) and ns == int(s["ns_ref"][i]), (name, i, ee, want)
                done += 1
    assert done > 10_000
