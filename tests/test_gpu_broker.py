"""The per-read drop-in under `moira.py --processors P` (moira/moira.py:398-399,431-454): P worker processes call
bernoulli.calculate_errors_PB per read; ONE broker process owns the GPU and micro-batches what they have pending
(moira_amd/broker.py, moira_amd/csrc/mpb_broker.cpp).  Bit-exact against the oracle from 8 concurrent processes, scores
above 254 and reads of more than 1024 DP rows included; the broker is started by whichever worker asks first."""
import multiprocessing as mp
import os
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
        n = int(rng.integers(1, 420))
        lo, hi = [(2, 41), (20, 41), (1, 8), (30, 42), (2, 94)][int(rng.integers(0, 5))]
        q = [int(v) for v in rng.integers(lo, hi, n)]
        s = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, n)].copy()
        amb = rng.random(n) < 0.02
        s[amb] = np.where(rng.random(int(amb.sum())) < 0.7, ord("N"), ord("n"))
        if k % 41 == 7:
            q[int(rng.integers(0, n))] = int(rng.choice([255, 300, 1000, 5000]))       # its own code table: run alone
        if k % 53 == 11:
            q[int(rng.integers(0, n))] = 0                                             # Q0 -> 1
        out.append((s.tobytes().decode(), q, float([0.005, 0.005, 0.05, 1e-4][int(rng.integers(0, 4))])))
    out.append(("A" * 1500, [1 + (i % 3) for i in range(1500)], 0.005))                # > 1024 DP rows: run alone (k_wide)
    return out


def _worker(name, seed, count, rounds, out):
    sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle")]
    import pb_oracle as O
    from moira_amd import broker
    cl = broker.client(0, name=name, idle_exit=5.0)
    reads = _reads(seed, count)
    want = [O.ee_rowwise(s, q, a)[:2] for s, q, a in reads]
    bad = []
    t0 = time.perf_counter()
    for r in range(rounds):
        for k, ((s, q, a), w) in enumerate(zip(reads, want)):
            got = cl.calculate_errors_PB(s, q, a)
            if got != w and len(bad) < 5:
                bad.append((k, got, w))
    dt = time.perf_counter() - t0
    cl.close()
    out.put((len(bad), bad, len(reads) * rounds, dt))


@pytest.mark.parametrize("form", ["direct", "copies", "lanes"])
def test_eight_concurrent_processes_are_bit_exact(oracle, monkeypatch, form):
    """The three serving forms of the broker: the resident server (k_serve: a wave per slot, no launch per call) reading the
    workers' shared-memory slots themselves (the segment registered with the runtime; the default) or the broker thread's
    copies of them (MPB_BROKER_DIRECT=0), and the launch-per-micro-batch lanes (MPB_BROKER_SERVER=0)."""
    from moira_amd import broker
    server = "0" if form == "lanes" else "1"
    monkeypatch.setenv("MPB_BROKER_SERVER", server)        # read by the broker process, which inherits a worker's environment
    monkeypatch.setenv("MPB_BROKER_DIRECT", "0" if form == "copies" else "1")
    name = "gputest%s_%d" % (form, os.getpid())
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(name, 500 + k, 120, 6, out)) for k in range(8)]
    t_start = time.time()
    for p in procs:
        p.start()
    res = [out.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(60)
    st = broker.stats(name)
    broker.shutdown(name)
    assert all(r[0] == 0 for r in res), [r[1] for r in res if r[0]]
    total = sum(r[2] for r in res)
    assert st is not None and st["served"] == total and st["pid"] > 0
    assert st["solo"] >= 8 * 6 and st["batches"] <= total                   # launches; fewer than reads: they shared launches
    if server == "1":                                                       # the resident server: a launch per lifetime (100 ms), not per call
        assert st["batches"] <= 12 * (time.time() - t_start) + 20, (st, time.time() - t_start)
    assert broker.stats(name) is None


def test_the_resident_server_leaves_and_comes_back(oracle):
    """k_serve's grid always drains: its waves leave when their lifetime (100 ms) is over, whatever the host does.  While calls
    keep coming the broker launches it again at once; after a pause the next call does.  Reads of 2047 bases go through the
    mailbox, longer ones (and scores above 254) through the ordinary per-read path; a shutdown with the server resident
    returns promptly."""
    from moira_amd import broker
    name = "gpuserve_%d" % os.getpid()
    cl = broker.client(0, name=name, idle_exit=5.0)
    rng = np.random.default_rng(5)
    reads = []
    for n in (1, 16, 300, 301, 1023, 2047, 2048, 3000):
        reads.append(("A" * n, [int(v) for v in rng.integers(25, 41, n)], 0.005))
    reads.append(("ACGT" * 10, [300] * 40, 0.05))
    want = [oracle.ee_rowwise(s, q, a)[:2] for s, q, a in reads]
    t_end = time.time() + 0.35                                  # a stream of calls across three lifetimes
    k = 0
    while time.time() < t_end:
        i = k % len(reads)
        assert cl.calculate_errors_PB(*reads[i]) == want[i], i
        k += 1
    st1 = broker.stats(name)
    assert st1["served"] == k and st1["batches"] >= 3, st1   # launched again while the calls kept coming
    time.sleep(0.4)                                             # the server has left and nobody called: it stays away ...
    st2 = broker.stats(name)
    assert st2["batches"] <= st1["batches"] + 1, (st1, st2)
    for i in range(len(reads)):                                 # ... until the next call
        assert cl.calculate_errors_PB(*reads[i]) == want[i], i
    st3 = broker.stats(name)
    assert st3["batches"] >= st2["batches"] + 1 and st3["solo"] >= 3 * 2, (st2, st3)
    cl.close()
    t0 = time.time()
    broker.shutdown(name)
    assert time.time() - t0 < 5.0 and broker.stats(name) is None


def _pool_task(args):
    s, q, a = args
    sys.path.insert(0, os.path.join(ROOT, "moira_amd", "dropin"))
    import bernoulli
    return bernoulli.calculate_errors_PB(s, q, a), os.getpid()


def test_the_dropin_module_uses_the_broker_in_pool_workers(oracle, monkeypatch):
    """What an unchanged moira.py does: Pool(P).apply_async(process_data ...) -> bernoulli.calculate_errors_PB in the
    workers.  The workers find (start) the broker by themselves; the parent process never touches it."""
    from moira_amd import broker
    name = "gpupool_%d" % os.getpid()
    monkeypatch.setenv("MOIRA_PB_BROKER_NAME", name)
    monkeypatch.setenv("PYTHONPATH", ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    reads = _reads(77, 200)[:200]
    want = [oracle.ee_rowwise(s, q, a)[:2] for s, q, a in reads]
    with mp.get_context("fork").Pool(4) as pool:           # fork, as Python 2's Pool did
        got = pool.map(_pool_task, reads, chunksize=5)
    assert [g[0] for g in got] == want
    assert len({g[1] for g in got}) > 1                    # several workers took part
    st = broker.stats(name)
    assert st is not None and st["served"] >= len(reads)
    broker.shutdown(name)
    # and in the parent (not a multiprocessing child) the module keeps a context of its own: no broker is started
    sys.path.insert(0, os.path.join(ROOT, "moira_amd", "dropin"))
    import bernoulli
    assert bernoulli.calculate_errors_PB("ACGT", [0] * 4, 0.005) == (3.987440567842452, 0)
    assert broker.stats(name) is None


def test_the_python2_form_of_the_dropin(oracle):
    """moira_amd/dropin/py2/bernoulli.py (ctypes + stdlib only, Python 2 and 3 syntax): the reference's known answers
    directly, and -- from the workers of a Pool -- through the broker it starts itself."""
    import importlib.util
    import subprocess
    import golden_io as G
    from moira_amd import broker
    path = os.path.join(ROOT, "moira_amd", "dropin", "py2", "bernoulli.py")
    spec = importlib.util.spec_from_file_location("bernoulli_py2form", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    kat = G.load_kat()["kat1"]
    assert mod.calculate_errors_PB(kat["seq"], kat["quals"], kat["alpha"]) == (kat["ee"], kat["ns"])
    assert mod.calculate_errors_PB("ACGT", [0] * 4, 0.005) == (3.987440567842452, 0)
    assert mod.calculate_errors_PB(b"ACNT", [30, 300, 7, 5000], 0.05) == oracle.ee_rowwise("ACNT", [30, 300, 7, 5000], 0.05)[:2]
    name = "gpupy2_%d" % os.getpid()
    code = (
        "import sys, multiprocessing as mp\n"
        "sys.path.insert(0, %r)\n"
        "import bernoulli\n"
        "def f(a): return bernoulli.calculate_errors_PB(*a)\n"
        "if __name__ == '__main__':\n"
        "    reads = [('ACGT' * 20, [2 + (i * 7) %% 39 for i in range(80)], 0.005)] * 50 + [('A' * 10, [40] * 10, 0.005)]\n"
        "    pool = mp.Pool(3)\n"
        "    print(repr(pool.map(f, reads, 3)))\n"
        "    pool.close(); pool.join()\n" % os.path.dirname(path))
    env = dict(os.environ, MOIRA_PB_BROKER_NAME=name)
    env.pop("MOIRA_PB_BROKER", None)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-1500:]
    got = eval(r.stdout.strip().splitlines()[-1])
    want = [oracle.ee_rowwise("ACGT" * 20, [2 + (i * 7) % 39 for i in range(80)], 0.005)[:2]] * 50 + [(0.0, 0)]
    assert got == want
    st = broker.stats(name)
    assert st is not None and st["served"] >= 51
    broker.shutdown(name)


def test_a_direct_slot_cannot_pick_the_kernels_modes(oracle):
    """in direct serving the resident kernel reads a request's 64 bytes of parameters from the CLIENT-writable slot; the
    broker thread's checks do not cover that path.  The wave takes only alpha's threshold and the prediction constants from them:
    forged opt-in flags, an ambiguity mode, a limit change nothing (the answer is the honest call's, bit for bit), and a threshold
    outside (0, 1) is answered "ask the host" (pass = 2), never obeyed."""
    import ctypes as C
    import struct
    from moira_amd import broker
    name = "gputestforge_%d" % os.getpid()
    cl = broker.client(0, name=name, idle_exit=5.0)
    try:
        rng = np.random.default_rng(5)
        s = "".join("ACGT"[int(v)] for v in rng.integers(0, 4, 300))
        s = s[:50] + "N" + s[51:]                                        # an 'N': forged ambiguity modes would show in ee / pass
        q = [int(v) for v in rng.integers(20, 41, 300)]
        want = oracle.ee_rowwise(s, q, 0.005)[:2]
        assert cl.calculate_errors_PB(s, q, 0.005) == want              # the kernel is up, the slot holds an honest request

        class Handle(C.Structure):                                       # the client handle starts with the Mapping {base, bytes}, then the slot index
            _fields_ = [("base", C.c_void_p), ("bytes", C.c_size_t), ("slot", C.c_int)]
        hd = Handle.from_address(cl.h.value)
        raw = (C.c_char * hd.bytes).from_address(hd.base)
        n_slots, slot_bytes = (int(v) for v in np.frombuffer(raw, np.int32, 2, 8))
        so = hd.bytes - n_slots * slot_bytes + hd.slot * slot_bytes
        door = np.frombuffer(raw, np.uint64, 1, so + 320)
        done = np.frombuffer(raw, np.uint32, 1, so + 384)
        d_ns = np.frombuffer(raw, np.int32, 1, so + 388)
        d_ee = np.frombuffer(raw, np.float64, 1, so + 392)
        d_pass = np.frombuffer(raw, np.uint8, 1, so + 400)
        prm = np.frombuffer(raw, np.uint8, 64, so + 448)
        honest = bytes(prm)
        thr, uncert, maxerr, z, zq, clow, mode, flags = struct.unpack_from("<dddfffiI", honest, 0)
        assert abs(thr - 0.995) < 1e-12 and flags == 0
        if int(door[0]) == 0 or int(done[0]) != int(door[0]) & 0xffffffff:
            pytest.skip("the broker is not serving the slots directly on this box (registration refused): nothing to forge")

        def forged_call(blob):
            prm[:] = np.frombuffer(blob, np.uint8)
            tok = (int(door[0]) & 0xffffffff) + 1
            door[0] = (300 << 32) | tok                                  # the row of the honest call still lies in the slot
            t0 = time.time()
            while int(done[0]) != tok and time.time() - t0 < 10:
                time.sleep(0.0005)
            assert int(done[0]) == tok
            return float(d_ee[0]), int(d_ns[0]), int(d_pass[0])
        # every opt-in flag, "disallow", a tiny uncert, a maxerrors: none of it may show
        blob = bytearray(honest)
        struct.pack_into("<dd", blob, 8, 1e-9, 1e-9)
        struct.pack_into("<iI", blob, 36, 2, 0xffffffff)
        ee, ns_, ps = forged_call(bytes(blob))
        assert (ee, ns_) == want and ps != 2
        for bad_thr in (2.0, 0.0, -1.0, float("nan")):
            blob = bytearray(honest)
            struct.pack_into("<d", blob, 0, bad_thr)
            assert forged_call(bytes(blob))[2] == 2                      # handed to the host, which checks alpha
        prm[:] = np.frombuffer(honest, np.uint8)
        assert cl.calculate_errors_PB(s, q, 0.005) == want              # the client's own next call: its token follows the forged ones
    finally:
        cl.close()
        broker.shutdown(name)
