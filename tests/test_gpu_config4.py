"""BASELINE configs[3]: "1 B synthetic 300 bp reads sharded host-side across 8 x MI355X" -- what ONE GPU of
that job does: a 125 M-read shard (40 GB resident, generated on the device from the counter-based
generator, read ids first_read..).  Round 5: all EIGHT shards of the 1 B-read job run on the one GPU, one after the other;
shards 0 and 7 are compared with the oracle in full and 10 M contiguous reads of each of the others (the matrix comes back
from HBM 5 M rows at a time), next to the size-independent properties (determinism, pass count == sum of flags ==
threshold test on ee, no NaN) and windows regenerated on the host (the device generator wrote what the host generator
writes); and the host-side split itself (a shard == the same slice of the unsplit batch) is checked at a size the oracle
covers."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.fixture(scope="module")
def eng():
    from moira_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def test_config4_all_eight_shards_at_their_stated_size(eng, oracle):
    """BASELINE configs[3] at its stated size, on one GPU: the EIGHT 125 M-read shards of the 1 B-read job
    (read ids r * 125 M ..), generated and filtered one after the other in the same 40 GB of HBM.  Shards 0 and 7 are compared
    with the oracle in full, 10 M contiguous reads of each of the other six (310 M reads in all); every shard also gets the
    size-independent checks (pass count == sum of flags == threshold test on ee, no NaN) and host-regenerated windows
    (the device generator wrote what the host generator writes for THOSE read ids); shard 0 is run twice (deterministic)."""
    from test_gpu_parity import compare_every_read
    from conftest import note_parity
    n, stride, L, seed = 125_000_000, 320, 300, 2
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    compared, passed = 0, 0
    try:
        rng = np.random.default_rng(4)
        for rank in range(8):
            first = rank * n
            eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed, first_read=first)
            c1 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
            ee1, ps1, ns1 = d_ee.download(np.float64, n), d_pass.download(np.uint8, n), d_ns.download(np.int32, n)
            assert c1.n_reads == n and c1.n_pass == int(ps1.sum(dtype=np.int64)) and c1.n_fail == n - c1.n_pass
            assert not np.isnan(ee1).any()
            assert np.array_equal(ps1.astype(bool), ee1 <= L * 0.01)            # the predicate, recomputed
            passed += c1.n_pass
            if rank == 0:
                c2 = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
                assert same(d_ee.download(np.float64, n), ee1) and (c2.n_pass, c2.n_overflow) == (c1.n_pass, c1.n_overflow)
            if rank in (0, 7):
                row0, m = 0, n
            else:
                row0, m = int(rng.integers(0, n - 10_000_000)), 10_000_000
            compared += compare_every_read(eng, oracle, d_q, m, stride, ee1, ns1, ps1, fixed_len=L, step=5_000_000, row0=row0,
                                           label="config 4, shard %d of 8 (read ids %d ..), rows %d .. %d" % (rank, first, row0, row0 + m - 1))
            starts = rng.integers(0, n - 64, 12)
            starts[:2] = (0, n - 64)
            for start in starts:
                hq, _ = oracle.synth_fill(64, stride, fixed_len=L, seed=seed, first_read=first + int(start))
                assert np.array_equal(d_q.download(np.uint8, 64 * stride, offset=int(start) * stride).reshape(64, stride), hq)
            del ee1, ps1, ns1
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()
    assert compared == 2 * n + 6 * 10_000_000
    note_parity("[config 4 at its stated size] 8 shards x 125 M reads generated and filtered on one GPU; %d reads compared bit for bit "
                "with the oracle (shards 0 and 7 in full); %d of 1,000,000,000 reads pass" % (compared, passed))


def test_shards_equal_slices_of_the_unsplit_batch(eng, oracle):
    """The host-side split of config 4 at a size the oracle covers: W ranks, each generating and filtering ONLY
    its own range on the device (moira_amd.shard.filter_synth_shard), concatenated == the unsplit batch."""
    from moira_amd.shard import filter_synth_shard, shard_bounds
    n, stride, L, seed = 300_001, 320, 300, 2
    hq, _ = oracle.synth_fill(n, stride, fixed_len=L, seed=seed)
    ee, ns, ps, _ = oracle.filter_batch(hq, fixed_len=L, threads=oracle.lib().pbo_max_threads())
    for world in (1, 3, 8):
        parts = [filter_synth_shard(eng, n, world, r, L, stride, seed) for r in range(world)]
        assert [len(p[0]) for p in parts] == [hi - lo for lo, hi in (shard_bounds(n, world, r) for r in range(world))]
        assert same(np.concatenate([p[0] for p in parts]), ee)
        assert np.array_equal(np.concatenate([p[1] for p in parts]), ns)
        assert np.array_equal(np.concatenate([p[2] for p in parts]), ps.astype(bool))
        assert sum(p[3][0] for p in parts) == int(ps.sum())


def test_bench_starts_its_own_ranks_from_the_plain_command():
    """`python bench.py --gpus 2` without torchrun (the driver's command shape): the script starts its two ranks as a
    child job before touching the GPU and hands on their JSON line and exit code.  Rehearsal mode: both ranks share
    this box's one GPU and the 24-byte collectives go over gloo; small shards so that it takes seconds."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu",
                        "--reads", "300000", "--steps", "3", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["scaling"] == "weak"
    assert line["config"]["reads_per_gpu"] == 300000 and "configs[3]" in line["config"]["workload"]
    assert len(line["reads_per_s_per_rank"]) == 2 and all(v > 0 for v in line["reads_per_s_per_rank"])
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 600000
    assert line["value"] > 0 and line["roofline"]["frac"] > 0
    assert len(line["devices"]) == 2 and all("GPU 0" in d for d in line["devices"])
    assert "gloo" in line["config"]["collective_backend"]


def test_bench_rccl_selftest_on_this_gpu():
    """"RCCL has never executed".  On a one-GPU box it can only run with one rank, but that still loads
    librccl, creates the communicator under bench.py's deadline thread and does the 24-byte all-reduce on the GPU --
    the code an N > 1 run uses for its totals."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--rccl-selftest", "--reads", "300000", "--steps", "3",
                        "--warmup", "1", "--no-extras", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    st = line["rccl_selftest"]
    assert st["rccl_group"] == "ok" and st["all_reduce_3xint64_on_gpu"] == [1, 2, 3], st
    assert line["n_gpus"] == 1 and line["value"] > 0


def test_bench_rccl_failure_falls_back_to_gloo_on_hardware():
    """The failure an N > 1 run has to survive, provoked on real hardware: two ranks that share this box's one GPU try to
    build an RCCL communicator ("Duplicate GPU detected"); the ranks vote, carry the 24-byte totals over gloo in the same
    process, print the line and leave with rc 0 (never re-exec a process that has touched the GPU)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--_try-rccl",
                        "--reads", "300000", "--steps", "3", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert "RCCL not used" in line["config"]["collective_backend"]
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 600000 and line["n_gpus"] == 2


def test_bench_four_rank_rehearsal_on_one_gpu():
    """the widest rehearsal a 1-GPU box allows (the pool admits 6 processes on a card: four ranks, this
    test process and one spare), 2 M reads per rank: port selection, the build lock under four simultaneous imports,
    the rank-uniform step plan, per-rank device strings and rates, totals over all ranks.  The 8-rank shape itself is
    rehearsed without a GPU in tests/test_bench_host.py."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--rehearse-on-one-gpu",
                        "--reads", "2000000", "--steps", "4", "--warmup", "1", "--no-extras"],
                       capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 4 and line["steps"] == 4 and line["config"]["world_size"] == 4
    assert len(line["reads_per_s_per_rank"]) == 4 and all(v > 0 for v in line["reads_per_s_per_rank"])
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 8_000_000
    assert line["settle_steps"] > 0 and line["t_step_rank_uniform_s"] > 0
    assert "--gpus 1 --reads 2000000" in line["weak_scaling_anchor"]      # which N = 1 run anchors this curve


def test_bench_threads_mode_four_contexts_on_one_gpu():
    """the launcher-free N > 1 measurement -- N contexts on N host threads of ONE process, device-resident
    shards, no collective.  Four contexts share this box's one GPU; the line must carry what a reader of SCALE needs to spot
    a slow GPU: every thread's own rate, per-device kernel times and the clock the device held."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch", "threads", "--rehearse-on-one-gpu",
                        "--reads", "2000000", "--steps", "6", "--warmup", "1"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch"] == "threads" and line["n_gpus"] == 4 and line["steps"] == 6 and line["scaling"] == "weak"
    assert line["config"]["reads_per_gpu"] == 2000000 and "configs[3]" in line["config"]["workload"]
    assert len(line["reads_per_s_per_rank"]) == 4 and all(v > 0 for v in line["reads_per_s_per_rank"])
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 8_000_000
    assert len(line["per_rank"]) == 4
    for r in line["per_rank"]:
        assert r["kernels_ms_per_step"]["dp"] > 0 and r["kernels_ms_per_step"]["prepass"] > 0
        assert r["held_clock"] is None or 300 < r["held_clock"]["mean_mhz"] < 3000
    assert any(r["held_clock"] for r in line["per_rank"]), "no clock samples from sysfs"
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["fell_back_from"] is None
    # the four shards are the four read-id ranges of the process-per-GPU run: same totals
    q = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "4", "--launch", "processes", "--rehearse-on-one-gpu",
                        "--reads", "2000000", "--steps", "2", "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert q.returncode == 0, q.stderr[-2000:]
    line2 = json.loads([l for l in q.stdout.splitlines() if l.startswith("{")][-1])
    assert line2["outcome"] == line["outcome"] and line2["launch"] == "processes"
    assert len(line2["per_rank"]) == 4 and all(r["kernels_ms_per_step"]["dp"] > 0 for r in line2["per_rank"])


def test_bench_falls_back_to_threads_when_the_process_launch_fails():
    """... and `auto` takes it by itself when the process job ends non-zero before printing its line (here: a launcher that
    dies at once): a fresh child, the same arguments, the same JSON line with "launch": "threads"."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_FAKE_LAUNCHER_FAILURE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--reads", "300000",
                        "--steps", "3", "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=600, env=env, cwd=root)
    assert p.returncode == 0, p.stderr[-2000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["launch"] == "threads" and "exit code 7" in line["fell_back_from"] and line["n_gpus"] == 2
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 600000


def test_config2_size_order_invariance(eng):
    """A size-independent property at BASELINE config 2's full size, with no oracle in the loop: a read's result
    does not depend on where it sits in the batch.  Batch B holds the same 10 M reads as batch A rotated by 3.7 M
    positions (two device fills), so its sort / tiling differs everywhere; results must be A's, rotated -- bit for bit."""
    n, stride, L, seed, rot = 10_000_000, 320, 300, 2, 3_700_001
    d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    try:
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed)
        ca = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        ee_a, ns_a, ps_a = d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n)
        # B = reads rot..n-1 followed by 0..rot-1
        eng.synth_fill(d_q.ptr, n - rot, stride, fixed_len=L, seed=seed, first_read=rot)
        eng.synth_fill(d_q.ptr + (n - rot) * stride, rot, stride, fixed_len=L, seed=seed, first_read=0)
        cb = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass)
        ee_b, ns_b, ps_b = d_ee.download(np.float64, n), d_ns.download(np.int32, n), d_pass.download(np.uint8, n)
        assert same(ee_b, np.roll(ee_a, -rot)) and np.array_equal(ns_b, np.roll(ns_a, -rot))
        assert np.array_equal(ps_b, np.roll(ps_a, -rot))
        assert (ca.n_pass, ca.n_overflow) == (cb.n_pass, cb.n_overflow)
    finally:
        for b in (d_q, d_ee, d_ns, d_pass):
            b.free()
