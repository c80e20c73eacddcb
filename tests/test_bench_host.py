"""Host-side logic of bench.py that runs before anything touches a GPU: the plain `python bench.py --gpus N` command must
start `torch.distributed.run` as a CHILD process (never exec, never after a GPU call) with the same arguments, and hand
its exit code on; under torchrun (WORLD_SIZE set) it must not start anything."""
import importlib.util
import json
import os
import subprocess
import sys
import time

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_plain_command_starts_its_ranks_as_a_child(monkeypatch):
    bench = load_bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    class Probe:                                                    # the device-count pre-flight (a throw-away child)
        stdout = "8\n"

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: Probe())
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5", "--launch", "processes"])
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                        # the child's exit code is handed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-9:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5", "--launch", "processes"]
    assert seen["env"].get("HSA_ENABLE_IPC_MODE_LEGACY") == "0" or os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY")
    if not torch_loaded_before:
        assert "torch" not in sys.modules                           # nothing GPU-capable was imported in the parent


def test_world_size_mismatch_is_an_error_not_a_launch(monkeypatch):
    bench = load_bench()
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch under torchrun"))
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4"])
    with pytest.raises(SystemExit, match="does not match"):
        bench.main()


def test_usable_cpus_is_positive_and_bounded():
    bench = load_bench()
    n = bench.usable_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    assert bench.CONFIG2_READS == 10_000_000 and bench.CONFIG4_SHARD * 8 == 1_000_000_000


def test_preflight_refuses_more_ranks_than_gpus(monkeypatch, capsys):
    """`--gpus 8` on a node that shows fewer GPUs must say so and return non-zero without starting
    anything (no rendezvous to time out in)."""
    bench = load_bench()

    class Probe:
        stdout = "4\n"

    monkeypatch.setattr(bench.subprocess, "run", lambda *a, **k: Probe())
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: pytest.fail("must not launch"))
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8"])
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 2
    assert "only 4 GPU" in capsys.readouterr().err


def test_step_plan_is_plain_arithmetic_on_the_rank_uniform_time():
    """`steps` and `settle` used to be decided from a rank-LOCAL step time, with a collective inside
    the branch -- at 48.5 ms/step x 20 steps = 0.97 s a 3 % slower rank took the other branch.  Now one unconditional
    all-reduce(MAX) makes the time rank-uniform and this pure function decides; same input, same output on every rank."""
    bench = load_bench()
    assert bench.plan_steps(0.0485, 20, 5) == (20, 4)                # 0.97 s < 1 s: settle to 0.5 s of back-to-back work
    assert bench.plan_steps(0.0500, 20, 5) == (20, 0)                # exactly 1 s: no settle -- on EVERY rank
    assert bench.plan_steps(0.0038, 20, 5) == (20, 125)
    steps, settle = bench.plan_steps(0.0038, 0, 3)                   # auto: >= 1 s of timed work
    assert steps * 0.0038 >= 1.0 and settle == 0
    assert bench.plan_steps(10.0, 0, 3)[0] == 10 and bench.plan_steps(1e-9, 0, 3)[0] == 4000
    import inspect
    body = inspect.getsource(bench.plan_steps)
    assert "all_reduce" not in body and "allmax" not in body           # no collective can hide in a branch here


def _run_bench(args, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    t = time.time()
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=env, cwd=ROOT)
    return p, time.time() - t


def test_eight_rank_rehearsal_on_cpu_with_unequal_ranks():
    """The launch shape the driver's 8-GPU run uses, at world size 8, on a box with no GPU: the plain command starts
    torchrun as a child, picks a port, every rank goes through the same collectives (gloo) although rank r's stub
    step is (1 + 4 r) ms long -- i.e. the ranks measure very different local step times -- and rank 0 prints one line
    with one rate and one device string per rank."""
    p, _ = _run_bench(["--gpus", "8", "--rehearse-on-cpu", "--steps", "6", "--warmup", "1", "--reads", "1000", "--_rank-delay-ms", "4"])
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 8 and line["steps"] == 6 and line["config"]["world_size"] == 8
    assert len(line["reads_per_s_per_rank"]) == 8 and len(line["devices"]) == 8
    assert all(d.startswith("rank %d:" % r) for r, d in enumerate(line["devices"]))
    rates = line["reads_per_s_per_rank"]
    assert rates[0] > 2 * rates[7] > 0                               # per-rank rates show the straggler
    assert line["t_step_rank_uniform_s"] >= 0.029                    # the slowest rank's step: 1 + 7 * 4 ms
    assert line["settle_steps"] > 0                                  # 6 x 29 ms < 1 s: every rank settled, the same count
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 8000
    assert "REHEARSAL" in line["config"]["workload"]


def test_a_dead_rank_fails_the_job_promptly():
    """A rank that dies before the timed region: the plain command must come back non-zero well inside the collective
    timeout, not hang (the driver gives the run 600 s)."""
    p, took = _run_bench(["--gpus", "3", "--rehearse-on-cpu", "--steps", "3", "--warmup", "1", "--reads", "1000", "--_die-rank", "1"])
    assert p.returncode != 0
    assert took < 150, took
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]


def test_rccl_failure_falls_back_to_gloo_in_the_same_process():
    """if the RCCL communicator does not come up the ranks must carry the 24-byte totals over gloo IN THE
    SAME PROCESS and still print the line.  Without a GPU the attempt fails on every rank (no device to put the
    communicator on): three ranks, the vote, the fallback, the totals, the exit without tearing down a half-made group."""
    p, took = _run_bench(["--gpus", "3", "--rehearse-on-cpu", "--steps", "3", "--warmup", "1", "--reads", "1000", "--_try-rccl"])
    assert p.returncode == 0, p.stderr[-3000:]
    line = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert line["outcome"]["pass"] + line["outcome"]["fail"] == 3000
    assert "RCCL not used" in line["config"]["collective_backend"]
    assert took < 120


def test_config3_extras_host_stages_run_without_a_gpu():
    """extras.config3_paired (round 4): the host stages (in-memory FASTQ text -> index -> contigs -> pack) and the
    pipelining are exercised here with a stand-in for the filter call; the GPU part is the driver's bench run."""
    import types
    import bench

    class NoGpu:
        calls = 0

        def filter(self, q, lens=None, fixed_len=None):
            NoGpu.calls += 1
            assert q.dtype.name == "uint8" and (lens is None or len(lens) == q.shape[0])
            return types.SimpleNamespace(n_pass=0 if lens is None else int((lens >= 300).sum()))
    r = bench.config3_paired_rate(NoGpu(), pairs_per_chunk=20_000, chunks=3)
    assert r["pairs"] == 60_000 and r["chunks"] == 3 and r["contigs_kept"] == 60_000
    assert 440 <= r["mean_contig_length"] <= 460                       # 2 x 300 bases, 150 of them shared
    assert NoGpu.calls == 1 + 1 + 3 + 3                                # warm-up of the slots, untimed chunk, two passes
    assert r["pipelined"]["pairs_per_s"] > 0 and r["projected_wall_s_for_100M_pairs"] > 0
    assert set(r["stage_pairs_per_s"]) == {"index_both_files", "contig_construction", "pack", "gpu_filter_incl_pcie"}


def test_a_failed_process_launch_is_retried_as_threads_in_a_fresh_child():
    """`python bench.py --gpus N` (launch auto).  The process job dies before printing anything (here: a launcher
    that exits 7 at once) -> the SAME measurement is started once more as `--launch threads`, in another fresh child; the parent
    never touches the GPU.  Without a GPU that second child can only say so -- which is what shows that it was started."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    env["BENCH_FAKE_LAUNCHER_FAILURE"] = "1"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--reads", "1000",
                        "--steps", "2", "--warmup", "1", "--no-extras"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert "fake launcher failure" in p.stderr
    assert "ended with code 7 before it printed a result" in p.stderr and "--launch threads" in p.stderr
    import moira_amd._lib as L
    if L.load().mpb_device_count() == 0:
        assert p.returncode != 0 and "needs an MI355X" in p.stderr          # the threads child ran, and has no GPU here
    # with --launch processes nothing is retried
    p2 = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--rehearse-on-one-gpu", "--reads", "1000",
                         "--launch", "processes"], capture_output=True, text=True, timeout=300, env=env, cwd=ROOT)
    assert p2.returncode == 7 and "--launch threads" not in p2.stderr


def test_held_clock_is_read_from_the_line_with_the_asterisk():
    bench = load_bench()
    assert bench.ClockSampler.parse("0: 500Mhz \n1: 2396Mhz *\n2: 2400Mhz\n") == 2396
    assert bench.ClockSampler.parse("S: 94Mhz *\n0: 500Mhz\n1: 2400Mhz\n") == 94
    assert bench.ClockSampler.parse("0: 500Mhz\n") is None
    s = bench.ClockSampler(None).start()
    assert s.stop() is None                                  # no device path: nothing sampled, nothing raised
    assert bench._decode_rank('{"k":{"dp":3.1},"mhz":[2010,1990,2100]}') == {
        "kernels_ms_per_step": {"dp": 3.1}, "held_clock": {"mean_mhz": 2010, "min_mhz": 1990, "max_mhz": 2100}}


def test_the_headline_is_out_before_the_extras_and_survives_a_kill():
    """bench.py prints the complete headline and flushes BEFORE any extra runs, and the full line again as the last
    line.  On the CPU rehearsal path (N = 1, stub step, stub extras that sleep): (a) killing the process inside the extras leaves a
    parseable headline on stdout; (b) left alone, the last line carries `extras` and the first one does not."""
    import signal
    env = dict(os.environ, PYTHONUNBUFFERED="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--rehearse-on-cpu", "--steps", "3", "--warmup", "1",
           "--reads", "1000", "--_stub-extras"]
    p = subprocess.Popen(cmd + ["60"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, cwd=ROOT)
    try:
        first = p.stdout.readline()                       # blocks until the headline is flushed; the "extras" then sleep 3 x 60 s
        assert first.startswith("{"), first
        head = json.loads(first)
        assert head["metric"].startswith("reads/sec") and head["value"] > 0 and "extras" not in head
        assert p.poll() is None                           # still inside the extras
        p.send_signal(signal.SIGKILL)
        rest = p.stdout.read()
        assert rest.strip() == ""                         # nothing else had been printed: the headline is all there is, and it parses
    finally:
        if p.poll() is None:
            p.kill()
        p.wait(timeout=30)
    q = subprocess.run(cmd + ["0.05"], capture_output=True, text=True, timeout=120, env=env, cwd=ROOT)
    assert q.returncode == 0, q.stderr[-2000:]
    lines = [l for l in q.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 2 and "extras" not in json.loads(lines[0])
    last = json.loads(lines[-1])
    assert sorted(last["extras"]) == ["stub_0", "stub_1", "stub_2"]
    assert {k: v for k, v in last.items() if k != "extras"} == json.loads(lines[0])


def test_bench_py_stays_small_and_the_extras_live_in_tools():
    n = len(open(os.path.join(ROOT, "bench.py")).read().splitlines())
    assert n < 600, n
    bench = load_bench()
    assert callable(bench.config3_paired_rate) and callable(bench.real_profile_batches)      # forwarded to tools/bench_extras.py
    with pytest.raises(AttributeError):
        bench.no_such_extra


def test_masked_share_reproduces_a_hand_made_tiling():
    """tools/bench_extras.masked_share: reads of one class are ordered by length bin (stable) and cut into tiles of 64 / G; a tile
    runs as long as its longest read."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extras as X
    caps = np.array([2] * 65 + [160] * 5)                 # G = 1: tiles of 64 reads; G = 16: tiles of 4
    lens = np.array([100] * 64 + [10] + [300, 280, 290, 310, 100])
    budget, issued, per_class = X.masked_share(caps, lens)
    assert budget == 2 * (6400 + 10) + 160 * 1280
    # class 2: one full tile of length-100 reads (bin 1) -- but the length-10 read (bin 0) sorts FIRST: tiles [10, 100 x 63], [100]
    # class 160: bins 100 -> 1, 280..310 -> 4: order [100, 300, 280, 290 | 310]: tiles of 4
    assert issued == 2 * 64 * (100 + 100) + 160 * 4 * (300 + 310)
    assert abs(per_class[2] - (1 - 2 * 6410 / (2 * 64 * 200))) < 1e-12


def _write_counter_csv(path, rows):
    import csv
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Correlation_Id", "Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value", "Start_Timestamp", "End_Timestamp"])
        for r in rows:
            w.writerow(r)


def test_live_counter_csvs_become_bytes_per_launch(tmp_path):
    """what bench.py does with the CSVs of its own rocprofv3 --pmc child passes -- counters averaged over a kernel's
    dispatches, reads = 32 / 64 / 128-byte requests x their sizes, writes = 64-byte requests x 64 + the others x 32
    (MI355X_MICROARCH.md, HBM)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extras as X
    dp = "void (anonymous namespace)::k_dp<false, false>(DpArgs)"
    rs = "void (anonymous namespace)::k_narrow_rs<2, true>(unsigned char const*, long)"
    rd_rows, wr_rows, sq_rows = [], [], []
    for d, (r32, r64, r128, wr, wr64) in enumerate([(100, 10, 1000, 300, 200), (300, 30, 3000, 500, 400)]):
        for name, val in (("TCC_EA0_RDREQ_sum", r32 + r64 + r128), ("TCC_EA0_RDREQ_32B_sum", r32), ("TCC_EA0_RDREQ_64B_sum", r64),
                          ("TCC_EA0_RDREQ_128B_sum", r128)):
            rd_rows.append([d, d, dp, name, val, 1000, 4000])
        for name, val in (("TCC_EA0_WRREQ_sum", wr), ("TCC_EA0_WRREQ_64B_sum", wr64)):
            wr_rows.append([d, d, dp, name, val, 1000, 4000])
        for name, val in (("SQ_INSTS_VALU", 512 * 2000), ("GRBM_GUI_ACTIVE", 8 * 4000)):
            sq_rows.append([d, d, dp, name, val, 0, 2000])                     # 2 us: 4000 cycles = 2 GHz
    rd_rows.append([9, 9, rs, "TCC_EA0_RDREQ_128B_sum", 7, 0, 10])
    rd_rows.append([9, 9, "some_other_kernel(int)", "TCC_EA0_RDREQ_128B_sum", 1, 0, 10])
    _write_counter_csv(str(tmp_path / "tccrd" / "host" / "1_counter_collection.csv"), rd_rows)
    _write_counter_csv(str(tmp_path / "tccwr" / "host" / "2_counter_collection.csv"), wr_rows)
    _write_counter_csv(str(tmp_path / "sq" / "host" / "3_counter_collection.csv"), sq_rows)
    k = X.parse_pmc_csvs(str(tmp_path))
    assert set(k) == {"k_dp<false, false>", "k_narrow_rs<2, true>", "some_other_kernel"}
    v = k["k_dp<false, false>"]
    assert v["TCC_EA0_RDREQ_128B_sum"] == 2000 and v["duration_us_tccrd"] == 3.0 and v["duration_us_sq"] == 2.0
    rd, wr = X.pmc_bytes(v)
    assert rd == 32 * 200 + 64 * 20 + 128 * 2000
    assert wr == 64 * 300 + 32 * (400 - 300)
    assert X.pmc_bytes(k["k_narrow_rs<2, true>"]) == (128 * 7, None)           # no write pass seen for it: None, never a guess


def test_a_failing_counter_pass_is_an_error_entry_not_an_exception(monkeypatch, tmp_path):
    """Any failure of the live passes leaves {'error': ...}: bench.py then keeps the committed profiles/ values and says so
    (`traffic_source`)."""
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import bench_extras as X
    fake = tmp_path / "rocprofv3"
    fake.write_text("#!/bin/sh\necho 'no counters today' >&2\nexit 3\n")
    fake.chmod(0o755)
    monkeypatch.setenv("PATH", str(tmp_path) + os.pathsep + os.environ.get("PATH", ""))
    r = X.pmc_live(1000, 300, 1, 320, deadline_s=20.0)
    assert set(r) == {"error"} and "exit code 3" in r["error"] and "no counters today" in r["error"]
    slow = tmp_path / "slow"
    slow.mkdir()
    (slow / "rocprofv3").write_text("#!/bin/sh\nsleep 30\n")
    (slow / "rocprofv3").chmod(0o755)
    monkeypatch.setenv("PATH", str(slow) + os.pathsep + os.environ.get("PATH", ""))
    r = X.pmc_live(1000, 300, 1, 320, deadline_s=6.0)
    assert set(r) == {"error"} and "deadline" in r["error"]
