"""Host-side logic of bench.py that runs before anything touches a GPU: the plain `python bench.py --gpus N` command must
start `torch.distributed.run` as a CHILD process (never exec, never after a GPU call) with the same arguments, and hand
its exit code on; under torchrun (WORLD_SIZE set) it must not start anything."""
import importlib.util
import os
import sys

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

    monkeypatch.setattr(bench.subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "8", "--steps", "20", "--warmup", "5"])
    torch_loaded_before = "torch" in sys.modules
    with pytest.raises(SystemExit) as e:
        bench.main()
    assert e.value.code == 7                                        # the child's exit code is handed on
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and int(cmd[cmd.index("--master-port") + 1]) > 0
    assert cmd[-7:] == [os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "20", "--warmup", "5"]
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
