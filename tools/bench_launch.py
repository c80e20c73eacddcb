"""Launch shapes and the few collectives of bench.py (moved out of bench.py in round 6): `python bench.py --gpus N`
without torchrun (self_launch), N contexts on N threads of one process (threads_main), the gloo / RCCL collectives of an N > 1
run (Collectives), the CPU rehearsal stub, the held-clock sampler and the rank-uniform step plan.  bench.py imports these names;
the reference's only parallel axis is moira/moira.py:398-399,431-454 (`Pool(args.processors)`)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH_PY = os.path.join(ROOT, "bench.py")

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK = 39.3e12       # v_mul/add_f64 lane-ops per second: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz
CONFIG2_READS = 10_000_000     # BASELINE configs[1]
CONFIG4_SHARD = 125_000_000    # BASELINE configs[3]: 1 B reads / 8 GPUs


def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def visible_gpus():
    """Number of GPUs this process could use, WITHOUT initialising any of them (torch.cuda.device_count() does not
    create a context on this image; HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES are honoured by it)."""
    try:
        import torch
        return int(torch.cuda.device_count())
    except Exception:
        return 0


def self_launch(n, rehearsal, fall_back_to_threads=False):
    """`python bench.py --gpus N` without torchrun: start the ranks as a child job and hand its exit code on.
    Nothing in THIS process has initialised the GPU (no HIP call; at most a device count in a grandchild).
    fall_back_to_threads: when the process job ends non-zero WITHOUT having printed its JSON line (a rendezvous that never
    forms, a launcher that is not there), the same measurement is started once more as `--launch threads` -- in another
    fresh child: a process that has touched the GPU is never re-executed, and this one never touches it."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not rehearsal:
        # pre-flight in a throw-away child, so that this process never imports torch
        try:
            have = int(subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"],
                                      capture_output=True, text=True, timeout=300, env=env).stdout.strip().splitlines()[-1])
        except Exception:
            have = -1
        if 0 <= have < n:
            sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible on this node; nothing was started "
                             "(use --rehearse-on-one-gpu / --rehearse-on-cpu for a dry run)\n" % (n, have))
            return 2
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH_PY] + sys.argv[1:]
    if os.environ.get("BENCH_FAKE_LAUNCHER_FAILURE"):          # tests: a launcher that dies before any rank exists
        cmd = [sys.executable, "-c", "import sys; sys.stderr.write('fake launcher failure\\n'); sys.exit(7)"]
    if not fall_back_to_threads:
        return subprocess.call(cmd, env=env)
    # the child's stdout is passed through line by line; a line that parses as the result means the job got there
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    printed = False
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
        if line.startswith("{") and '"metric"' in line:
            printed = True
    rc = proc.wait()
    if rc == 0 or printed:
        return rc
    sys.stderr.write("bench.py: the %d-process job ended with code %d before it printed a result; running the same measurement "
                     "as N contexts on N threads of one fresh process (--launch threads)\n" % (n, rc))
    argv = [a for a in sys.argv[1:]]
    if "--launch" in argv:
        k = argv.index("--launch")
        del argv[k:k + 2]
    argv = [a for a in argv if not a.startswith("--launch=")]
    env["BENCH_FELL_BACK_FROM"] = "processes (exit code %d)" % rc
    return subprocess.call([sys.executable, BENCH_PY] + argv + ["--launch", "threads"], env=env)


COLLECTIVE_TIMEOUT_S = 120          # rendezvous and every gloo collective; the RCCL attempt has its own deadline below


RCCL_DEADLINE_S = 90


class Collectives:
    """The few, tiny collectives of an N > 1 run (see the module docstring).  world == 1: all no-ops."""

    def __init__(self, world, rank, local_rank, use_gpu, try_rccl, selftest=False):
        self.world, self.rank = world, rank
        self.totals_backend = None
        self.rccl_error = None
        self._rccl = None
        self._hung = False
        self.selftest = None
        if world == 1 and not selftest:
            return
        import datetime
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        if world == 1:
            # --rccl-selftest on one GPU: a one-rank group, so that the code below (gloo group, RCCL communicator created
            # under a deadline, a 24-byte all-reduce on the GPU, the vote) runs on real hardware at least once before
            # the driver's 8-GPU run -- it says nothing about xGMI, only that RCCL loads and initialises here
            dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%d" % _free_port(), rank=0, world_size=1,
                                    timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
            ok = self._try_rccl(local_rank)
            got = self.sum_totals_rccl((1, 2, 3)) if ok else None
            self.selftest = {"rccl_group": "ok" if ok else "failed: %s" % self.rccl_error,
                             "all_reduce_3xint64_on_gpu": got, "nccl_version": ".".join(str(v) for v in torch.cuda.nccl.version())
                             if hasattr(torch.cuda, "nccl") else None}
            self.world = 1
            self.close_selftest()
            return
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=COLLECTIVE_TIMEOUT_S))
        self.totals_backend = "gloo"
        if try_rccl:
            ok = self._try_rccl(local_rank)
            vote = torch.tensor([1 if ok else 0], dtype=torch.int32)
            dist.all_reduce(vote, op=dist.ReduceOp.MIN)             # every rank, unconditionally
            if int(vote.item()) == 1:
                self.totals_backend = "nccl"
            else:
                if ok:
                    self.rccl_error = "another rank's RCCL communicator did not come up"
                self._rccl = None
                # some rank's communicator is half-made: tearing the groups down could wait for it, AFTER the line is
                # out -- every rank leaves through os._exit once the closing barrier has been passed
                self._hung = True

    def _try_rccl(self, local_rank):
        """Create the RCCL group and run one 24-byte all-reduce on it, in a helper thread with a deadline: an exception
        or a hang on ANY rank turns into "use gloo" for all of them, never into a lost run."""
        import datetime
        import threading
        torch, dist = self.torch, self.dist
        box = {}

        def attempt():
            try:
                torch.cuda.set_device(local_rank)               # the current device is per thread
                g = dist.new_group(backend="nccl", timeout=datetime.timedelta(minutes=30))
                t = torch.ones(3, dtype=torch.int64, device="cuda")
                dist.all_reduce(t, group=g)
                torch.cuda.synchronize()
                if int(t[0].item()) != self.world:
                    raise RuntimeError("RCCL all-reduce returned %d, expected %d" % (int(t[0].item()), self.world))
                box["group"] = g
            except Exception as e:                              # noqa: BLE001 -- whatever it is, gloo takes over
                box["error"] = repr(e)

        th = threading.Thread(target=attempt, daemon=True)
        th.start()
        th.join(RCCL_DEADLINE_S)
        if th.is_alive():
            self._hung = True                                   # the thread stays parked; the process leaves through os._exit
            self.rccl_error = "RCCL group creation / first all-reduce did not finish within %d s" % RCCL_DEADLINE_S
            return False
        if "error" in box:
            self.rccl_error = box["error"]
            return False
        self._rccl = box["group"]
        return True

    # -- rank-uniform scalars and barriers: gloo, CPU tensors --
    def barrier(self):
        if self.world > 1:
            self.dist.barrier()

    def allmax(self, x):
        if self.world == 1:
            return float(x)
        t = self.torch.tensor([float(x)], dtype=self.torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def gather_floats(self, x):
        if self.world == 1:
            return [float(x)]
        t = self.torch.zeros(self.world, dtype=self.torch.float64)
        t[self.rank] = float(x)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [float(v) for v in t.tolist()]

    def gather_text(self, text, width=96):
        """One short ASCII string per rank (device index / uuid), as fixed-width bytes: nothing is pickled."""
        if self.world == 1:
            return [text]
        raw = text.encode("ascii", "replace")[:width].ljust(width, b" ")
        t = self.torch.zeros(self.world, width, dtype=self.torch.uint8)
        t[self.rank] = self.torch.frombuffer(bytearray(raw), dtype=self.torch.uint8)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return [bytes(row.tolist()).decode("ascii").rstrip() for row in t]

    def sum_totals(self, triple):
        """pass / fail / overflow totals of all ranks: the path's one optional collective, over RCCL when up."""
        if self.world == 1:
            return [int(v) for v in triple]
        torch, dist = self.torch, self.dist
        if self._rccl is not None:
            t = torch.tensor(list(triple), dtype=torch.int64, device="cuda")
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self._rccl)
            return [int(v) for v in t.tolist()]
        t = torch.tensor(list(triple), dtype=torch.int64)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [int(v) for v in t.tolist()]

    def sum_totals_rccl(self, triple):
        t = self.torch.tensor(list(triple), dtype=self.torch.int64, device="cuda")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self._rccl)
        return [int(v) for v in t.tolist()]

    def close_selftest(self):
        if self._hung:
            return                                              # the parked thread is a daemon; the N = 1 run goes on
        try:
            self.dist.destroy_process_group()
        except Exception:                                       # noqa: BLE001 -- a self-test never costs the headline
            pass
        self._rccl = None

    def close(self):
        if self.world == 1:
            return
        self.dist.barrier()
        if self._hung:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(0)                                         # a parked RCCL thread must not hold the exit
        self.dist.destroy_process_group()


class StubEngine:
    """--rehearse-on-cpu: no GPU, no library -- a step is a sleep.  Exists so that the launch shape, the port choice,
    the rank-uniform step arithmetic and every collective of an N-rank run can be exercised at any world size on a box
    without GPUs.  Never produces a number anyone should read."""

    class _Counts:
        n_pass = n_fail = n_overflow = 0

    def __init__(self, n, rank, delay_ms):
        self.n, self.delay = n, (1.0 + delay_ms * rank) * 1e-3
        self.pending = 0

    def step(self, counts=False):
        self.pending += 1
        if counts:
            self.synchronize()
            c = self._Counts()
            c.n_pass, c.n_fail, c.n_overflow = self.n // 2, self.n - self.n // 2, 0
            return c
        return None

    def synchronize(self):
        time.sleep(self.delay * self.pending)
        self.pending = 0


class ClockSampler:
    """The shader clock a GPU HOLDS while it works, read from sysfs (pp_dpm_sclk of the device's PCI function: the line with
    the asterisk) every 25 ms by a host thread: a throttled GPU shows here and in its kernel times, not only in the total."""

    def __init__(self, pci_bus_id):
        import threading
        self.path = "/sys/bus/pci/devices/%s/pp_dpm_sclk" % pci_bus_id.lower() if pci_bus_id else None
        self.mhz = []
        self._stop = threading.Event()
        self._th = None

    @staticmethod
    def parse(text):
        for line in text.splitlines():
            if line.rstrip().endswith("*"):
                digits = "".join(ch for ch in line.split(":", 1)[-1] if ch.isdigit())
                return int(digits) if digits else None
        return None

    def _run(self):
        while not self._stop.is_set():
            try:
                v = self.parse(open(self.path).read())
                if v:
                    self.mhz.append(v)
            except OSError:
                return
            self._stop.wait(0.025)

    def start(self):
        import threading
        if self.path and os.path.exists(self.path):
            self._th = threading.Thread(target=self._run, daemon=True)
            self._th.start()
        return self

    def stop(self):
        self._stop.set()
        if self._th:
            self._th.join(1.0)
        if not self.mhz:
            return None
        return {"mean_mhz": sum(self.mhz) / len(self.mhz), "min_mhz": min(self.mhz), "max_mhz": max(self.mhz),
                "samples": len(self.mhz), "source": self.path}


def pci_bus_id_of(device):
    """'0000:c1:00.0' of HIP device `device`, through the runtime the library is linked against (no torch)."""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(device)) == 0:
            return buf.value.decode()
    except OSError:
        pass
    return None


def threads_main(args):
    """--launch threads: the N > 1 measurement without a launcher and without any collective.  ONE process,
    N contexts (one per GPU) on N host threads: each thread owns its device-resident shard (read ids r*R .., generated on its
    own device), runs the same warm-up / settle / timed steps behind a thread barrier on both sides, and the line carries the
    max over the threads, every thread's own rate, per-device kernel times (HIP events) and the clock each device held.
    ctypes releases the GIL in every library call, and a step is asynchronous, so the threads never wait for each other
    outside the two barriers.  Replaces moira/moira.py:398-399 (`Pool(args.processors)`) the way SURVEY 8e puts it."""
    import threading
    from moira_amd import _lib as ML
    from moira_amd.engine import Engine
    world, L = args.gpus, args.length
    n = args.reads or CONFIG4_SHARD
    stride = args.stride or (L + 63) // 64 * 64
    have = ML.load().mpb_device_count()
    if have < 1:
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if not args.rehearse_on_one_gpu and have < world:
        sys.stderr.write("bench.py: --gpus %d but only %d GPU(s) are visible on this node\n" % (world, have))
        return 2
    bar = threading.Barrier(world)
    t_one, dts, res, errs = [0.0] * world, [0.0] * world, [None] * world, [None] * world
    plan = {}

    def work(r):
        eng = None
        try:
            dev = 0 if args.rehearse_on_one_gpu else r
            eng = Engine(dev)
            bus = pci_bus_id_of(dev)
            d_q, d_ee, d_ns, d_pass = eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
            eng.synth_fill(d_q, n, stride, fixed_len=L, seed=args.seed, first_read=r * n)
            params = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma)
            step = lambda c=False: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                                     params=params, want_counts=c)
            t0 = time.perf_counter()
            for _ in range(max(args.warmup, 1)):
                step()
            eng.synchronize()
            t_warm = (time.perf_counter() - t0) / max(args.warmup, 1)
            t1 = time.perf_counter(); step(); eng.synchronize()
            t_one[r] = max(min(time.perf_counter() - t1, t_warm), 1e-5)
            if bar.wait() == 0:                              # one thread turns the common step time into the common plan
                plan["steps"], plan["settle"] = plan_steps(max(t_one), args.steps, args.warmup)
                plan["t_step"] = max(t_one)
            bar.wait()
            steps, settle = plan["steps"], plan["settle"]
            for _ in range(settle):
                step()
            eng.synchronize()
            clock = ClockSampler(bus).start()
            bar.wait()                                       # ---- the timed region ----
            t0 = time.perf_counter()
            for _ in range(steps):
                step()
            eng.synchronize()
            dts[r] = time.perf_counter() - t0
            bar.wait()
            t_all = time.perf_counter() - t0                 # after the barrier: the slowest thread's end, as every thread saw it
            held = clock.stop()
            counts = step(True)
            ev = min(steps, 10)
            eng.timing(True); eng.timing_reset()
            for _ in range(ev):
                step()
            eng.synchronize()
            kt = {k: v[0] / max(v[1], 1) for k, v in eng.kernel_times().items() if v[1]}
            eng.timing(False)
            res[r] = {"device": dev, "pci_bus_id": bus, "t_all": t_all, "kernels_ms_per_step": kt, "held_clock": held,
                      "pass": counts.n_pass, "fail": counts.n_fail, "overflow": counts.n_overflow,
                      "path": eng.last_path()["narrow_rows"]}
            for b in (d_q, d_ee, d_ns, d_pass):
                b.free()
        except BaseException as e:                           # noqa: BLE001 -- whatever it is, nobody waits for this thread
            errs[r] = e
            bar.abort()
        finally:
            if eng is not None:
                try:
                    eng.close()
                except Exception:                            # noqa: BLE001
                    pass

    ths = [threading.Thread(target=work, args=(r,)) for r in range(world)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    bad = [(r, e) for r, e in enumerate(errs) if e is not None and not isinstance(e, threading.BrokenBarrierError)]
    if bad or any(x is None for x in res):
        for r, e in bad:
            sys.stderr.write("bench.py --launch threads: device thread %d failed: %r\n" % (r, e))
        return 1
    steps, settle = plan["steps"], plan["settle"]
    dt = max(x["t_all"] for x in res)
    wl = ("BASELINE configs[3]: %d synthetic %d bp reads sharded host-side across %d x MI355X, %d reads (%.1f GB resident, "
          "generated on device, read ids rank*R..) per GPU, poisson_binomial filter, alpha 0.005, uncert 0.01 (uint8 %d x %d "
          "per GPU, seed %d)" % (n * world, L, world, n, n * stride / 1e9, n, stride, args.seed))
    if args.rehearse_on_one_gpu:
        wl = "REHEARSAL: %d contexts share GPU 0 -- exercises the threads launch, not a scaling number; " % world + wl
    dp = [x["kernels_ms_per_step"].get("dp") for x in res]
    line = {"metric": "reads/sec filtered (300 bp synthetic)", "value": n * world * steps / dt, "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "settle_steps": settle, "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "launch": "threads",
            "fell_back_from": os.environ.get("BENCH_FELL_BACK_FROM"),
            "config": {"workload": wl, "reads_per_gpu": n, "read_length": L, "row_stride": stride,
                       "parallelism": "host-side split, %d device threads of one process, no data-path collective" % world,
                       "collective_backend": "none (one process: the pass / fail totals are summed on the host)",
                       "world_size": world, "mode": "fast_fma (NOT bit-exact)" if args.fast_fma else "bit-exact (no FMA)"},
            "weak_scaling_anchor": "every GPU holds %d reads; the N = 1 point of this curve is `python bench.py --gpus 1 --reads %d` "
                                   "(= extras.config4_shard of the plain N = 1 line)" % (n, n),
            "devices": ["thread %d: GPU %d pci %s" % (r, x["device"], x["pci_bus_id"]) for r, x in enumerate(res)],
            "timed_region_s": dt, "t_step_rank_uniform_s": plan["t_step"],
            "reads_per_s_per_rank": [n * steps / t for t in dts],
            "per_rank": [{"kernels_ms_per_step": x["kernels_ms_per_step"], "held_clock": x["held_clock"]} for x in res],
            "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "k_dp",
                         "algorithmic_bytes_per_launch": (L + 13) * n,
                         "avg_launch_ms": (sum(dp) / len(dp)) if all(dp) else None,
                         "achieved": ((L + 13) * n / (sum(dp) / len(dp)) / 1e6) if all(dp) else None,
                         "frac": ((L + 13) * n / (sum(dp) / len(dp)) / 1e6 / HBM_PEAK_GBS) if all(dp) else None,
                         "frac_whole_step": (L + 13) * n * world / (dt / steps) / 1e9 / HBM_PEAK_GBS / world, "traffic": None},
            "outcome": {"pass": sum(x["pass"] for x in res), "fail": sum(x["fail"] for x in res),
                        "overflow_reruns": sum(x["overflow"] for x in res)}}
    print(json.dumps(line))
    sys.stdout.flush()
    return 0


def plan_steps(t_step_max, steps_arg, warmup):
    """(steps, settle) from the RANK-UNIFORM step time (the max over ranks) -- plain arithmetic, so that every rank
    takes the same branches and issues the same collectives (a collective under a rank-local condition
    pairs up wrongly across ranks).  steps_arg <= 0: >= 1 s of timed work.  A timed region shorter than 1 s is
    preceded by untimed steps until >= 0.5 s of back-to-back work has run (the chip lowers its clock under sustained
    FP64 load; the number reported is the sustained one)."""
    t = max(t_step_max, 1e-5)
    steps = steps_arg if steps_arg > 0 else int(max(10.0, min(4000.0, 1.25 / t + 1)))
    settle = int(max(0.0, 0.5 / t - max(warmup, 1) - 1)) if steps * t < 1.0 else 0
    return steps, settle


def _decode_rank(text):
    """one rank's {"k": kernel ms per step, "mhz": [mean, min, max] held clock} as gathered by Collectives.gather_text"""
    try:
        d = json.loads(text)
        mhz = d.get("mhz")
        return {"kernels_ms_per_step": d.get("k"),
                "held_clock": {"mean_mhz": mhz[0], "min_mhz": mhz[1], "max_mhz": mhz[2]} if mhz else None}
    except ValueError:
        return {"undecodable": text}
