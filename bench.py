#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X Poisson-binomial read filter on BASELINE.json's configs.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both launch shapes work for every N: started WITHOUT torchrun and with --gpus N > 1, this script
starts `python -m torch.distributed.run ... bench.py <same args>` as a CHILD process (before anything
in this process has touched the GPU) and exits with the child's return code.

Workload (config.workload):
  N = 1   BASELINE.json configs[1] -- 10,000,000 synthetic single-end 300 bp reads (row stride 320,
          seed 2, counter-based generator of include/mpb_synth.h, filled ON DEVICE so the inputs are
          resident in HBM when the timed region starts).
  N > 1   BASELINE.json configs[3] -- "1 B synthetic 300 bp reads sharded host-side across 8 x
          MI355X": every rank owns 125,000,000 reads (40 GB resident; read ids rank*R .. rank*R+R-1,
          generated on its own device), no data-path collective: weak
          scaling (N = 2, 4 are the partial node with the same per-GPU shard).
A "step" = one pass of the whole hot path over the resident batch: prepass -> scan -> scatter ->
DP -> overflow pass, producing ee / Ns / pass for every read.  The wall-clock region carries no
per-kernel events; the per-kernel HIP-event timing behind `roofline` is taken in a separate pass
right after it.  Without --steps the timed region is sized to last >= 1 s (1.25 s of steps at the
warm-up rate).

Collectives (N > 1): the data path has none.  Barriers, the rank-uniform step count and the max over ranks go over a
gloo group (CPU tensors: nothing that can hang on a GPU communicator brackets the timed region beyond
torch.cuda.synchronize()); the one optional collective of the path -- the 24-byte pass / fail / overflow totals -- goes
over RCCL (xGMI) when the communicator comes up, and over gloo otherwise: the RCCL group is created and tried in a helper
thread with a deadline, every rank votes, and on any failure all ranks use gloo IN THE SAME PROCESS (a process that has
touched the GPU is never re-executed).  The line records which backend carried the totals and each rank's device.
`--rehearse-on-one-gpu` (ranks share device 0; at most 5 ranks: a GPU box allows 6 processes on its card) and
`--rehearse-on-cpu` (no GPU at all: the step is a stub; any world size) exercise this launch / collective code.

Weak-scaling anchor: N > 1 times 125 M reads per GPU (configs[3]); the plain N = 1 run times configs[1] (10 M reads, the
metric's own configuration).  The single-GPU point of the SAME per-GPU workload is `extras.config4_shard` of the N = 1
line, or `python bench.py --gpus 1 --reads 125000000`.

One JSON line on rank 0.  Extra objects:
  roofline     dominant kernel (k_dp): algorithmic bytes (L + 13 per read) per launch
               / its mean duration measured with HIP events on the library's stream.
  cpu_baseline the real reference extension (oracle/_ref, kind "reference") called per read from
               Python exactly as moira.py does: on 1 core, and (`all_cores`) in P worker processes
               with one contiguous shard each -- what `moira.py --processors P` amounts to --
               P = the CPUs this box grants (cgroup quota); or the oracle's reference-shaped port
               (kind "port") when oracle/_ref is absent.  Bounded samples of the same reads.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK = 39.3e12       # v_mul/add_f64 lane-ops per second: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz
CONFIG2_READS = 10_000_000     # BASELINE configs[1]
CONFIG4_SHARD = 125_000_000    # BASELINE configs[3]: 1 B reads / 8 GPUs


def usable_cpus():
    """CPUs this process may use: the cgroup quota when there is one, else the affinity mask."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period) + 0.5)))
    except (OSError, ValueError):
        pass
    return n


# ---- CPU baseline (the checker's code, timed beside the GPU number; never the thing shipped) ----

def _ref_worker(argv):
    """Child process of the all-cores leg: one contiguous shard of the workload through the real reference
    extension, one Python call per read (what a moira.py Pool worker does).  Prints one JSON line."""
    first, n, seed, L, stride = (int(x) for x in argv)
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pb_oracle as O
    ref = O.reference_module()
    q, _ = O.synth_fill(n, stride, fixed_len=L, seed=seed, first_read=first)
    rows = q[:, :L]
    seqs = ["".join("N" if v == 0 else "A" for v in r) for r in rows]
    quals = [[int(v) if v else 20 for v in r] for r in rows]
    sys.stdout.write("ready\n")
    sys.stdout.flush()
    sys.stdin.readline()                                  # common start signal
    t0 = time.time()
    for s, qq in zip(seqs, quals):
        ref.calculate_errors_PB(s, qq, 0.005)
    t1 = time.time()
    print(json.dumps({"n": n, "t0": t0, "t1": t1}))


def cpu_baseline(seed, L, stride, budget_s=8.0):
    """Time the CPU path on bounded samples of the same workload (rank 0, N=1 only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import pb_oracle as O
    O.build()
    P = usable_cpus()
    out = {}
    # port: reference-shaped loop nest, 1 thread
    probe = 2000
    q, _ = O.synth_fill(probe, stride, fixed_len=L, seed=seed)
    t = time.perf_counter(); O.filter_batch(q, fixed_len=L, shape=1, threads=1); dt = time.perf_counter() - t
    n_port = int(max(probe, min(400000, budget_s * 0.5 * probe / dt)))
    q, _ = O.synth_fill(n_port, stride, fixed_len=L, seed=seed)
    t = time.perf_counter(); O.filter_batch(q, fixed_len=L, shape=1, threads=1); dt = time.perf_counter() - t
    port = {"value": n_port / dt, "unit": "reads/s", "cores": 1, "kind": "port",
            "sample": "first %d reads of the workload, oracle/pb_oracle.c reference-shaped loop nest" % n_port}
    # the same port on all granted cores (OpenMP, one contiguous shard per thread)
    n_port_p = n_port * min(P, O.lib().pbo_max_threads())
    qp, _ = O.synth_fill(n_port_p, stride, fixed_len=L, seed=seed)
    pt = min(P, O.lib().pbo_max_threads())
    t = time.perf_counter(); O.filter_batch(qp, fixed_len=L, shape=1, threads=pt); dtp = time.perf_counter() - t
    port["all_cores"] = {"value": n_port_p / dtp, "unit": "reads/s", "cores": pt,
                         "sample": "first %d reads, %d OpenMP threads, one contiguous shard each" % (n_port_p, pt)}
    # fast restatement on all cores (for orientation only)
    n_fast = min(len(qp), 400000)
    t = time.perf_counter(); O.filter_batch(qp[:n_fast], fixed_len=L, shape=0, threads=pt); dt = time.perf_counter() - t
    out["cpu_fast_restatement"] = {"value": n_fast / dt, "unit": "reads/s", "cores": pt,
                                   "note": "two-term recurrence (not the reference's algorithmic shape)"}
    ref = O.reference_module()
    if ref is None:
        out["cpu_baseline"] = port
        return out
    n_ref = int(max(500, min(200000, budget_s * port["value"] / 1.6)))
    rows = q[:n_ref, :L]
    seqs = ["".join("N" if v == 0 else "A" for v in r) for r in rows]
    quals = [[int(v) if v else 20 for v in r] for r in rows]
    t = time.perf_counter()
    for s, qq in zip(seqs, quals):
        ref.calculate_errors_PB(s, qq, 0.005)
    dt = time.perf_counter() - t
    base = {"value": n_ref / dt, "unit": "reads/s", "cores": 1, "kind": "reference",
            "sample": "first %d reads of the workload through oracle/_ref/bernoulli.so "
                      "(moira/bernoullimodule.c built unmodified), one Python call per read "
                      "as moira.py --processors 1 does" % n_ref}
    # all granted cores: P worker processes, one contiguous shard of n_ref reads each, started together
    procs = []
    try:
        for w in range(P):
            procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__), "--_ref-worker",
                                           str(w * n_ref), str(n_ref), str(seed), str(L), str(stride)],
                                          stdin=subprocess.PIPE, stdout=subprocess.PIPE, text=True))
        for p in procs:
            if p.stdout.readline().strip() != "ready":
                raise RuntimeError("reference worker failed to start")
        for p in procs:
            p.stdin.write("go\n"); p.stdin.flush()
        res = [json.loads(p.stdout.readline()) for p in procs]
        for p in procs:
            p.wait(timeout=60)
        wall = max(r["t1"] for r in res) - min(r["t0"] for r in res)
        base["all_cores"] = {"value": sum(r["n"] for r in res) / wall, "unit": "reads/s", "cores": P,
                             "processes": P,
                             "sample": "first %d reads of the workload, %d worker processes with one contiguous shard "
                                       "of %d reads each through oracle/_ref/bernoulli.so (what moira.py "
                                       "--processors %d does, minus its per-read pickling)" % (P * n_ref, P, n_ref, P)}
    except Exception as e:                                  # the 1-core figure stands on its own
        base["all_cores"] = {"value": None, "cores": P, "error": repr(e)}
        for p in procs:
            if p.poll() is None:
                p.kill()
    out["cpu_baseline"] = base
    out["cpu_baseline_all_cores"] = base["all_cores"]
    out["cpu_port"] = port
    return out


# ---- launch shape --------------------------------------------------------------------------------

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
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
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
    return subprocess.call([sys.executable, os.path.abspath(__file__)] + argv + ["--launch", "threads"], env=env)


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


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "--_ref-worker":
        return _ref_worker(sys.argv[2:])
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="timed steps (default: as many as make the timed region >= 1 s)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=0, help="reads per GPU (default: 10 M at N=1 = configs[1]; "
                                                          "125 M at N>1 = one eighth of configs[3]; "
                                                          "`--gpus 1 --reads 125000000` is the N=1 point of the weak-scaling curve)")
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--fast-fma", action="store_true", help="non-bit-exact FMA mode (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stride", type=int, default=0, help="row stride in bytes (default: length rounded up to 64; experiments)")
    ap.add_argument("--no-extras", action="store_true", help="skip the opt-in-mode extra runs (profiling)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank dry run on a 1-GPU box: every rank uses device 0 and the (tiny) "
                         "collectives go over gloo; exercises the N>1 code path, not a scaling number")
    ap.add_argument("--rehearse-on-cpu", action="store_true",
                    help="multi-rank dry run without any GPU: the step is a stub; exercises launch + collectives only")
    ap.add_argument("--no-rccl", action="store_true", help="do not try RCCL for the 24-byte totals (gloo only)")
    ap.add_argument("--launch", choices=["auto", "processes", "threads"], default="auto",
                    help="N > 1: `processes` = one rank per GPU under torch.distributed.run (the driver's shape); `threads` = N "
                         "contexts on N host threads of ONE fresh process, no launcher, no collective (SURVEY 8e: 'one host "
                         "thread (or process) + one HIP stream per device'); `auto` = processes, and threads when the "
                         "process job ends non-zero without having printed its line")
    ap.add_argument("--rccl-selftest", action="store_true",
                    help="N = 1 only: before the run, bring up a ONE-rank gloo + RCCL group on this GPU and do the 24-byte "
                         "all-reduce on it (under the same deadline as an N > 1 run); reported as `rccl_selftest`")
    ap.add_argument("--_rank-delay-ms", type=float, default=0.0, help=argparse.SUPPRESS)    # tests: rank r's stub step takes 1 + r*this ms
    ap.add_argument("--_die-rank", type=int, default=-1, help=argparse.SUPPRESS)             # tests: this rank dies before the timed region
    ap.add_argument("--_try-rccl", action="store_true", help=argparse.SUPPRESS)              # tests: attempt RCCL even in a rehearsal (where it must fail) -> the fallback
    args = ap.parse_args()
    rehearsal = args.rehearse_on_one_gpu or args.rehearse_on_cpu

    if args.launch == "threads" and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return threads_main(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if "WORLD_SIZE" not in os.environ and args.gpus > 1:
            sys.exit(self_launch(args.gpus, rehearsal, fall_back_to_threads=(args.launch == "auto" and not args.rehearse_on_cpu)))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = not args.rehearse_on_cpu
    if use_gpu:
        import torch
        have = visible_gpus()
        if have < 1:
            raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
        if args.rehearse_on_one_gpu:
            local_rank = 0
        elif local_rank >= have:
            raise SystemExit("rank %d wants GPU %d but only %d GPU(s) are visible (--gpus %d needs %d)"
                             % (rank, local_rank, have, args.gpus, args.gpus))
        torch.cuda.set_device(local_rank)
    coll = Collectives(world, rank, local_rank, use_gpu,
                       try_rccl=(use_gpu and not rehearsal and not args.no_rccl) or args._try_rccl,
                       selftest=args.rccl_selftest and world == 1 and use_gpu)

    L = args.length
    n = args.reads or (CONFIG2_READS if world == 1 else CONFIG4_SHARD)
    stride = args.stride or (L + 63) // 64 * 64          # 300 -> 320
    if use_gpu:
        from moira_amd.engine import Engine
        eng = Engine(local_rank)
        props = torch.cuda.get_device_properties(local_rank)
        dev_text = "rank %d: GPU %d %s uuid %s" % (rank, local_rank, props.name, getattr(props, "uuid", "n/a"))
        d_q = eng.alloc(n * stride)
        d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=args.seed, first_read=rank * n)
        params = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma)

        def step(counts=False):
            return eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                     params=params, want_counts=counts)

        def device_sync():
            eng.synchronize()
            torch.cuda.synchronize()
    else:
        eng = StubEngine(n, rank, args._rank_delay_ms)
        dev_text = "rank %d: no GPU (--rehearse-on-cpu)" % rank
        step, device_sync = eng.step, eng.synchronize
    sys.stderr.write(dev_text + "\n")
    devices = coll.gather_text(dev_text)

    def barrier():
        device_sync()
        coll.barrier()

    t0 = time.perf_counter()
    for _ in range(max(args.warmup, 1)):
        step()
    device_sync()
    t_warm = (time.perf_counter() - t0) / max(args.warmup, 1)
    t1 = time.perf_counter(); step(); device_sync(); t_one = time.perf_counter() - t1
    # the step time every decision below is made from is the SAME number on every rank (one unconditional collective)
    t_step = coll.allmax(max(min(t_one, t_warm), 1e-5))
    steps, settle = plan_steps(t_step, args.steps, args.warmup)
    for _ in range(settle):
        step()
    device_sync()
    if rank == args._die_rank:
        os._exit(3)                                      # tests: a rank that dies must fail the whole job, promptly
    # ---- the timed region: no events, no host round trips (a host thread reads the device's clock from sysfs meanwhile) ----
    clock = ClockSampler(pci_bus_id_of(local_rank) if use_gpu else None).start()
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    device_sync()
    dt_rank = time.perf_counter() - t0                   # this rank's own time, before it waits for the others
    barrier()
    dt = coll.allmax(time.perf_counter() - t0)
    held_clock = clock.stop()
    per_rank = [n * steps / t for t in coll.gather_floats(dt_rank)]
    counts = step(counts=True)
    n_pass, n_fail, n_ovf = coll.sum_totals((counts.n_pass, counts.n_fail, counts.n_overflow))

    if not use_gpu:
        if rank == 0:
            print(json.dumps({"metric": "reads/sec filtered (300 bp synthetic)", "value": n * world * steps / dt, "unit": "reads/s",
                              "n_gpus": world, "steps": steps, "warmup": args.warmup, "settle_steps": settle,
                              "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                              "dtype": "f64", "data": "none (CPU rehearsal of the launch / collective code: the step is a stub)",
                              "config": {"workload": "REHEARSAL ON CPU -- not a measurement", "reads_per_gpu": n,
                                         "world_size": world,
                                         "collective_backend": "gloo (rehearsal)" if not args._try_rccl else
                                         "gloo (RCCL not used: %s)" % coll.rccl_error},
                              "t_step_rank_uniform_s": t_step, "timed_region_s": dt, "reads_per_s_per_rank": per_rank,
                              "devices": devices, "outcome": {"pass": n_pass, "fail": n_fail, "overflow_reruns": n_ovf}}))
        coll.close()
        return

    # ---- per-kernel HIP events on the library's stream, in a pass of their own ----
    ev_steps = min(steps, 20)
    eng.timing(True)
    eng.timing_reset()
    for _ in range(ev_steps):
        step()
    eng.synchronize()
    times = eng.kernel_times()
    eng.timing(False)
    # every rank's own kernel times and held clock, so that a throttled GPU shows in the line (fixed-width text, nothing pickled)
    mine = {"k": {k: round(v[0] / max(v[1], 1), 4) for k, v in times.items() if v[1]},
            "mhz": [round(held_clock[x]) for x in ("mean_mhz", "min_mhz", "max_mhz")] if held_clock else None}
    per_rank_text = coll.gather_text(json.dumps(mine, separators=(",", ":")), width=480)
    hist = eng.class_histogram()
    # one more pass, untimed, that also sums the algorithmic DP cells on the device (fp64_valu.frac_algorithmic)
    prm_cells = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma, count_cells=True)
    eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_cells, want_counts=False)
    alg_cells = eng.algorithmic_cells()
    # extra (NOT the headline, work is skipped by design): opt-in MPB_FLAG_DECISION_ONLY, same batch
    extras = {}
    # N = 1 only: with more ranks the others would sit in the closing barrier (120 s timeout) while rank 0 runs them,
    # and they describe the single-GPU library, not the multi-GPU run
    if not args.no_extras and rank == 0 and world == 1:
        def rate(prm, reps=5, **kw):
            a = dict(d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm)
            a.update(kw)
            for _ in range(2):
                eng.filter_device(want_counts=False, **a)
            eng.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                eng.filter_device(want_counts=False, **a)
            eng.synchronize()
            return (time.perf_counter() - t1) / reps

        prm_do = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma, decision_only=True)
        dt_do = rate(prm_do, d_q=d_q, n=n, stride=stride, fixed_len=L)
        counts_do = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_do)
        hist_do = eng.class_histogram()
        extras = {"decision_only_mode": {
            "note": "opt-in MPB_FLAG_DECISION_ONLY on the same resident batch of rank 0: reads proven to fail "
                    "(Chernoff bound) skip their DP and report ee=+inf; identical pass/fail flags; NOT the headline",
            "reads_per_s_this_rank": n / dt_do, "ms_per_step": dt_do * 1e3,
            "pass": counts_do.n_pass, "reads_run_through_dp": int(sum(hist_do.values()))}}
        if not args.fast_fma:
            # opt-in MPB_FLAG_FAST_FMA: 2 FP64 ops per DP cell instead of 3; ee within 1e-9 relative (north_star's
            # tolerance), NOT bit-identical, so not the default and not the headline
            prm_f = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=True)
            dt_f = rate(prm_f, d_q=d_q, n=n, stride=stride, fixed_len=L)
            c_f = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_f)
            extras["fast_fma_mode"] = {
                "note": "opt-in MPB_FLAG_FAST_FMA (a*v + b*w contracted into one fma): ee within 1e-9 relative of the "
                        "reference instead of bit-identical; NOT the headline", "reads_per_s_this_rank": n / dt_f,
                "ms_per_step": dt_f * 1e3, "pass": c_f.n_pass}
        # BASELINE configs[4] (ragged 50-600 bp) on the same GPU: a parity-test case, reported for reference
        nr, sr = max(min(n, CONFIG2_READS) // 2, 1), 608
        r_q, r_len = eng.alloc(nr * sr), eng.alloc(nr * 4)
        eng.synth_fill(r_q, nr, sr, fixed_len=0, min_len=50, max_len=600, d_len=r_len, seed=5)
        dt_r = rate(params, d_q=r_q, n=nr, stride=sr, d_len=r_len)
        extras["ragged_config5"] = {
            "note": "lengths U{50..600} in one stride-608 matrix, reads sorted by (class, length bin) on the device; "
                    "bit-exact mode; NOT the headline", "reads": nr, "reads_per_s_this_rank": nr / dt_r,
            "ms_per_step": dt_r * 1e3}
        r_q.free()
        r_len.free()
        if world == 1:
            extras["classified_at_source"] = classified_rate(eng, d_q, n, stride, L, params, d_ee, d_ns, d_pass)
            extras["long_reads_ragged_50_2000"] = long_ragged_rate(eng, params)
            extras["host_fed"] = host_fed_rate(eng, L, stride, args.seed)
            try:
                extras["config3_paired"] = config3_paired_rate(eng)
            except Exception as e:                      # an extra must never cost the headline line
                extras["config3_paired"] = {"error": repr(e)}
            extras["per_read_broker"] = per_read_broker_rate()
            extras["per_read_in_process"] = per_read_in_process_rate(eng)
            if n != CONFIG4_SHARD:
                extras["config4_shard"] = config4_shard_rate(eng, L, stride, args.seed, params)
        extras["poisson_error_calc"] = poisson_rate(eng, d_q, n, stride, L, d_ee, d_ns)
        if world == 1:
            extras["high_quality_300"] = high_quality_rate(eng, min(n, CONFIG2_READS), stride, L, args.seed, d_ee, d_ns, d_pass)
            extras["real_profile"] = real_profile_rate(eng)

    if rank == 0:
        total_reads = n * world * steps
        value = total_reads / dt
        dp_ms, dp_n = times["dp"]
        dp_avg_s = dp_ms / max(dp_n, 1) / 1e3
        alg_bytes = (L + 13) * n
        achieved = alg_bytes / dp_avg_s / 1e9 if dp_n else None
        cells = sum(cap * cnt for cap, cnt in hist.items()) * L        # DP cells one launch evaluates
        opc = 2 if args.fast_fma else 3
        # FP64 lane-operations actually issued: row 0 of a one-lane-per-read class (budget <= 16) is a single
        # multiply (there is no row -1), every other cell costs `opc`
        issued = sum((opc * cap - (opc - 1 if cap <= 16 else 0)) * cnt for cap, cnt in hist.items()) * L
        # HBM bytes of one k_dp launch from the PMC counters (collected by tools/collect_profiles.sh in
        # separate rocprofv3 passes, corrected as MI355X_MICROARCH.md prescribes); only valid for the
        # workload it was measured on
        traffic, valu_pmc, pmc_source = None, None, None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.fast_fma:
            t = json.load(open(tpath))
            w = t.get("workload", {})
            if (w.get("reads"), w.get("length"), w.get("seed")) == (n, L, args.seed):
                traffic = t["hbm_bytes_per_launch"]
                valu_pmc = t.get("valu")
                pmc_source = t.get("source")
        step_s = dt / steps
        fp64_floor_ms = cells * opc / FP64_VALU_PEAK * 1e3
        hbm40_ms = alg_bytes / (0.40 * HBM_PEAK_GBS * 1e9) * 1e3
        if world == 1:
            wl = ("BASELINE configs[1]: %d synthetic single-end %d bp reads, poisson_binomial filter, alpha 0.005, "
                  "uncert 0.01, resident in HBM (uint8 %d x %d, seed %d)" % (n, L, n, stride, args.seed))
            if n == CONFIG4_SHARD:
                wl = ("one shard of BASELINE configs[3] on one GPU (the N = 1 point of the weak-scaling curve): " + wl.split(": ", 1)[1])
        else:
            wl = ("BASELINE configs[3]: %d synthetic %d bp reads sharded host-side across %d x MI355X, %d reads "
                  "(%.1f GB resident, generated on device, read ids rank*R..) per GPU, poisson_binomial filter, "
                  "alpha 0.005, uncert 0.01 (uint8 %d x %d per GPU, seed %d)"
                  % (n * world, L, world, n, n * stride / 1e9, n, stride, args.seed))
        if world == 1 and n != CONFIG4_SHARD:
            shard = extras.get("config4_shard", {})
            anchor = ("this N = 1 line times configs[1] (%d reads); runs with N > 1 time %d reads per GPU (configs[3]).  The "
                      "single-GPU point of that per-GPU workload is extras.config4_shard of this line%s, or "
                      "`python bench.py --gpus 1 --reads %d`"
                      % (n, CONFIG4_SHARD, (" (%.4g reads/s)" % shard["reads_per_s"]) if "reads_per_s" in shard else "", CONFIG4_SHARD))
        else:
            anchor = ("every GPU holds %d reads; the N = 1 point of this curve is `python bench.py --gpus 1 --reads %d` "
                      "(= extras.config4_shard of the plain N = 1 line), not the plain N = 1 line itself, which times "
                      "configs[1] (%d reads)" % (n, n, CONFIG2_READS))
        line = {
            "metric": "reads/sec filtered (300 bp synthetic)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": steps, "warmup": args.warmup, "settle_steps": settle,
            "ms_per_step": step_s * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": wl,
                       "reads_per_gpu": n, "read_length": L, "row_stride": stride,
                       "parallelism": "host-side split, %d rank(s), no data-path collective" % world,
                       "collective_backend": (None if world == 1 else
                                              {"nccl": "RCCL (24-byte totals); gloo (barriers, step count, max over ranks)",
                                               "gloo": "gloo%s" % (" (rehearsal)" if rehearsal and not args._try_rccl else
                                                                   " (RCCL not used: %s)" % (coll.rccl_error or "--no-rccl"))}[coll.totals_backend]),
                       "world_size": world,
                       "mode": "fast_fma (NOT bit-exact)" if args.fast_fma else "bit-exact (no FMA)"},
            "weak_scaling_anchor": anchor,
            "devices": devices,
            "rccl_selftest": coll.selftest,
            "timed_region_s": dt,
            "t_step_rank_uniform_s": t_step,
            "reads_per_s_per_rank": per_rank,
            "launch": "processes" if world > 1 else "single process",
            "per_rank": [_decode_rank(t) for t in per_rank_text],
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per k_dp launch (PMC, profiles/pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel": "k_dp", "avg_launch_ms": dp_avg_s * 1e3, "launches": dp_n,
                         "algorithmic_bytes_per_read": L + 13,
                         "frac_whole_step": alg_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                         "unreachable_note": "exact (3-rounding) arithmetic makes this path FP64-VALU-issue bound, not HBM bound: "
                                             "%.3g DP cells x %d FP64 ops / %.3g lane-op/s nominal peak = %.2f ms floor per "
                                             "launch, vs the %.2f ms that 40 %% of the 8 TB/s HBM roof would need for these %.3g "
                                             "algorithmic bytes; k_dp runs %.2f ms"
                                             % (cells, opc, FP64_VALU_PEAK, fp64_floor_ms, hbm40_ms, alg_bytes, dp_avg_s * 1e3)},
            "fp64_valu": {"cells_per_launch": cells, "ops_per_cell": opc,
                          "achieved_ops_per_s": cells * opc / dp_avg_s if dp_n else None,
                          "peak_ops_per_s": FP64_VALU_PEAK,
                          "frac": cells * opc / dp_avg_s / FP64_VALU_PEAK if dp_n else None,
                          "floor_ms_per_launch": fp64_floor_ms,
                          "issued_ops_per_launch": issued,
                          "frac_issued": issued / dp_avg_s / FP64_VALU_PEAK if dp_n else None,
                          # the cells the ALGORITHM needs -- sum_k min(k + 1, J) per read, J from the epilogue's crossing row,
                          # summed on the device in one extra untimed pass (MPB_FLAG_COUNT_CELLS) -- against the cells the
                          # row-budget classes pay for (`cells_per_launch`)
                          "cells_algorithmic_per_launch": alg_cells,
                          "cells_algorithmic_per_read": (alg_cells / n) if alg_cells else None,
                          "frac_algorithmic": alg_cells * opc / dp_avg_s / FP64_VALU_PEAK if (dp_n and alg_cells) else None,
                          # VALU issue slots used by k_dp, normalised by the clock the chip held (PMC pass of the same
                          # command; the counters do not travel with the driver's run, the committed summary does)
                          "valu_busy_pmc": valu_pmc,
                          "sources": {"avg_launch_ms, cells_*, frac*": "measured live in this run (HIP events on the library's "
                                                                      "stream; class histogram; MPB_FLAG_COUNT_CELLS pass)",
                                      "valu_busy_pmc, roofline.traffic": pmc_source,
                                      "kernel average to compare avg_launch_ms with": "profiles/r04_kernel_stats.csv "
                                                                                      "(rocprofv3 --kernel-trace --stats)"},
                          "note": "the binding roof: a scalar FP64 recurrence (SURVEY §8d); `frac` counts 3 ops for every cell of the row-budget classes, `frac_issued` the FP64 instructions actually issued (row 0 of a one-lane class is one multiply), `frac_algorithmic` 3 ops for every cell the algorithm needs"},
            "kernels_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in times.items()},
            "kernel_event_pass": {"steps": ev_steps, "note": "HIP events on the library's stream, separate from the wall-clock region"},
            "outcome": {"pass": n_pass, "fail": n_fail, "overflow_reruns": n_ovf},
            "row_budget_histogram": {str(k): v for k, v in hist.items() if v},
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            line.update(cpu_baseline(args.seed, L, stride))
        print(json.dumps(line))
        sys.stdout.flush()
    for b in (d_q, d_ee, d_ns, d_pass):
        b.free()
    eng.close()
    coll.close()


def config3_paired_rate(eng, pairs_per_chunk=1_000_000, chunks=4, stage_chunks=None):
    """BASELINE configs[2] ("100M 2x300 bp paired reads, NW contig on CPU then GPU filter") at a size a bench run can
    afford: `chunks` x `pairs_per_chunk` synthetic 2 x 300-base pairs (450-base fragments: 150 bases of overlap) as FASTQ
    TEXT IN MEMORY -> record index -> contig construction on the host cores (mothur-style NW + consensus, the build's own
    libmoira_contig.so; north_star keeps it on the CPU; ref: moira/moira.py:789-801, moira/nw_align.pyx:49-201) -> pack ->
    GPU filter from host memory.  Stage rates from a pass with the stages one after the other; the end-to-end rate from a
    pass in which index + contigs of chunk k+1 run on a second thread while chunk k is packed and filtered (what the CLI
    does).  The same chunk of text is processed `chunks` times (its content does not change what any stage costs).
    stage_chunks: chunks of the one-after-the-other pass (default: all of them; tools/config3_full.py streams 100 chunks through
    the pipelined pass and takes the stage rates from 4)."""
    import threading
    import numpy as np
    from moira_amd import contig as CT, fastio as F
    n, L, frag, W = pairs_per_chunk, 300, 450, 615
    rng = np.random.default_rng(3)
    base = min(n, 250_000)                                   # distinct pairs generated; tiled up to a chunk
    B = np.frombuffer(b"ACGT", np.uint8)
    comp = np.zeros(256, np.uint8)
    for a, b in zip(b"ACGT", b"TGCA"):
        comp[a] = b
    frags = B[rng.integers(0, 4, (base, frag))]
    fwd, rev = frags[:, :L].copy(), comp[frags[:, frag - L:][:, ::-1]]
    for a in (fwd, rev):                                     # ~0.7 % substitutions, concentrated towards the 3' end
        pos = np.minimum((rng.random((base, 2)) ** 0.4 * L).astype(int), L - 1)
        a[np.arange(base)[:, None], pos] = B[rng.integers(0, 4, (base, 2))]
    qual = (np.clip(38 - (np.arange(L) / L) ** 3 * rng.integers(4, 30, (base, 1)) - rng.integers(0, 6, (base, L)), 2, 40) + 33).astype(np.uint8)
    bufs = []
    for arr in (fwd, rev):                                   # fixed-width records: "@p%08d\n" seq "\n+\n" qual "\n" = 614 bytes
        rec = np.empty((n, W), np.uint8)
        rec[:, 0], rec[:, 1] = ord("@"), ord("p")
        ids = np.arange(n)
        for d in range(8):
            rec[:, 9 - d] = 48 + (ids // 10 ** d) % 10
        rec[:, 10] = 10
        reps = (n + base - 1) // base
        rec[:, 11:311] = np.tile(arr, (reps, 1))[:n]
        rec[:, 311], rec[:, 312], rec[:, 313] = 10, ord("+"), 10
        rec[:, 314:614] = np.tile(qual, (reps, 1))[:n]
        rec[:, 614] = 10
        bufs.append(rec.reshape(-1))
    del frags, fwd, rev, qual
    threads = CT.usable_cpus()
    fbuf, rbuf = bufs
    eng.filter(np.full((8, 608), 30, np.uint8), fixed_len=600)          # warm-up of the host pipeline's slots

    def front(_k):
        fidx, fc, e1 = F.index(fbuf, True, n, threads=threads)
        ridx, rc_, e2 = F.index(rbuf, True, n, threads=threads)
        assert e1 is None and e2 is None and len(fidx) == len(ridx) == n
        t = time.perf_counter()
        out = CT.contigs_from_fastq(fbuf, fidx, rbuf, ridx, 33, threads=threads)
        return out, t

    def back(cb):
        cbuf, cidx, aux = cb
        t0 = time.perf_counter()
        q, lens, has_n = F.pack(cbuf, cidx, None, 33, 0, stride=608, reuse=True)
        t1 = time.perf_counter()
        r = eng.filter(q, lens=lens)
        return r.n_pass, t1 - t0, time.perf_counter() - t1, float(lens.mean())
    back(front(0)[0])                                        # untimed: thread pools, page faults of the work buffers
    # pass 1: one stage after the other (stage rates)
    t_index = t_contig = t_pack = t_filter = 0.0
    kept = 0
    t_all = time.perf_counter()
    n_stage = chunks if stage_chunks is None else max(1, min(chunks, stage_chunks))
    for k in range(n_stage):
        t0 = time.perf_counter()
        cb, t_c0 = front(k)
        t1 = time.perf_counter()
        t_index += t_c0 - t0
        t_contig += t1 - t_c0
        np_, tp, tf, mean_len = back(cb)
        kept += np_
        t_pack += tp
        t_filter += tf
    seq_wall = time.perf_counter() - t_all
    # pass 2: pipelined (front of chunk k+1 beside back of chunk k)
    res = {}

    def worker(k):
        res[k] = front(k)[0]
    t_all = time.perf_counter()
    th = threading.Thread(target=worker, args=(0,))
    th.start()
    kept2 = 0
    for k in range(chunks):
        th.join()
        cb = res.pop(k)
        if k + 1 < chunks:
            th = threading.Thread(target=worker, args=(k + 1,))
            th.start()
        kept2 += back(cb)[0]
    pipe_wall = time.perf_counter() - t_all
    total = n * chunks
    stage_total = n * n_stage
    assert kept2 * n_stage == kept * chunks
    rate = total / pipe_wall
    return {"note": "BASELINE configs[2] at bench size: synthetic 2 x 300-base pairs (150 bases of overlap) as FASTQ text in host "
                    "memory -> index -> NW + consensus on the host cores (north_star keeps contig construction on the CPU) -> "
                    "pack -> GPU filter from host memory; NOT the headline (it is bound by the host stages, not by the GPU)",
            "pairs": total, "chunks": chunks, "host_threads": threads, "mean_contig_length": mean_len,
            "contigs_kept": kept2,
            "stage_pairs_per_s": {"index_both_files": stage_total / t_index, "contig_construction": stage_total / t_contig,
                                  "pack": stage_total / t_pack, "gpu_filter_incl_pcie": stage_total / t_filter},
            "stages_one_after_the_other": {"wall_s": seq_wall, "pairs_per_s": stage_total / seq_wall, "pairs": stage_total},
            "pipelined": {"wall_s": pipe_wall, "pairs_per_s": rate,
                          "note": "index + contigs of chunk k+1 on a second thread while chunk k is packed and filtered"},
            "projected_wall_s_for_100M_pairs": 1e8 / rate,
            "gpu_share_of_the_pipelined_wall": (t_filter / n_stage * chunks) / pipe_wall}


def classified_rate(eng, d_q, n, stride, L, params, d_ee, d_ns, d_pass):
    """Round 3 (SURVEY f-4): the batch arrives as raw FASTQ text resident in HBM (built here from the workload's packed
    matrix by mpb_encode_ascii_device, not timed).  Two ways to results: decode, then the ordinary filter (the packed
    matrix is written, then read by the prepass, then by the DP) -- or classified at source (the decode pass classifies,
    the filter starts at the scan: written once, read once).  Same results bit for bit
    (tests/test_gpu_classified.py).  A different input form than the metric's (text, 2 bytes per base): NOT the headline."""
    out = {"note": "raw FASTQ text resident in HBM -> results: mpb_decode_ascii_device + mpb_filter_device against "
                   "mpb_decode_classify_device + mpb_filter_device_classified (no k_prepass launch); NOT the headline", "reads": n}
    bufs = []
    try:
        bufs = [eng.alloc(n * stride) for _ in range(3)]
        d_seq, d_qual, d_out = bufs
        eng.encode_ascii_device(d_q, n, stride, d_seq, d_qual)
        eng.synchronize()

        def two_pass():
            eng.decode_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L)
            eng.filter_device(d_out, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)

        def at_source():
            eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                    params=params, want_counts=False)
        res = {}
        for name, fn in (("decode_then_filter", two_pass), ("classified_at_source", at_source)):
            for _ in range(2):
                fn()
            eng.synchronize()
            k = 20
            t = time.perf_counter()
            for _ in range(k):
                fn()
            eng.synchronize()
            dt = (time.perf_counter() - t) / k
            eng.timing(True)
            eng.timing_reset()
            for _ in range(5):
                fn()
            kt = {kk: v[0] / max(v[1], 1) for kk, v in eng.kernel_times().items() if v[1]}
            eng.timing(False)
            res[name] = {"ms_per_step": dt * 1e3, "reads_per_s": n / dt, "kernels_ms": kt}
        c = eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params)
        out.update(res)
        out["pass"] = c.n_pass
        out["saved_ms_per_step"] = res["decode_then_filter"]["ms_per_step"] - res["classified_at_source"]["ms_per_step"]
        out["kernels_note"] = ("kernels_ms.prepass of classified_at_source is the fused decode + classify pass (the plain decode "
                               "of decode_then_filter is not event-timed: it is the difference of the two step times minus the prepass)")
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    for b in bufs:
        try:
            b.free()
        except Exception:
            pass
    return out


def long_ragged_rate(eng, params, n=1_000_000, stride=2048):
    """Round 3: ragged 50-2,000 bp reads (full-length-16S territory) in one stride-2048 matrix, resident: the tile
    classes on long rows, the prepass' panel loop and -- for the reads that need more than 1024 DP rows -- k_wide.
    A parity-test case (tests/test_gpu_long_reads.py), reported for reference; NOT the headline."""
    out = {"note": "lengths U{50..2000} in one stride-2048 matrix (synthetic quality model of include/mpb_synth.h), resident, "
                   "bit-exact mode; NOT the headline", "reads": n}
    bufs = []
    try:
        bufs = [eng.alloc(n * stride), eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
        d_q, d_len, d_ee, d_ns, d_pass = bufs
        eng.synth_fill(d_q, n, stride, fixed_len=0, min_len=50, max_len=2000, d_len=d_len, seed=7)
        run = lambda c=False: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                                params=params, want_counts=c)
        for _ in range(2):
            run()
        eng.synchronize()
        k = 5
        t = time.perf_counter()
        for _ in range(k):
            run()
        eng.synchronize()
        dt = (time.perf_counter() - t) / k
        c = run(True)
        hist = eng.class_histogram()
        out.update({"ms_per_step": dt * 1e3, "reads_per_s": n / dt, "mean_length": 1025, "bases_per_s": n * 1025 / dt,
                    "pass": c.n_pass, "overflow_reruns": c.n_overflow,
                    "reads_in_wide_kernel": int(n - sum(hist.values()))})
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    for b in bufs:
        try:
            b.free()
        except Exception:
            pass
    return out


def config4_shard_rate(eng, L, stride, seed, params, n=CONFIG4_SHARD, rank=3):
    """What ONE GPU of BASELINE configs[3] does, timed in the same (driver-run) process: a 125 M-read shard (40 GB
    resident, read ids rank*n ..) -- the per-GPU workload of `bench.py --gpus 8`.  NOT the headline of an N = 1 run."""
    out = {"note": "one 125 M-read shard of configs[3] (1 B reads over 8 GPUs) on this GPU: the per-rank workload of "
                   "--gpus 8, >= 1 s of steps; NOT the headline", "reads": n}
    bufs = []
    try:
        bufs = [eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
        d_q, d_ee, d_ns, d_pass = bufs
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed, first_read=rank * n)
        run = lambda c=False: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                                params=params, want_counts=c)
        for _ in range(3):
            run()
        eng.synchronize()
        k = 22
        t = time.perf_counter()
        for _ in range(k):
            run()
        eng.synchronize()
        dt = (time.perf_counter() - t) / k
        c = run(True)
        out.update({"ms_per_step": dt * 1e3, "reads_per_s": n / dt, "steps": k, "pass": c.n_pass, "overflow_reruns": c.n_overflow})
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    for b in bufs:
        try:
            b.free()
        except Exception:
            pass
    return out


def _decode_rank(text):
    """one rank's {"k": kernel ms per step, "mhz": [mean, min, max] held clock} as gathered by Collectives.gather_text"""
    try:
        d = json.loads(text)
        mhz = d.get("mhz")
        return {"kernels_ms_per_step": d.get("k"),
                "held_clock": {"mean_mhz": mhz[0], "min_mhz": mhz[1], "max_mhz": mhz[2]} if mhz else None}
    except ValueError:
        return {"undecodable": text}


def _wall_rate(eng, run, seconds=0.6, settle_s=0.5):
    """ms per call of `run` (asynchronous calls back to back, one synchronisation at the end) after `settle_s` of untimed calls."""
    run(); eng.synchronize()
    t = time.perf_counter(); run(); eng.synchronize()
    one = max(time.perf_counter() - t, 1e-5)
    for _ in range(max(3, int(settle_s / one))):
        run()
    eng.synchronize()
    k = max(5, int(seconds / one))
    t = time.perf_counter()
    for _ in range(k):
        run()
    eng.synchronize()
    return (time.perf_counter() - t) / k * 1e3, k


def high_quality_rate(eng, n, stride, L, seed, d_ee, d_ns, d_pass):
    """the HBM-bound regime.  The same shape as configs[1] (n x 300 bp, stride 320, resident) with the clean
    quality profile of include/mpb_synth.h (profile 1: Q33..Q40, 0.003 % ambiguous bases: every read's CDF crosses 1 - alpha
    on the second row of the table).  The library picks its pass from a sample of <= 0.1 % of the reads (mpb_path_info): here
    the natural-order narrow pass (k_narrow_rs / k_narrow, the matrix read once) -- timed against the sorted pipeline on the same batch
    (MPB_FLAG_NO_NARROW).  Algorithmic bytes per read = L + 13 as everywhere.  NOT the headline."""
    out = {"note": "10 M x 300 bp of the CLEAN synthetic profile (Q33..Q40; include/mpb_synth.h profile 1), resident; the pass "
                   "is the library's own choice; roofline as for the headline: (L + 13) x reads / time / 8 TB/s; NOT the headline",
           "reads": n, "read_length": L, "row_stride": stride, "profile": 1, "seed": seed}
    d_q = None
    try:
        d_q = eng.alloc(n * stride)
        eng.synth_fill(d_q, n, stride, fixed_len=L, seed=seed, profile=1)
        alg = n * (L + 13)
        prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors")
        run = lambda p=prm, c=False: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                                      params=p, want_counts=c)
        run()
        first = eng.last_path()
        ms, k = _wall_rate(eng, run)
        path = eng.last_path()
        c = run(c=True)
        eng.timing(True); eng.timing_reset()
        for _ in range(10):
            run()
        kt = {name: v[0] / 10 for name, v in eng.kernel_times().items() if v[1]}
        eng.timing(False)
        nar_ms = kt.get("narrow")
        out.update({"ms_per_step": ms, "steps": k, "reads_per_s": n / ms * 1e3,
                    "pass_taken": {"narrow_rows": path["narrow_rows"], "reads_handed_to_the_sorted_pipeline": path["n_fallback"],
                                   "sample_rows_histogram": {str(r): v for r, v in enumerate(first["sample_hist"]) if v},
                                   "sample_reads": sum(first["sample_hist"])},
                    "kernels_ms_per_step": kt,
                    "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": alg,
                                 "whole_step": {"achieved": alg / ms / 1e6, "frac": alg / ms / 1e6 / HBM_PEAK_GBS},
                                 "kernel": "k_narrow_rs" if stride % 64 == 0 and not os.environ.get("MPB_NAR_NO_RS") else "k_narrow",
                                 "avg_launch_ms": nar_ms,
                                 "achieved": (alg / nar_ms / 1e6) if nar_ms else None,
                                 "frac": (alg / nar_ms / 1e6 / HBM_PEAK_GBS) if nar_ms else None},
                    "outcome": {"pass": c.n_pass, "fail": c.n_fail, "overflow_reruns": c.n_overflow}})
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic_hq.json")
        if os.path.exists(tpath):
            t = json.load(open(tpath))
            w = t.get("workload", {})
            if (w.get("reads"), w.get("length"), w.get("seed"), w.get("profile")) == (n, L, seed, 1):
                out["roofline"]["traffic"] = t.get("hbm_bytes_per_launch")
                out["roofline"]["traffic_source"] = t.get("source")
                out["valu_busy_pmc"] = t.get("valu")
        # the same batch through the sorted pipeline (what round 4 did with it)
        prm_s = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", no_narrow=True)
        ms_s, k_s = _wall_rate(eng, lambda: run(prm_s), seconds=0.3, settle_s=0.2)
        hist = eng.class_histogram()
        out["sorted_pipeline_on_the_same_batch"] = {"ms_per_step": ms_s, "steps": k_s, "frac_whole_step": alg / ms_s / 1e6 / HBM_PEAK_GBS,
                                                    "row_budget_histogram": {str(a): b for a, b in hist.items() if b}}
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    if d_q is not None:
        try:
            d_q.free()
        except Exception:
            pass
    return out


def real_profile_batches(eng, rows_fixed=10_000_000, rows_ragged=5_000_000, seed=7):
    """The reference's OWN reads as resident batches: (a) the 1,000 reads of moira/test/test1.fastq (251 bp;
    tests/golden/test1.fastq.gz is that file) and (b) the 400 representative contigs of its paired golden run
    (tests/golden/reference_test_results/paired.qc.{good,bad}: 241-502 bp), each tiled to millions of rows in a random order
    (a read's copies are never adjacent on purpose: the order is a seeded permutation of the tiling).
    -> [(label, q uint8[n, stride], lens or None, fixed_len, source_index int32[n], unique_q, unique_lens)]"""
    import gzip
    import numpy as np
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import golden_io as G
    rng = np.random.default_rng(seed)
    out = []
    lines = gzip.open(os.path.join(ROOT, "tests", "golden", "test1.fastq.gz"), "rt").read().split("\n")
    seqs, quals = [lines[i + 1] for i in range(0, len(lines) - 3, 4)], [lines[i + 3] for i in range(0, len(lines) - 3, 4)]
    uq, ul = eng.pack_batch_ascii(seqs, quals, fastq_offset=33, stride=256)
    idx = rng.permutation(np.arange(rows_fixed, dtype=np.int64) % len(seqs)).astype(np.int32)
    out.append(("test1.fastq (1,000 reads x 251 bp)", uq[idx], None, int(ul[0]), idx, uq, ul))
    recs = []
    for kind in ("good", "bad"):
        recs += G.read_fasta_qual(os.path.join(ROOT, "tests", "golden", "reference_test_results", "paired.qc." + kind))
    cq, cl = eng.pack([r[2] for r in recs], [r[3] for r in recs], stride=512)
    idx = rng.permutation(np.arange(rows_ragged, dtype=np.int64) % len(recs)).astype(np.int32)
    out.append(("paired golden contigs (400 representatives, 241-502 bp)", cq[idx], cl[idx], 0, idx, cq, cl))
    return out


def real_profile_rate(eng):
    """Throughput on the reference's own quality profiles, next to the synthetic headline.  NOT the headline."""
    import numpy as np
    out = {"note": "the reference's own reads (moira/test/test1.fastq; the contigs of its paired golden run) tiled in a seeded "
                   "random order to resident batches; bit-exact mode, the library's own choice of pass; NOT the headline"}
    try:
        for label, q, lens, fixed_len, _idx, _uq, _ul in real_profile_batches(eng):
            n, stride = q.shape
            bufs = [eng.alloc(n * stride), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)] + ([eng.alloc(n * 4)] if lens is not None else [])
            try:
                bufs[0].upload(q)
                if lens is not None:
                    bufs[4].upload(lens)
                run = lambda c=False: eng.filter_device(bufs[0], n, stride, d_len=bufs[4] if lens is not None else None,
                                                        fixed_len=fixed_len, d_ee=bufs[1], d_ns=bufs[2], d_pass=bufs[3], want_counts=c)
                ms, k = _wall_rate(eng, run, seconds=0.4, settle_s=0.3)
                path = eng.last_path()
                c = run(True)
                hist = eng.class_histogram() if path["narrow_rows"] == 0 else {}
                eng.timing(True); eng.timing_reset()
                for _ in range(5):
                    run()
                kt = {name: v[0] / 5 for name, v in eng.kernel_times().items() if v[1]}
                eng.timing(False)
                L = fixed_len if lens is None else float(lens.mean())
                out[label] = {"reads": n, "row_stride": stride, "mean_length": L, "ms_per_step": ms, "steps": k,
                              "reads_per_s": n / ms * 1e3, "bases_per_s": n * L / ms * 1e3,
                              "pass_taken": {"narrow_rows": path["narrow_rows"], "handed_back": path["n_fallback"]},
                              "kernels_ms_per_step": kt, "row_budget_histogram": {str(a): b for a, b in hist.items() if b},
                              "outcome": {"pass": c.n_pass, "fail": c.n_fail, "overflow_reruns": c.n_overflow}}
            finally:
                for b in bufs:
                    b.free()
            del q
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    return out


def poisson_rate(eng, d_q, n, stride, L, d_lam, d_ns):
    """--error_calc poisson on the same resident batch: the device part is a pure streaming
    reduction (per-read in-order sum of error probabilities, k_lambda) -- the one kernel of the path that IS
    HBM-bound; the scalar CDF tail stays on the host (same libm exp / pow as the reference).  NOT the headline."""
    import ctypes as C
    import numpy as np
    from moira_amd import _lib as ML
    out = {"note": "device part of --error_calc poisson on the resident batch of rank 0 (k_lambda, HIP events) and the host "
                   "tail on the CPUs this box grants; NOT the headline"}
    try:
        for _ in range(2):
            ML.check(eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, None, L, d_lam.ptr, d_ns.ptr))
        eng.timing(True)
        eng.timing_reset()
        for _ in range(5):
            ML.check(eng.lib.mpb_poisson_lambda_device(eng.ctx, d_q.ptr, n, stride, None, L, d_lam.ptr, d_ns.ptr))
        ms, cnt = eng.kernel_times()["lambda"]
        eng.timing(False)
        ms /= max(cnt, 1)
        out["k_lambda"] = {"ms_per_launch": ms, "reads_per_s": n / ms * 1e3,
                           "algorithmic_GBps": n * (L + 12) / ms / 1e6, "frac_of_hbm_peak": n * (L + 12) / (ms * 1e-3) / 8e12}
        m = min(n, 4_000_000)
        lam, ns = d_lam.download(np.float64, m), d_ns.download(np.int32, m)
        ee, ps = np.empty(m), np.empty(m, np.uint8)
        prm = eng.params()
        t = time.perf_counter()
        ML.check(eng.lib.mpb_poisson_finish_host(lam.ctypes.data, ns.ctypes.data, None, L, m, C.byref(prm), ee.ctypes.data, ps.ctypes.data))
        dt = time.perf_counter() - t
        out["host_tail"] = {"reads_per_s": m / dt, "reads": m, "pass": int(ps.sum())}
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    return out


def per_read_in_process_rate(eng, calls=4000):
    """bernoulli.calculate_errors_PB(contig, contig_quals, alpha) called read by read from THIS process (what moira.py
    --processors 1 does, moira/moira.py:817): since round 5 the context keeps the one-read kernel resident while such calls come
    (k_serve, one mailbox entry in pinned host memory: no launch per call; MPB_SERVE=0 switches it off).  Never `value`."""
    import numpy as np
    out = {"note": "calculate_errors_PB per read from one Python process on the bench's own context (300-base reads); "
                   + ("resident one-read kernel, no launch per call" if os.environ.get("MPB_SERVE", "1") != "0" else "a k_small launch per call (MPB_SERVE=0)")
                   + "; the reference extension per read from Python is cpu_baseline (1 core); NOT the headline"}
    try:
        rng = np.random.default_rng(1)
        seq = "".join(rng.choice(list("ACGT"), 300))
        quals = [int(x) for x in np.clip(38 - (np.arange(300) / 300) ** 3 * 20 - rng.integers(0, 6, 300), 2, 40)]
        for _ in range(50):
            eng.calculate_errors_PB(seq, quals, 0.005)
        t = time.perf_counter()
        for _ in range(calls):
            eng.calculate_errors_PB(seq, quals, 0.005)
        dt = time.perf_counter() - t
        out.update({"calls": calls, "us_per_call": dt / calls * 1e6, "calls_per_s": calls / dt})
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    return out


def per_read_broker_rate():
    """What an UNCHANGED moira.py --processors P gets from the drop-in module (moira/moira.py:398-399,431-454: Pool workers
    calling bernoulli.calculate_errors_PB per read): P = the granted CPUs worker processes through ONE GPU-owning broker
    process (moira_amd/broker.py).  Runs tools/per_read_concurrency.py as a child process; never `value`."""
    server = os.environ.get("MPB_BROKER_SERVER", "1") != "0"
    out = {"note": "P worker processes call bernoulli.calculate_errors_PB per read (300-base reads) through the broker: one "
                   "GPU-owning process serves them -- " + ("a resident kernel (k_serve), a wave per worker slot polling its mailbox "
                   "entry in pinned host memory: no launch per call" if server else "micro-batches of what they have pending, a "
                   "launch each (MPB_BROKER_SERVER=0)") + "; the reference's own extension on the same cores "
                   "is cpu_baseline.all_cores; NOT the headline",
           "serving": "resident kernel" if server else "launch per micro-batch"}
    try:
        from moira_amd.contig import usable_cpus
        p = max(2, min(16, usable_cpus()))
        env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MOIRA_PB_BROKER")}
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "per_read_concurrency.py"), "--json", "1", str(p)],
                           capture_output=True, text=True, timeout=240, env=env, cwd=ROOT)
        rows = json.loads([l for l in r.stdout.splitlines() if l.startswith("[")][-1])
        out["one_worker"] = {"calls_per_s": rows[0]["calls_per_s"], "us_per_call": rows[0]["us_per_call_in_a_worker"]}
        out["workers"] = rows[1]["workers"]
        out["calls_per_s"] = rows[1]["calls_per_s"]
        out["us_per_call_in_a_worker"] = rows[1]["us_per_call_in_a_worker"]
        b = rows[1]["broker"] or {}
        if b.get("batches"):
            out["reads_per_launch_since_the_broker_started"] = b["served"] / (b["batches"] + b["solo"])
            out["launches_since_the_broker_started"] = b["batches"]
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    return out


def host_fed_rate(eng, L, stride, seed, n=8_000_000):
    """PCIe-inclusive rate of the host-buffer entry (mpb_filter_host), never `value`: packed reads in host
    memory in, ee / Ns / pass in host memory out, through the pinned double-buffered pipeline."""
    import numpy as np
    out = {"note": "mpb_filter_host on %d reads of the workload: H2D + kernels + D2H overlapped; PCIe-inclusive, "
                   "NOT the headline" % n, "reads": n}
    try:
        d = eng.alloc(n * stride)
        eng.synth_fill(d, n, stride, fixed_len=L, seed=seed)
        for kind in ("pinned", "pageable"):
            if kind == "pinned":
                q = eng.host_alloc((n, stride), np.uint8)
            else:
                q = np.empty((n, stride), np.uint8)
            q.reshape(-1)[:] = d.download(np.uint8, n * stride)
            eng.filter(q[:200000], fixed_len=L)
            res = (np.zeros(n), np.zeros(n, np.int32), np.zeros(n, np.uint8))     # reused result arrays
            best = None
            for _ in range(3):
                t = time.perf_counter()
                eng.filter(q, fixed_len=L, out=res)
                dt = time.perf_counter() - t
                best = dt if best is None else min(best, dt)
            out[kind + "_source"] = {"reads_per_s": n / best, "qscore_GBps": n * stride / best / 1e9}
            if kind == "pinned":
                eng.host_free(q)
            del q
        d.free()
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    return out


if __name__ == "__main__":
    sys.exit(main() or 0)
