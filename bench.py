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

Output (rank 0): the headline as ONE JSON line as soon as the timed region, the event pass and cpu_baseline are done -- before
anything optional runs -- and, at N = 1, the same line again with the live counters (`pmc_live`, roofline.traffic: rocprofv3 --pmc
passes in fresh child processes over tools/pmc_probe.py) and with `extras` (tools/bench_extras.py) as the LAST line: a run that is
killed inside an extra still leaves a complete headline on stdout.  Extra objects of the headline:
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

sys.path.insert(0, os.path.join(ROOT, "tools"))
# launch shapes, collectives, the step plan (tools/bench_launch.py) and the extras (tools/bench_extras.py: imported lazily --
# nothing in it runs before the headline is on stdout)
from bench_launch import (HBM_PEAK_GBS, FP64_VALU_PEAK, CONFIG2_READS, CONFIG4_SHARD, Collectives, StubEngine,  # noqa: E402,F401
                          ClockSampler, pci_bus_id_of, plan_steps, self_launch, threads_main, visible_gpus, _decode_rank)


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

def __getattr__(name):
    """`bench.config3_paired_rate`, `bench.real_profile_batches`, ...: the extras live in tools/bench_extras.py (PEP 562)."""
    import bench_extras
    try:
        return getattr(bench_extras, name)
    except AttributeError:
        raise AttributeError("module 'bench' has no attribute %r" % name) from None


def committed_traffic(n, L, seed):
    """roofline.traffic / valu_busy_pmc as committed under profiles/ (builder-side PMC runs): only for the workload they were
    measured on.  Replaced by the live counters when bench.py's own PMC passes succeed (traffic_source says which)."""
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tpath):
        t = json.load(open(tpath))
        w = t.get("workload", {})
        if (w.get("reads"), w.get("length"), w.get("seed")) == (n, L, seed):
            return t["hbm_bytes_per_launch"], t.get("valu"), t.get("source")
    return None, None, None


def emit(line):
    """One JSON line on stdout, flushed at once: the driver parses the LAST line that is JSON; the first one already carries the
    whole headline (everything but `extras` and the live counters), so a run that is killed inside an extra still leaves it."""
    sys.stdout.write(json.dumps(line) + "\n")
    sys.stdout.flush()


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
    ap.add_argument("--no-extras", action="store_true", help="the headline only: no extras, no live counter passes (profiling)")
    ap.add_argument("--no-live-pmc", action="store_true", help="keep the committed profiles/ counters (no rocprofv3 child passes)")
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
    ap.add_argument("--_stub-extras", type=float, default=0.0, help=argparse.SUPPRESS)      # tests (CPU rehearsal, N = 1): after the headline, "extras" that sleep this many seconds each
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
            line = {"metric": "reads/sec filtered (300 bp synthetic)", "value": n * world * steps / dt, "unit": "reads/s",
                    "n_gpus": world, "steps": steps, "warmup": args.warmup, "settle_steps": settle,
                    "ms_per_step": dt / steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
                    "dtype": "f64", "data": "none (CPU rehearsal of the launch / collective code: the step is a stub)",
                    "config": {"workload": "REHEARSAL ON CPU -- not a measurement", "reads_per_gpu": n,
                               "world_size": world,
                               "collective_backend": "gloo (rehearsal)" if not args._try_rccl else
                               "gloo (RCCL not used: %s)" % coll.rccl_error},
                    "t_step_rank_uniform_s": t_step, "timed_region_s": dt, "reads_per_s_per_rank": per_rank,
                    "devices": devices, "outcome": {"pass": n_pass, "fail": n_fail, "overflow_reruns": n_ovf}}
            emit(line)
            if args._stub_extras > 0 and world == 1:
                # tests: the order of a real N = 1 run -- the headline is out, the extras follow, the full line closes
                line["extras"] = {}
                for k in range(3):
                    sys.stderr.write("bench.py: extra stub_%d ...\n" % k)
                    time.sleep(args._stub_extras)
                    line["extras"]["stub_%d" % k] = {"slept_s": args._stub_extras}
                emit(line)
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
    # row budgets and algorithmic cells: one untimed pass through the sorted pipeline (what the headline batch takes by itself;
    # forced, because the histogram describes the sorted pipeline's classes) that also sums the algorithmic DP cells on the device
    prm_cells = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma, count_cells=True, no_narrow=True)
    eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_cells, want_counts=False)
    hist = eng.class_histogram()
    alg_cells = eng.algorithmic_cells()

    line = None
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
        traffic, valu_pmc, pmc_source = (None, None, None) if args.fast_fma else committed_traffic(n, L, args.seed)
        step_s = dt / steps
        fp64_floor_ms = cells * opc / FP64_VALU_PEAK * 1e3
        hbm40_ms = alg_bytes / (0.40 * HBM_PEAK_GBS * 1e9) * 1e3
        if world == 1:
            wl = ("BASELINE configs[1]: %d synthetic single-end %d bp reads, poisson_binomial filter, alpha 0.005, "
                  "uncert 0.01, resident in HBM (uint8 %d x %d, seed %d)" % (n, L, n, stride, args.seed))
            if n == CONFIG4_SHARD:
                wl = ("one shard of BASELINE configs[3] on one GPU (the N = 1 point of the weak-scaling curve): " + wl.split(": ", 1)[1])
            anchor = ("this N = 1 line times configs[1] (%d reads); runs with N > 1 time %d reads per GPU (configs[3]).  The "
                      "single-GPU point of that per-GPU workload is extras.config4_shard of this line, or "
                      "`python bench.py --gpus 1 --reads %d`" % (n, CONFIG4_SHARD, CONFIG4_SHARD))
        else:
            wl = ("BASELINE configs[3]: %d synthetic %d bp reads sharded host-side across %d x MI355X, %d reads "
                  "(%.1f GB resident, generated on device, read ids rank*R..) per GPU, poisson_binomial filter, "
                  "alpha 0.005, uncert 0.01 (uint8 %d x %d per GPU, seed %d)"
                  % (n * world, L, world, n, n * stride / 1e9, n, stride, args.seed))
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
                         "traffic_unit": "bytes per k_dp launch (L2 memory-side request counters, PMC)",
                         "traffic_source": ("committed: " + pmc_source) if pmc_source else None,
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
                          "valu_busy_pmc": valu_pmc,
                          "sources": {"avg_launch_ms, cells_*, frac*": "measured live in this run (HIP events on the library's "
                                                                      "stream; class histogram; MPB_FLAG_COUNT_CELLS pass)",
                                      "valu_busy_pmc, roofline.traffic": "see roofline.traffic_source",
                                      "kernel average to compare avg_launch_ms with": "profiles/r06_kernel_stats.csv "
                                                                                      "(rocprofv3 --kernel-trace --stats)"},
                          "note": "the binding roof: a scalar FP64 recurrence (SURVEY §8d); `frac` counts 3 ops for every cell of the row-budget classes, `frac_issued` the FP64 instructions actually issued (row 0 of a one-lane class is one multiply), `frac_algorithmic` 3 ops for every cell the algorithm needs"},
            "kernels_ms_per_step": {k: v[0] / max(v[1], 1) for k, v in times.items()},
            "kernel_event_pass": {"steps": ev_steps, "note": "HIP events on the library's stream, separate from the wall-clock region"},
            "outcome": {"pass": n_pass, "fail": n_fail, "overflow_reruns": n_ovf},
            "row_budget_histogram": {str(k): v for k, v in hist.items() if v},
        }
        if world == 1 and not args.no_cpu_baseline:
            line.update(cpu_baseline(args.seed, L, stride))
        # ---- the headline is complete: out it goes, before anything optional runs ----
        emit(line)
        # N = 1 only: with more ranks the others would sit in the closing barrier (120 s timeout) while rank 0 runs the extras,
        # and they describe the single-GPU library, not the multi-GPU run
        if world == 1 and not args.no_extras:
            import bench_extras as X
            if not args.no_live_pmc and not args.fast_fma:
                sys.stderr.write("bench.py: live counter passes (rocprofv3 --pmc in fresh child processes) ...\n")
                live = X.pmc_live(n, L, args.seed, stride)
                line["pmc_live"] = live
                k = (live.get("kernels") or {}).get("k_dp<false, false>")
                if k and k.get("hbm_bytes_per_launch"):
                    line["roofline"].update({"traffic": k["hbm_bytes_per_launch"], "traffic_read_bytes": k["hbm_read_bytes_per_launch"],
                                             "traffic_write_bytes": k["hbm_write_bytes_per_launch"],
                                             "traffic_source": "live: rocprofv3 --pmc TCC_EA0_RDREQ_{32B,64B,128B} / TCC_EA0_WRREQ{,_64B} "
                                                               "passes started by this run (tools/pmc_probe.py, same workload), "
                                                               "request counts x their sizes"})
                    if k.get("valu"):
                        line["fp64_valu"]["valu_busy_pmc"] = dict(k["valu"], source="live (this run's SQ pass)")
                emit(line)

            def progress(name):
                sys.stderr.write("bench.py: extra %s ...\n" % name)
                sys.stderr.flush()
            line["extras"] = X.run_all(eng, args, d_q, n, stride, L, params, d_ee, d_ns, d_pass, dp_avg_s * 1e3 if dp_n else None, progress)
            hq = line["extras"].get("high_quality_300", {})
            live_k = (line.get("pmc_live", {}).get("kernels") or {})
            for name, e in live_k.items():                      # the narrow passes' live traffic next to their extras
                if name.startswith("k_narrow_rs") or name.startswith("k_narrow<"):
                    if "roofline" in hq:
                        hq["roofline"].update({"traffic": e.get("hbm_bytes_per_launch"), "traffic_source": "live (pmc_live.kernels[%r])" % name})
                if name.startswith("k_narrow_rg"):
                    hr = line["extras"].get("high_quality_ragged", {})
                    if "roofline" in hr:
                        hr["roofline"].update({"traffic": e.get("hbm_bytes_per_launch"), "traffic_source": "live (pmc_live.kernels[%r]): k_narrow_rg alone" % name})
            emit(line)
    for b in (d_q, d_ee, d_ns, d_pass):
        b.free()
    eng.close()
    coll.close()


if __name__ == "__main__":
    sys.exit(main() or 0)
