#!/usr/bin/env python3
"""bench.py -- reads/s of the MI355X Poisson-binomial read filter on BASELINE.json's config.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--reads R]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Workload (config.workload): BASELINE.json configs[1] -- R = 10,000,000 synthetic single-end
300 bp reads per GPU (row stride 320, seed 2, counter-based generator of include/mpb_synth.h,
filled ON DEVICE so the inputs are resident in HBM when the timed region starts).  With N > 1
every rank holds its own R reads (read ids rank*R .. rank*R+R-1: the host-side split of
config 4), no data-path collective: weak scaling.
A "step" = one pass of the whole hot path over the resident batch: prepass -> scan -> scatter
-> DP -> overflow pass, producing ee / Ns / pass for every read.

One JSON line on rank 0.  Extra objects:
  roofline     dominant kernel (k_dp): algorithmic bytes (L + 13 per read) per launch
               / its mean duration measured with HIP events on the library's stream.
  cpu_baseline the real reference extension (oracle/_ref, kind "reference") called per read from
               Python exactly as moira.py does with --processors 1, or the oracle's
               reference-shaped port (kind "port"), on a bounded sample of the same reads.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8 TB/s spec
FP64_VALU_PEAK = 39.3e12       # v_mul/add_f64 lane-ops per second: 256 CU x 4 SIMD x 16 lanes/clk x 2.4 GHz


def cpu_baseline(seed, L, stride, budget_s=15.0):
    """Time the CPU path on a bounded sample of the same workload (rank 0, N=1 only)."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import pb_oracle as O
    O.build()
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
    # fast restatement on all cores (for orientation only)
    nt = O.lib().pbo_max_threads()
    n_fast = min(len(q), 400000)
    t = time.perf_counter(); O.filter_batch(q[:n_fast], fixed_len=L, shape=0, threads=nt); dt = time.perf_counter() - t
    out["cpu_fast_restatement"] = {"value": n_fast / dt, "unit": "reads/s", "cores": nt,
                                   "note": "two-term recurrence (not the reference's algorithmic shape)"}
    ref = O.reference_module()
    if ref is not None:
        n_ref = int(max(500, min(200000, budget_s * 0.5 * port["value"] / 1.6)))
        rows = q[:n_ref, :L]
        seqs = ["".join("N" if v == 0 else "A" for v in r) for r in rows]
        quals = [[int(v) if v else 20 for v in r] for r in rows]
        t = time.perf_counter()
        for s, qq in zip(seqs, quals):
            ref.calculate_errors_PB(s, qq, 0.005)
        dt = time.perf_counter() - t
        out["cpu_baseline"] = {"value": n_ref / dt, "unit": "reads/s", "cores": 1, "kind": "reference",
                               "sample": "first %d reads of the workload through oracle/_ref/bernoulli.so "
                                         "(moira/bernoullimodule.c built unmodified), one Python call per read "
                                         "as moira.py --processors 1 does" % n_ref}
        out["cpu_port"] = port
    else:
        out["cpu_baseline"] = port
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--reads", type=int, default=10_000_000, help="reads per GPU")
    ap.add_argument("--length", type=int, default=300)
    ap.add_argument("--seed", type=int, default=2)
    ap.add_argument("--fast-fma", action="store_true", help="non-bit-exact FMA mode (not the headline)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--stride", type=int, default=0, help="row stride in bytes (default: length rounded up to 64; experiments)")
    ap.add_argument("--no-extras", action="store_true", help="skip the opt-in-mode extra runs (profiling)")
    ap.add_argument("--rehearse-on-one-gpu", action="store_true",
                    help="multi-rank dry run on a 1-GPU box: every rank uses device 0 and the (tiny) "
                         "collectives go over gloo; exercises the N>1 code path, not a scaling number")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from moira_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d "
                             "--master-addr 127.0.0.1 bench.py --gpus %d ..." % (args.gpus, args.gpus))
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    if args.rehearse_on_one_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    coll_dev = "cpu" if args.rehearse_on_one_gpu else "cuda"
    if world > 1:
        if args.rehearse_on_one_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    n, L = args.reads, args.length
    stride = args.stride or (L + 63) // 64 * 64          # 300 -> 320
    eng = Engine(local_rank)
    d_q = eng.alloc(n * stride)
    d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
    eng.synth_fill(d_q, n, stride, fixed_len=L, seed=args.seed, first_read=rank * n)
    params = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma)

    def step(counts=False):
        return eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass,
                                 params=params, want_counts=counts)

    for _ in range(args.warmup):
        step()
    eng.synchronize()
    eng.timing(True)
    eng.timing_reset()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    eng.synchronize()
    barrier()
    dt = time.perf_counter() - t0
    times = eng.kernel_times()
    eng.timing(False)
    counts = step(counts=True)
    hist = eng.class_histogram()
    # extra (NOT the headline, work is skipped by design): opt-in MPB_FLAG_DECISION_ONLY, same batch
    extras = {}
    if not args.no_extras:
        prm_do = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=args.fast_fma, decision_only=True)
        for _ in range(2):
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_do, want_counts=False)
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_do, want_counts=False)
        eng.synchronize()
        dt_do = (time.perf_counter() - t1) / 5
        counts_do = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_do)
        hist_do = eng.class_histogram()
        extras = {"decision_only_mode": {
            "note": "opt-in MPB_FLAG_DECISION_ONLY on the same resident batch of rank 0: reads proven to fail "
                    "(Chernoff bound) skip their DP and report ee=NaN; identical pass/fail flags; NOT the headline",
            "reads_per_s_this_rank": n / dt_do, "ms_per_step": dt_do * 1e3,
            "pass": counts_do.n_pass, "reads_run_through_dp": int(sum(hist_do.values()))}}
        if not args.fast_fma:
            # opt-in MPB_FLAG_FAST_FMA: 2 FP64 ops per DP cell instead of 3; ee within 1e-9 relative (north_star's
            # tolerance), NOT bit-identical, so not the default and not the headline
            prm_f = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=True)
            for _ in range(2):
                eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_f, want_counts=False)
            eng.synchronize()
            t1 = time.perf_counter()
            for _ in range(5):
                eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_f, want_counts=False)
            eng.synchronize()
            dt_f = (time.perf_counter() - t1) / 5
            c_f = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_f)
            extras["fast_fma_mode"] = {
                "note": "opt-in MPB_FLAG_FAST_FMA (a*v + b*w contracted into one fma): ee within 1e-9 relative of the "
                        "reference instead of bit-identical; NOT the headline", "reads_per_s_this_rank": n / dt_f,
                "ms_per_step": dt_f * 1e3, "pass": c_f.n_pass}
        # BASELINE configs[4] (ragged 50-600 bp) on the same GPU: a parity-test case, reported for reference
        nr, sr = max(n // 2, 1), 608
        r_q, r_len = eng.alloc(nr * sr), eng.alloc(nr * 4)
        eng.synth_fill(r_q, nr, sr, fixed_len=0, min_len=50, max_len=600, d_len=r_len, seed=5)
        for _ in range(2):
            eng.filter_device(r_q, nr, sr, d_len=r_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(5):
            eng.filter_device(r_q, nr, sr, d_len=r_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=params, want_counts=False)
        eng.synchronize()
        dt_r = (time.perf_counter() - t1) / 5
        extras["ragged_config5"] = {
            "note": "lengths U{50..600} in one stride-608 matrix, reads sorted by (class, length bin) on the device; "
                    "bit-exact mode; NOT the headline", "reads": nr, "reads_per_s_this_rank": nr / dt_r,
            "ms_per_step": dt_r * 1e3}
        r_q.free()
        r_len.free()

    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
        cc = torch.tensor([counts.n_pass, counts.n_fail, counts.n_overflow], dtype=torch.int64, device=coll_dev)
        dist.all_reduce(cc, op=dist.ReduceOp.SUM)      # the one optional collective: 24 bytes of totals
        n_pass, n_fail, n_ovf = (int(x) for x in cc.tolist())
    else:
        n_pass, n_fail, n_ovf = counts.n_pass, counts.n_fail, counts.n_overflow

    if rank == 0:
        total_reads = n * world * args.steps
        value = total_reads / dt
        dp_ms, dp_n = times["dp"]
        dp_avg_s = dp_ms / max(dp_n, 1) / 1e3
        alg_bytes = (L + 13) * n
        achieved = alg_bytes / dp_avg_s / 1e9 if dp_n else None
        cells = sum(cap * cnt for cap, cnt in hist.items()) * L        # DP cells one launch evaluates
        # HBM bytes of one k_dp launch from the PMC counters (collected by tools/collect_profiles.sh in
        # separate rocprofv3 passes, corrected as MI355X_MICROARCH.md prescribes); only valid for the
        # workload it was measured on
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tpath) and not args.fast_fma:
            t = json.load(open(tpath))
            w = t.get("workload", {})
            if (w.get("reads"), w.get("length"), w.get("seed")) == (n, L, args.seed):
                traffic = t["hbm_bytes_per_launch"]
        line = {
            "metric": "reads/sec filtered (300 bp synthetic)", "value": value, "unit": "reads/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "BASELINE configs[1]: %d synthetic single-end %d bp reads per GPU, "
                                   "poisson_binomial filter, alpha 0.005, uncert 0.01, resident in HBM "
                                   "(uint8 %d x %d, seed %d)" % (n, L, n, stride, args.seed),
                       "reads_per_gpu": n, "read_length": L, "row_stride": stride,
                       "parallelism": "host-side split, %d rank(s), no data-path collective" % world,
                       "mode": "fast_fma (NOT bit-exact)" if args.fast_fma else "bit-exact (no FMA)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": (achieved / HBM_PEAK_GBS) if achieved else None, "traffic": traffic,
                         "traffic_unit": "bytes per k_dp launch (PMC, profiles/pmc_traffic.json)",
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel": "k_dp", "avg_launch_ms": dp_avg_s * 1e3, "launches": dp_n,
                         "algorithmic_bytes_per_read": L + 13},
            "fp64_valu": {"cells_per_launch": cells, "ops_per_cell": 2 if args.fast_fma else 3,
                          "achieved_ops_per_s": cells * (2 if args.fast_fma else 3) / dp_avg_s if dp_n else None,
                          "peak_ops_per_s": FP64_VALU_PEAK,
                          "frac": cells * (2 if args.fast_fma else 3) / dp_avg_s / FP64_VALU_PEAK if dp_n else None,
                          "note": "the binding roof: a scalar FP64 recurrence (SURVEY §8d)"},
            "kernels_ms_per_step": {k: v[0] / max(args.steps, 1) for k, v in times.items()},
            "outcome": {"pass": n_pass, "fail": n_fail, "overflow_reruns": n_ovf},
            "row_budget_histogram": {str(k): v for k, v in hist.items() if v},
            "extras": extras,
        }
        if world == 1 and not args.no_cpu_baseline:
            line.update(cpu_baseline(args.seed, L, stride))
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    for b in (d_q, d_ee, d_ns, d_pass):
        b.free()
    eng.close()


if __name__ == "__main__":
    main()
