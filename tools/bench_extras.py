"""The extras of bench.py's N = 1 line (moved out of bench.py in round 6): every function here measures something
that is NOT the headline -- another BASELINE config at bench size, an opt-in mode, a regime, a host-side path -- and returns a
dict that goes under `extras`.  bench.py prints the headline BEFORE any of them runs and the full line again after; each extra
catches its own exceptions (an extra never costs the headline)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench_launch import HBM_PEAK_GBS, FP64_VALU_PEAK, CONFIG2_READS, CONFIG4_SHARD  # noqa: E402,F401


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
        path = eng.last_path()
        hist = eng.class_histogram() if path["narrow_rows"] == 0 else None
        out.update({"ms_per_step": dt * 1e3, "reads_per_s": n / dt, "mean_length": 1025, "bases_per_s": n * 1025 / dt,
                    "pass": c.n_pass, "overflow_reruns": c.n_overflow, "pass_taken": {"narrow_rows": path["narrow_rows"]},
                    "reads_in_wide_kernel": int(n - sum(hist.values())) if hist is not None else None})
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
                                 "kernel": "k_narrow_rs" if stride % 64 == 0 else "k_narrow",
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
                alg = n * (fixed_len + 13) if lens is None else int((lens.astype(np.int64) + 17).sum())
                out[label] = {"reads": n, "row_stride": stride, "mean_length": L, "ms_per_step": ms, "steps": k,
                              "reads_per_s": n / ms * 1e3, "bases_per_s": n * L / ms * 1e3,
                              "pass_taken": {"narrow_rows": path["narrow_rows"], "handed_back": path["n_fallback"]},
                              "kernels_ms_per_step": kt, "row_budget_histogram": {str(a): b for a, b in hist.items() if b},
                              "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": alg,
                                           "algorithmic_bytes_per_read": "L + 13" if lens is None else "len + 17",
                                           "frac_whole_step": alg / ms / 1e6 / HBM_PEAK_GBS},
                              "outcome": {"pass": c.n_pass, "fail": c.n_fail, "overflow_reruns": c.n_overflow}}
                if hist and kt.get("dp"):
                    # what 40 % of the HBM roof would allow against what the batch's row budgets cost in exact arithmetic
                    lm = L if lens is None else None
                    caps = eng.read_budgets(n)
                    cells = int((np.asarray(caps, np.int64) * (np.full(n, fixed_len, np.int64) if lens is None else lens.astype(np.int64))).sum())
                    out[label]["fp64_valu"] = fp64_block(cells, None, 3, kt["dp"])
                    out[label]["fp64_valu"]["ms_allowed_by_40_percent_of_the_hbm_roof"] = alg / (0.40 * HBM_PEAK_GBS * 1e9) * 1e3
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


# ---- round 6: the opt-in modes with the headline's accounting, configs[4] with it, the ragged HBM-bound regime ---------------

# row-budget classes (moira_amd/csrc/mpb_internal.h MPB_CLASSES): cap = R x G rows; a tile holds 64 / G reads
CLASS_RG = [(2, 1), (3, 1), (4, 1), (5, 1), (6, 1), (7, 1), (8, 1), (9, 1), (10, 1), (12, 1), (14, 1), (16, 1),
            (10, 2), (12, 2), (14, 2), (16, 2), (10, 4), (12, 4), (14, 4), (16, 4),
            (9, 8), (10, 8), (11, 8), (12, 8), (14, 8), (16, 8), (10, 16), (12, 16), (16, 16), (12, 32), (16, 32), (16, 64)]
G_OF_CAP = {r * g: g for r, g in CLASS_RG}


def _event_pass(eng, run, reps=10):
    eng.timing(True); eng.timing_reset()
    for _ in range(reps):
        run()
    kt = {name: v[0] / reps for name, v in eng.kernel_times().items() if v[1]}
    eng.timing(False)
    return kt


def fp64_block(cells_budget, cells_alg, opc, dp_ms):
    """The headline's fp64_valu accounting for another batch / mode: `opc` FP64 operations per DP cell against the nominal
    vector FP64 rate."""
    if not dp_ms:
        return None
    s = dp_ms / 1e3
    return {"cells_per_launch": cells_budget, "ops_per_cell": opc, "peak_ops_per_s": FP64_VALU_PEAK,
            "floor_ms_per_launch": cells_budget * opc / FP64_VALU_PEAK * 1e3,
            "frac": cells_budget * opc / s / FP64_VALU_PEAK,
            "cells_algorithmic_per_launch": cells_alg,
            "frac_algorithmic": (cells_alg * opc / s / FP64_VALU_PEAK) if cells_alg else None}


def opt_in_modes(eng, d_q, n, stride, L, d_ee, d_ns, d_pass, fast_fma_headline):
    """MPB_FLAG_DECISION_ONLY and MPB_FLAG_FAST_FMA on the headline's resident batch (the FMA mode is a
    north_star-compliant mode -- ee within 1e-9 relative, decisions identical -- and gets its own kernel split and FP64 block:
    2 operations per cell)."""
    out = {}

    def rate(prm, reps=5):
        run = lambda: eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
        for _ in range(2):
            run()
        eng.synchronize()
        t1 = time.perf_counter()
        for _ in range(reps):
            run()
        eng.synchronize()
        return (time.perf_counter() - t1) / reps, run
    try:
        prm_do = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=fast_fma_headline, decision_only=True)
        dt_do, _ = rate(prm_do)
        c = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_do)
        hist_do = eng.class_histogram()
        out["decision_only_mode"] = {
            "note": "opt-in MPB_FLAG_DECISION_ONLY on the same resident batch: reads proven to fail (Chernoff bound) skip their "
                    "DP and report ee=+inf; identical pass/fail flags; NOT the headline",
            "reads_per_s_this_rank": n / dt_do, "ms_per_step": dt_do * 1e3, "pass": c.n_pass,
            "reads_run_through_dp": int(sum(hist_do.values()))}
    except Exception as e:
        out["decision_only_mode"] = {"error": repr(e)}
    if fast_fma_headline:
        return out
    try:
        prm_f = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=True)
        dt_f, run_f = rate(prm_f, reps=20)
        c_f = eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm_f)
        hist = eng.class_histogram()
        kt = _event_pass(eng, run_f)
        cells = sum(cap * cnt for cap, cnt in hist.items()) * L
        eng.filter_device(d_q, n, stride, fixed_len=L, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, want_counts=False,
                          params=eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", fast_fma=True, count_cells=True))
        alg_cells = eng.algorithmic_cells()
        out["fast_fma_mode"] = {
            "note": "opt-in MPB_FLAG_FAST_FMA (a*v + b*w contracted into one fma: 2 FP64 operations per cell): ee within 1e-9 "
                    "relative of the reference (north_star's tolerance) instead of bit-identical, decisions identical (reads within "
                    "1e-9 of a threshold are recomputed exactly); NOT the headline (the headline is the bit-exact mode)",
            "reads_per_s_this_rank": n / dt_f, "ms_per_step": dt_f * 1e3, "pass": c_f.n_pass,
            "kernels_ms_per_step": kt,
            "fp64_valu": fp64_block(cells, alg_cells, 2, kt.get("dp")),
            "roofline": {"bound": "hbm", "kernel": "k_dp<FMA>", "avg_launch_ms": kt.get("dp"),
                         "frac": (n * (L + 13) / kt["dp"] / 1e6 / HBM_PEAK_GBS) if kt.get("dp") else None,
                         "frac_whole_step": n * (L + 13) / dt_f / 1e9 / HBM_PEAK_GBS}}
    except Exception as e:
        out["fast_fma_mode"] = {"error": repr(e)}
    return out


def masked_share(caps, lens, len_shift=6):
    """The share of k_dp's lane-steps that are masked identity steps on a RAGGED batch: the device sorts reads by (class,
    floor(len / 2^len_shift)), stable, and a tile of 64 / G consecutive reads of one class runs as long as its longest read
    (moira_amd/csrc/mpb_kernels.hip: k_scatter, dp_tiles).  Reproduced on the host from the per-read row budgets
    (mpb_last_read_budgets) and lengths -> (budget cells = sum cap x len, issued cells = sum over tiles of slots x cap x longest)."""
    import numpy as np
    caps = np.asarray(caps, np.int64)
    lens = np.asarray(lens, np.int64)
    budget = int((caps * lens).sum())
    issued = 0
    per_class = {}
    for cap in np.unique(caps):
        g = G_OF_CAP.get(int(cap))
        if g is None:
            continue
        rpt = 64 // g
        idx = np.nonzero(caps == cap)[0]
        ln = lens[idx]
        order = np.argsort(ln >> len_shift, kind="stable")       # (the class is fixed: the key's second half, stable)
        ln = ln[order]
        pad = (-len(ln)) % rpt
        tiles = np.concatenate([ln, np.zeros(pad, np.int64)]).reshape(-1, rpt)
        iss = int(tiles.max(axis=1).sum()) * rpt * int(cap)
        issued += iss
        per_class[int(cap)] = 1.0 - int(cap) * int(ln.sum()) / max(iss, 1)
    return budget, issued, per_class


def ragged_config5_rate(eng, params, headline_ps_per_base, nr=5_000_000, sr=608, seed=5):
    """BASELINE configs[4] (ragged 50-600 bp) with the headline's accounting: kernel split, row-budget
    histogram, budget and algorithmic cells, FP64 fractions, the share of the roof at B = len + 17 bytes per read, and the
    share of k_dp's lane-steps that are masked identity steps (reads of one tile differ in length)."""
    import numpy as np
    out = {"note": "lengths U{50..600} in one stride-%d matrix, BASELINE's quality model, resident; the pass is the library's own "
                   "choice (the sorted pipeline: reads sorted by (class, length bin) on the device); bit-exact mode; NOT the headline" % sr,
           "reads": nr, "row_stride": sr}
    bufs = []
    try:
        bufs = [eng.alloc(nr * sr), eng.alloc(nr * 4), eng.alloc(nr * 8), eng.alloc(nr * 4), eng.alloc(nr)]
        r_q, r_len, d_ee, d_ns, d_pass = bufs
        eng.synth_fill(r_q, nr, sr, fixed_len=0, min_len=50, max_len=600, d_len=r_len, seed=seed)
        lens = r_len.download(np.int32, nr).astype(np.int64)
        run = lambda p=params, c=False: eng.filter_device(r_q, nr, sr, d_len=r_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=p, want_counts=c)
        ms, k = _wall_rate(eng, run, seconds=0.5, settle_s=0.4)
        path = eng.last_path()
        c = run(c=True)
        kt = _event_pass(eng, run, reps=8)
        alg_bytes = int((lens + 17).sum())
        bases = int(lens.sum())
        out.update({"ms_per_step": ms, "steps": k, "reads_per_s_this_rank": nr / ms * 1e3, "mean_length": bases / nr,
                    "bases_per_s": bases / ms * 1e3,
                    "pass_taken": {"narrow_rows": path["narrow_rows"], "handed_back": path["n_fallback"]},
                    "kernels_ms_per_step": kt, "outcome": {"pass": c.n_pass, "fail": c.n_fail, "overflow_reruns": c.n_overflow}})
        dp_ms = kt.get("dp")
        out["roofline"] = {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "kernel": "k_dp",
                           "algorithmic_bytes_per_launch": alg_bytes, "algorithmic_bytes_per_read": "len + 17",
                           "avg_launch_ms": dp_ms, "achieved": (alg_bytes / dp_ms / 1e6) if dp_ms else None,
                           "frac": (alg_bytes / dp_ms / 1e6 / HBM_PEAK_GBS) if dp_ms else None,
                           "frac_whole_step": alg_bytes / ms / 1e6 / HBM_PEAK_GBS}
        if path["narrow_rows"] == 0:
            prm_s = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", no_narrow=True)
            run(p=prm_s)
            hist = eng.class_histogram()
            caps = eng.read_budgets(nr)
            budget, issued, per_class = masked_share(caps, lens)
            run(p=eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", no_narrow=True, count_cells=True))
            alg_cells = eng.algorithmic_cells()
            out["row_budget_histogram"] = {str(a): b for a, b in hist.items() if b}
            out["fp64_valu"] = fp64_block(budget, alg_cells, 3, dp_ms)
            out["fp64_valu"]["cells_issued_per_launch"] = issued
            out["fp64_valu"]["frac_issued"] = (issued * 3 / (dp_ms / 1e3) / FP64_VALU_PEAK) if dp_ms else None
            out["masked_lane_steps"] = {
                "share": 1.0 - budget / max(issued, 1),
                "note": "1 - (sum cap x len) / (sum over tiles of slots x cap x the tile's longest read): identity steps of lanes "
                        "whose read is shorter than their tile's longest, and empty slots of each class's last tile; reproduced on "
                        "the host from the per-read budgets and the device's sort key (class, len >> 6)",
                "by_class_cap": {str(a): round(b, 4) for a, b in sorted(per_class.items())}}
            if dp_ms and headline_ps_per_base:
                ps = dp_ms * 1e9 / bases
                e_l2_over_l = float((lens * lens).sum()) / bases
                out["k_dp_cost"] = {"ps_per_base": ps, "headline_ps_per_base": headline_ps_per_base, "ratio": ps / headline_ps_per_base,
                                    "E_L2_over_E_L": e_l2_over_l, "rows_scale_with_length_factor": e_l2_over_l / 300.0,
                                    "ratio_after_the_length_correction": ps / headline_ps_per_base / (e_l2_over_l / 300.0),
                                    "ps_per_budget_cell": dp_ms * 1e9 / budget, "ps_per_issued_cell": dp_ms * 1e9 / issued,
                                    "note": "a read's row budget grows with its length (J ~ L), so the cells per base grow like "
                                            "E[L^2] / E[L] against a fixed 300; what is left after that correction is masking "
                                            "(masked_lane_steps) and wider classes"}
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    for b in bufs:
        try:
            b.free()
        except Exception:
            pass
    return out


def high_quality_ragged_rate(eng, n=5_000_000, stride=640, seed=6):
    """the HBM-bound regime on a RAGGED batch -- lengths U{50..600} of the clean quality profile in one stride-640
    matrix (rows are whole 128-byte lines), resident.  The library picks its pass from a sample (mpb_path_info): the narrow pass
    of ragged batches (k_rag_sort + k_narrow_rg: reads sorted by length inside windows of 4096, only the lines their bases lie in
    are fetched), timed against the sorted pipeline on the same batch.  Algorithmic bytes per read = len + 17.  NOT the headline."""
    import numpy as np
    out = {"note": "5 M reads of U{50..600} bases of the CLEAN synthetic profile (Q33..Q40), stride 640, resident; the pass is the "
                   "library's own choice; roofline: sum(len + 17) / time / 8 TB/s; NOT the headline",
           "reads": n, "row_stride": stride, "profile": 1, "seed": seed}
    bufs = []
    try:
        bufs = [eng.alloc(n * stride), eng.alloc(n * 4), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
        d_q, d_len, d_ee, d_ns, d_pass = bufs
        eng.synth_fill(d_q, n, stride, min_len=50, max_len=600, d_len=d_len, seed=seed, profile=1)
        lens = d_len.download(np.int32, n).astype(np.int64)
        alg = int((lens + 17).sum())
        prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors")
        run = lambda p=prm, c=False: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=p, want_counts=c)
        run()
        first = eng.last_path()
        ms, k = _wall_rate(eng, run)
        path = eng.last_path()
        c = run(c=True)
        kt = _event_pass(eng, run)
        nar_ms = kt.get("narrow")
        R = path["narrow_rows"]
        bases = int(lens.sum())
        out.update({"ms_per_step": ms, "steps": k, "reads_per_s": n / ms * 1e3, "mean_length": bases / n, "bases_per_s": bases / ms * 1e3,
                    "pass_taken": {"narrow_rows": R, "reads_handed_to_the_sorted_pipeline": path["n_fallback"],
                                   "sample_histogram_in_16_byte_chunks": {str(r): v for r, v in enumerate(first["sample_hist"]) if v}},
                    "kernels_ms_per_step": kt,
                    "roofline": {"bound": "hbm", "peak": HBM_PEAK_GBS, "unit": "GB/s", "algorithmic_bytes_per_launch": alg,
                                 "algorithmic_bytes_per_read": "len + 17",
                                 "whole_step": {"achieved": alg / ms / 1e6, "frac": alg / ms / 1e6 / HBM_PEAK_GBS},
                                 "kernel": "k_rag_sort + k_rag_scan + k_narrow_rg + list compaction (one span)", "avg_launch_ms": nar_ms,
                                 "achieved": (alg / nar_ms / 1e6) if nar_ms else None,
                                 "frac": (alg / nar_ms / 1e6 / HBM_PEAK_GBS) if nar_ms else None},
                    "outcome": {"pass": c.n_pass, "fail": c.n_fail, "overflow_reruns": c.n_overflow}})
        if R:
            # the binding roof of this pass is vector issue, not the stream: 3 R - 2 FP64 operations + the table address per base
            ops = bases * (3 * R - 2)
            out["fp64_valu"] = {"rows": R, "fp64_ops_per_base": 3 * R - 2, "ops_per_launch": ops,
                                "floor_ms_per_launch": ops / FP64_VALU_PEAK * 1e3,
                                "frac": (ops / (nar_ms / 1e3) / FP64_VALU_PEAK) if nar_ms else None,
                                "note": "rows 0..R-1 for every base of every read in exact three-rounding arithmetic; clean reads of "
                                        "more than ~400 bases need the third row"}
        prm_s = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", no_narrow=True)
        ms_s, k_s = _wall_rate(eng, lambda: run(p=prm_s), seconds=0.3, settle_s=0.2)
        out["sorted_pipeline_on_the_same_batch"] = {"ms_per_step": ms_s, "steps": k_s, "frac_whole_step": alg / ms_s / 1e6 / HBM_PEAK_GBS}
    except Exception as e:                                  # an extra must never cost the headline line
        out["error"] = repr(e)
    for b in bufs:
        try:
            b.free()
        except Exception:
            pass
    return out


# ---- PMC counters collected INSIDE the bench run ------------------------------------------------------------

PMC_PASSES = [("tccrd", ["TCC_EA0_RDREQ_sum", "TCC_EA0_RDREQ_32B_sum", "TCC_EA0_RDREQ_64B_sum", "TCC_EA0_RDREQ_128B_sum"]),
              ("tccwr", ["TCC_EA0_WRREQ_sum", "TCC_EA0_WRREQ_64B_sum"]),
              ("sq", ["SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_BUSY_CYCLES", "GRBM_GUI_ACTIVE"])]


def pmc_bytes(v):
    """HBM-side bytes of one launch from the L2's memory-side request counters (MI355X_MICROARCH.md, HBM: exact request sizes)."""
    rd = wr = None
    if "TCC_EA0_RDREQ_128B_sum" in v:
        rd = 32 * v.get("TCC_EA0_RDREQ_32B_sum", 0) + 64 * v.get("TCC_EA0_RDREQ_64B_sum", 0) + 128 * v["TCC_EA0_RDREQ_128B_sum"]
    if "TCC_EA0_WRREQ_sum" in v:
        wr = 64 * v.get("TCC_EA0_WRREQ_64B_sum", 0) + 32 * (v["TCC_EA0_WRREQ_sum"] - v.get("TCC_EA0_WRREQ_64B_sum", 0))
    return rd, wr


def parse_pmc_csvs(root_dir):
    """{kernel short name: {counter: mean over its dispatches, 'duration_us_<pass>': mean}} from rocprofv3's counter_collection CSVs."""
    import collections
    import csv
    import glob
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for sub, _ in PMC_PASSES:
        for f in glob.glob(os.path.join(root_dir, sub, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                try:
                    acc[k]["duration_us_" + sub].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                except (KeyError, ValueError):
                    pass
    out = {}
    for k, d in acc.items():
        # every counter row of one dispatch repeats the dispatch's times: the mean is unaffected
        out[k] = {c: sum(x) / len(x) for c, x in d.items()}
    return out


def pmc_live(n, L, seed, stride, deadline_s=75.0):
    """Start a FRESH child per counter pass -- `rocprofv3 --pmc ... -- python3 tools/pmc_probe.py` (the program itself after `--`;
    never a re-exec of this GPU-holding process) -- over a few launches of the headline's k_dp and of the clean batch's narrow
    passes, and turn the CSVs into bytes per launch.  Any failure or the deadline: {'error': ...} and the caller keeps the
    committed profiles/ values."""
    import shutil
    import tempfile
    t0 = time.time()
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return {"error": "rocprofv3 not found"}
    tmp = tempfile.mkdtemp(prefix="bench_pmc_")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["TMPDIR"] = "/tmp"
    try:
        for sub, counters in PMC_PASSES:
            left = deadline_s - (time.time() - t0)
            if left < 5:
                return {"error": "deadline of %.0f s reached before pass %s" % (deadline_s, sub)}
            cmd = [exe, "--pmc"] + counters + ["--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, sub), "--",
                                               sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py"),
                                               str(n), str(L), str(seed), str(stride)]
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=left, env=env, cwd="/tmp")
            if r.returncode != 0:
                return {"error": "pass %s: rocprofv3 exit code %d: %s" % (sub, r.returncode, (r.stderr or r.stdout)[-300:])}
        k = parse_pmc_csvs(tmp)
        res = {"seconds": round(time.time() - t0, 1), "kernels": {}}
        for name, v in k.items():
            if not (name.startswith("k_dp<false, false>") or name.startswith("k_narrow")):
                continue
            rd, wr = pmc_bytes(v)
            e = {"hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr,
                 "hbm_bytes_per_launch": (rd + wr) if rd is not None and wr is not None else None}
            if "GRBM_GUI_ACTIVE" in v and v.get("duration_us_sq"):
                cyc = v["GRBM_GUI_ACTIVE"] / 8                       # summed over the 8 XCDs
                e["valu"] = {"SQ_INSTS_VALU": v.get("SQ_INSTS_VALU"), "GRBM_GUI_ACTIVE": v["GRBM_GUI_ACTIVE"],
                             "duration_us": v["duration_us_sq"], "clock_ghz": cyc / v["duration_us_sq"] / 1e3,
                             "valu_busy": (v["SQ_INSTS_VALU"] * 4 / (cyc * 1024)) if v.get("SQ_INSTS_VALU") else None}
            res["kernels"][name] = e
        if not res["kernels"]:
            return {"error": "no k_dp / k_narrow rows in the counter CSVs"}
        return res
    except subprocess.TimeoutExpired:
        return {"error": "a counter pass did not finish inside the %.0f s deadline" % deadline_s}
    except Exception as e:                                    # noqa: BLE001 -- the committed values stand in
        return {"error": repr(e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def run_all(eng, args, d_q, n, stride, L, params, d_ee, d_ns, d_pass, headline_dp_ms, progress):
    """Every extra of the N = 1 line, one after the other; `progress(name)` is called before each (bench.py writes a stderr line,
    so that a long run never looks hung) and each extra catches its own exceptions."""
    extras = {}

    def step(name, fn):
        progress(name)
        try:
            extras[name] = fn()
        except Exception as e:                                # noqa: BLE001 -- an extra must never cost the line
            extras[name] = {"error": repr(e)}
    progress("opt-in modes")
    extras.update(opt_in_modes(eng, d_q, n, stride, L, d_ee, d_ns, d_pass, args.fast_fma))
    ps_per_base = (headline_dp_ms * 1e9 / (n * L)) if headline_dp_ms else None
    step("ragged_config5", lambda: ragged_config5_rate(eng, params, ps_per_base, nr=max(min(n, CONFIG2_READS) // 2, 1)))
    step("high_quality_ragged", lambda: high_quality_ragged_rate(eng, n=max(min(n, CONFIG2_READS) // 2, 1)))
    step("high_quality_300", lambda: high_quality_rate(eng, min(n, CONFIG2_READS), stride, L, args.seed, d_ee, d_ns, d_pass))
    step("real_profile", lambda: real_profile_rate(eng))
    step("poisson_error_calc", lambda: poisson_rate(eng, d_q, n, stride, L, d_ee, d_ns))
    step("classified_at_source", lambda: classified_rate(eng, d_q, n, stride, L, params, d_ee, d_ns, d_pass))
    step("long_reads_ragged_50_2000", lambda: long_ragged_rate(eng, params))
    step("host_fed", lambda: host_fed_rate(eng, L, stride, args.seed))
    step("config3_paired", lambda: config3_paired_rate(eng))
    step("per_read_in_process", lambda: per_read_in_process_rate(eng))
    step("per_read_broker", per_read_broker_rate)
    if n != CONFIG4_SHARD:
        step("config4_shard", lambda: config4_shard_rate(eng, L, stride, args.seed, params))
    return extras
