"""Step time of ragged resident batches through mpb_filter_device: the library's own choice, the narrow pass forced with 2 / 3 / 4
rows (k_rag_sort + k_rag_scan + k_narrow_rg) and the sorted pipeline, with the per-kernel split (HIP events) and the share of
the 8 TB/s roof at B = sum(len + 17) algorithmic bytes.

    python tools/ragged_rate.py [--reads 5000000] [--only hq|contigs|config5]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def wall(eng, run, seconds=0.4):
    run(); eng.synchronize()
    t = time.perf_counter(); run(); eng.synchronize()
    one = max(time.perf_counter() - t, 1e-5)
    for _ in range(max(3, int(0.3 / one))):
        run()
    eng.synchronize()
    k = max(5, int(seconds / one))
    t = time.perf_counter()
    for _ in range(k):
        run()
    eng.synchronize()
    return (time.perf_counter() - t) / k * 1e3


def batches(eng, n, only):
    import golden_io as G
    if only in (None, "hq"):
        yield "clean profile, U{50..600}, stride 640", dict(stride=640, synth=dict(min_len=50, max_len=600, seed=6, profile=1))
    if only in (None, "config5"):
        yield "BASELINE profile, U{50..600}, stride 640", dict(stride=640, synth=dict(min_len=50, max_len=600, seed=5, profile=0))
    if only in (None, "contigs"):
        recs = []
        for kind in ("good", "bad"):
            recs += G.read_fasta_qual(os.path.join(ROOT, "tests", "golden", "reference_test_results", "paired.qc." + kind))
        cq, cl = eng.pack([r[2] for r in recs], [r[3] for r in recs], stride=512)
        idx = np.random.default_rng(7).permutation(np.arange(n, dtype=np.int64) % len(recs)).astype(np.int32)
        yield "paired golden contigs (241-502 bp), stride 512", dict(stride=512, q=cq[idx], lens=cl[idx])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reads", type=int, default=5_000_000)
    ap.add_argument("--only", default=None)
    args = ap.parse_args()
    from moira_amd.engine import Engine
    eng = Engine(0)
    n = args.reads
    for label, b in batches(eng, n, args.only):
        stride = b["stride"]
        d_q, d_len = eng.alloc(n * stride), eng.alloc(n * 4)
        d_ee, d_ns, d_pass = eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)
        if "synth" in b:
            eng.synth_fill(d_q, n, stride, d_len=d_len, **b["synth"])
            lens = d_len.download(np.int32, n)
        else:
            d_q.upload(b["q"]); d_len.upload(b["lens"]); lens = b["lens"]
        alg = float((lens.astype(np.int64) + 17).sum())
        print("== %s: %d reads, mean length %.1f, algorithmic bytes %.3f GB" % (label, n, lens.mean(), alg / 1e9), flush=True)
        ref = None
        for name, kw in (("library's choice", {}), ("narrow R=2", dict(narrow_rows=2)), ("narrow R=3", dict(narrow_rows=3)),
                         ("narrow R=4", dict(narrow_rows=4)), ("sorted pipeline", dict(no_narrow=True))):
            prm = eng.params(alpha=0.005, uncert=0.01, ambigs="treat_as_errors", **kw)
            run = lambda: eng.filter_device(d_q, n, stride, d_len=d_len, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, params=prm, want_counts=False)
            ms = wall(eng, run)
            path = eng.last_path()
            ee = d_ee.download(np.float64, n); ps = d_pass.download(np.uint8, n)
            if ref is None:
                ref = (ee, ps)
            same = np.array_equal(ee, ref[0], equal_nan=True) and np.array_equal(ps, ref[1])
            eng.timing(True); eng.timing_reset()
            for _ in range(5):
                run()
            kt = {k: round(v[0] / 5, 4) for k, v in eng.kernel_times().items() if v[1]}
            eng.timing(False)
            print("  %-18s %.3f ms/step  %.3f of 8 TB/s  rows %d handed back %d  same results %s  kernels %s"
                  % (name, ms, alg / ms / 1e6 / 8000.0, path["narrow_rows"], path["n_fallback"], same, json.dumps(kt)), flush=True)
        for x in (d_q, d_len, d_ee, d_ns, d_pass):
            x.free()
    eng.close()


if __name__ == "__main__":
    main()
