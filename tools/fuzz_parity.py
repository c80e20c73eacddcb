#!/usr/bin/env python3
"""Randomised differential test: HIP path (through the C ABI) vs the CPU oracle over random batch
shapes, length distributions, quality profiles, alphas and modes.  Bit-exact or it prints the case.
    python tools/fuzz_parity.py [rounds] [seed]
"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np  # noqa: E402
import pb_oracle as O  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402
from poisson_ref import calculate_errors_poisson  # noqa: E402

rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 100
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
rng = np.random.default_rng(seed)
threads = O.lib().pbo_max_threads()
bad = 0
t0 = time.time()
with Engine(0) as eng:
    for it in range(rounds):
        n = int(rng.choice([1, 7, 63, 64, 65, 255, 257, 1000, 5000, 20000]))
        stride = int(rng.choice([16, 32, 48, 64, 160, 192, 256, 304, 320, 448, 512, 608, 1024]))
        if rng.random() < 0.15:                                      # round 3: long rows (tile classes on long rows, k_wide)
            stride = int(rng.choice([1040, 1536, 2048, 3072, 4096, 8192, 16384, 16384, 32768, 65536]))   # round 4: up to 65535 bases
            n = min(n, max(1, int(3e9 / (stride * stride))))         # bounds the oracle's work (J * L cells per read, J ~ L)
        max_l = min(stride, 65535)
        fixed = rng.random() < 0.4
        if fixed:
            L = int(rng.integers(0, max_l + 1))
            lens = np.full(n, L, np.int32)
        else:
            lens = rng.integers(0, max_l + 1, n).astype(np.int32)
        kind = rng.integers(0, 6)
        if kind == 0:
            q = rng.integers(1, 42, (n, stride))
        elif kind == 1:
            q = rng.integers(1, 8, (n, stride))                      # very low quality: wide classes
        elif kind == 2:
            q = rng.integers(30, 42, (n, stride))                    # very high quality: tiny J, UB-case reads
        elif kind == 3:
            q = rng.integers(1, 255, (n, stride))
        elif kind == 4:
            base = rng.integers(2, 41, (n, 1))
            q = np.clip(base + rng.integers(-3, 4, (n, stride)), 1, 60)
        else:
            q = np.clip(40 - (np.arange(stride)[None, :] / max(stride, 1)) * rng.integers(0, 40, (n, 1)), 1, 41)
        q = q.astype(np.uint8)
        amb = rng.random((n, stride)) < rng.choice([0, 0.001, 0.02, 0.3])
        q[amb] = np.where(rng.random(int(amb.sum())) < 0.8, 0, 255)
        if rng.random() < 0.5:                                       # garbage in the padding
            pad = np.arange(stride)[None, :] >= lens[:, None]
            q[pad] = rng.integers(0, 256, int(pad.sum()), dtype=np.uint8)
        kw = dict(alpha=float(rng.choice([0.005, 0.001, 0.05, 0.3, 1e-5, 0.9])),
                  ambigs=str(rng.choice(["treat_as_errors", "ignore", "disallow"])),
                  round_=bool(rng.random() < 0.2))
        if rng.random() < 0.3:
            kw["maxerrors"] = float(rng.choice([0.5, 1.0, 3.0, 10.0]))
        else:
            kw["uncert"] = float(rng.choice([0.01, 0.02, 0.1, 1.0]))
        extra = {}
        if rng.random() < 0.15:
            extra["test_underpredict"] = True
        if rng.random() < 0.15:
            extra["decision_only"] = True
        elif rng.random() < 0.2:
            extra["fast_fma"] = True                                 # ee within 1e-9 relative, decisions exact
        if rng.random() < 0.1:                                       # --error_calc poisson against the reference's formula
            m = min(n, 300)
            qp = q[:m].copy()
            qp[qp == 255] = 17                                       # the Python reference scores 'n' as a normal base
            r = eng.filter_poisson(qp, lens=lens[:m], alpha=kw["alpha"], ambigs=kw["ambigs"], round_=kw["round_"],
                                   **({"maxerrors": kw["maxerrors"]} if "maxerrors" in kw else {"uncert": kw.get("uncert", 0.01)}))
            okp = True
            for i in range(m):
                row = qp[i, :lens[i]]
                try:
                    e, k = calculate_errors_poisson("".join("N" if v == 0 else "A" for v in row), [20 if v == 0 else int(v) for v in row], kw["alpha"])
                except OverflowError:
                    okp = okp and np.isnan(r.ee[i])
                    continue
                if kw["ambigs"] == "treat_as_errors":
                    e = e + k
                if kw["round_"]:
                    e = float(np.floor(e))
                okp = okp and r.ee[i] == e and r.ns[i] == k
            if not okp:
                bad += 1
                print("POISSON MISMATCH round %d: n=%d stride=%d kw=%s" % (it, m, stride, kw), flush=True)
        eng.batched_only = bool(rng.random() < 0.5)                  # small batches: pipeline or one-read-per-wave path
        ee, ns, ps, rows = O.filter_batch(q, lens=lens, threads=threads, **kw)
        too = rows > 16384                                           # more rows than 16 waves hold (only beyond 16383 bases):
        ee[too], ps[too] = np.nan, 0                                 # no result, never a wrong one
        r = eng.filter(q, lens=None if fixed else lens, fixed_len=int(lens[0]) if fixed else None, **kw, **extra)
        ok = np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
        if extra.get("decision_only"):
            sk = np.isinf(r.ee) & ~np.isinf(ee)
            ok = ok and np.array_equal(r.ee[~sk], ee[~sk], equal_nan=True) and not ps[sk].any()
        elif extra.get("fast_fma"):
            both = ~np.isnan(r.ee) & ~np.isnan(ee)
            rel = np.abs(r.ee[both] - ee[both]) / np.maximum(np.abs(ee[both]), 1e-300)
            ok = ok and np.array_equal(np.isnan(r.ee), np.isnan(ee)) and (rel.size == 0 or rel.max() <= 1e-9)
        else:
            ok = ok and np.array_equal(r.ee, ee, equal_nan=True)
        if rng.random() < 0.3 and int(q[(q != 255)].max(initial=0)) <= 222 and stride <= 16384:      # (the fused pass: rows of <= 16384 bytes)
            # classified at source (round 3): the same reads as FASTQ text resident in HBM, decoded and classified by one
            # pass (k_classify_linear), the filter starting at the scan -- same oracle results, and the packed matrix back
            seq = np.where(q == 0, ord("N"), np.where(q == 255, ord("n"), ord("A"))).astype(np.uint8)
            amb_q = rng.integers(33, 75, q.shape, dtype=np.uint8)
            qual = np.where((q == 0) | (q == 255), amb_q, q + 33).astype(np.uint8)
            pad = np.arange(stride)[None, :] >= lens[:, None]
            seq[pad] = rng.integers(0, 256, int(pad.sum()), dtype=np.uint8)          # text past the end is never looked at
            qual[pad] = rng.integers(0, 256, int(pad.sum()), dtype=np.uint8)
            bufs = [eng.alloc(n * stride).upload(seq), eng.alloc(n * stride).upload(qual), eng.alloc(n * stride),
                    eng.alloc(n * 4).upload(lens), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n), eng.alloc(4).upload(np.zeros(1, np.int32))]
            d_seq, d_qual, d_out, d_len, d_ee, d_ns, d_pass, d_err = bufs
            c = eng.filter_ascii_device(d_seq, d_qual, n, stride, d_out, d_len=None if fixed else d_len,
                                        fixed_len=int(lens[0]) if fixed else 0, d_ee=d_ee, d_ns=d_ns, d_pass=d_pass, d_err=d_err,
                                        params=eng.params(**kw))
            wantq = np.where(pad, 0, q).astype(np.uint8)
            okc = (np.array_equal(d_out.download(np.uint8, n * stride).reshape(n, stride), wantq)
                   and np.array_equal(d_ee.download(np.float64, n), ee, equal_nan=True)
                   and np.array_equal(d_ns.download(np.int32, n), ns) and np.array_equal(d_pass.download(np.uint8, n), ps)
                   and c.n_pass == int(ps.sum()) and int(d_err.download(np.int32, 1)[0]) == 0)
            for b in bufs:
                b.free()
            if not okc:
                bad += 1
                print("CLASSIFIED-AT-SOURCE MISMATCH round %d: n=%d stride=%d fixed=%s kind=%d kw=%s" % (it, n, stride, fixed, kind, kw), flush=True)
        if fixed and int(lens[0]) >= 1 and not too.any():
            # round 5: the same (fixed-length) batch resident in HBM through the natural-order narrow pass, forced with 2 / 3 / 4 rows
            # (k_narrow finishes what crosses within them, counts the 'N's, hands the rest -- and every read with an 'n' -- to the
            # sorted pipeline); ragged batches and the opt-in flags never take that pass
            rows0 = int(rng.integers(2, 5))
            bufs = [eng.alloc(n * stride).upload(q), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n)]
            c = eng.filter_device(bufs[0], n, stride, fixed_len=int(lens[0]), d_ee=bufs[1], d_ns=bufs[2], d_pass=bufs[3],
                                  params=eng.params(narrow_rows=rows0, **kw))
            path = eng.last_path()
            amb_n = (q[:, :int(lens[0])] == 255).any(1)
            okn = (path["narrow_rows"] == rows0 and path["n_fallback"] == int((amb_n | (rows > rows0)).sum())
                   and np.array_equal(bufs[1].download(np.float64, n), ee, equal_nan=True)
                   and np.array_equal(bufs[2].download(np.int32, n), ns) and np.array_equal(bufs[3].download(np.uint8, n), ps)
                   and c.n_pass == int(ps.sum()))
            for b in bufs:
                b.free()
            if not okn:
                bad += 1
                print("NARROW-PASS MISMATCH round %d: n=%d stride=%d L=%d rows0=%d kind=%d kw=%s path=%s"
                      % (it, n, stride, int(lens[0]), rows0, kind, kw, path), flush=True)
        if not fixed and stride <= 4096 and not too.any():
            # round 6: the same RAGGED batch resident in HBM through the narrow pass of ragged batches, forced with 2 / 3 / 4 rows
            # (k_rag_sort + k_narrow_rg: length-sorted windows, per-read lengths; the rest goes to the sorted pipeline with its lengths)
            rows0 = int(rng.integers(2, 5))
            split = int(rng.choice([0, 0, 3, 10, 20, 40, 255])) if rows0 >= 3 else 0       # mixed rows: short groups with a row less
            bufs = [eng.alloc(n * stride).upload(q), eng.alloc(n * 8), eng.alloc(n * 4), eng.alloc(n), eng.alloc(n * 4).upload(lens)]
            c = eng.filter_device(bufs[0], n, stride, d_len=bufs[4], d_ee=bufs[1], d_ns=bufs[2], d_pass=bufs[3],
                                  params=eng.params(narrow_rows=rows0, narrow_split=split, **kw))
            path = eng.last_path()
            amb_n = ((q == 255) & (np.arange(stride)[None, :] < lens[:, None])).any(1)
            lo_back, hi_back = int((amb_n | (rows > rows0)).sum()), int((amb_n | (rows > rows0 - (1 if split else 0))).sum())
            okr = (path["narrow_rows"] == rows0 and lo_back <= path["n_fallback"] <= hi_back
                   and np.array_equal(bufs[1].download(np.float64, n), ee, equal_nan=True)
                   and np.array_equal(bufs[2].download(np.int32, n), ns) and np.array_equal(bufs[3].download(np.uint8, n), ps)
                   and c.n_pass == int(ps.sum()))
            for b in bufs:
                b.free()
            if not okr:
                bad += 1
                print("RAGGED-NARROW MISMATCH round %d: n=%d stride=%d rows0=%d split=%d kind=%d kw=%s path=%s"
                      % (it, n, stride, rows0, split, kind, kw, path), flush=True)
        if (it + 1) % 100 == 0:
            print("fuzz: %d rounds done, %d mismatching, %.0f s" % (it + 1, bad, time.time() - t0), flush=True)
        if not ok:
            bad += 1
            d = np.nonzero(~((r.ee == ee) | (np.isnan(r.ee) & np.isnan(ee))))[0]
            print("MISMATCH round %d: n=%d stride=%d fixed=%s kind=%d kw=%s extra=%s batched_only=%s first diffs %s rows %s"
                  % (it, n, stride, fixed, kind, kw, extra, eng.batched_only, d[:5], rows[d[:5]]), flush=True)
print("fuzz: %d rounds, %d mismatching rounds, %.1f s (oracle threads %d)" % (rounds, bad, time.time() - t0, threads))
sys.exit(1 if bad else 0)
