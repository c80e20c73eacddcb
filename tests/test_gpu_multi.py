"""Single-process multi-GPU host entry (SURVEY §8e "one host thread (or process) + one HIP stream per
device"): MultiEngine / mpb_filter_host_multi with two and three contexts.  A GPU box has one card, so the contexts
share device 0 -- they are independent objects (own streams, slots, workspace), which is all the split relies on."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def same(a, b):
    return np.array_equal(a, b, equal_nan=True)


@pytest.mark.parametrize("nctx", [2, 3])
def test_multi_engine_matches_oracle(oracle, nctx):
    from moira_amd.shard import MultiEngine
    with MultiEngine([0] * nctx) as me:
        assert me.shards(10) == [(0, 5), (5, 10)] if nctx == 2 else me.shards(10) == [(0, 4), (4, 7), (7, 10)]
        # ragged, large enough that every shard goes through the chunked pipeline, not divisible by nctx
        q, lens = oracle.synth_fill(200_003, 608, min_len=50, max_len=600, seed=5)
        ee, ns, ps, _ = oracle.filter_batch(q, lens=lens, threads=oracle.lib().pbo_max_threads())
        r = me.filter(q, lens=lens)
        assert same(r.ee, ee) and np.array_equal(r.ns, ns) and np.array_equal(r.passed, ps.astype(bool))
        assert r.n_pass == int(ps.sum()) and r.n_fail == len(ps) - int(ps.sum())
        # fixed length, other parameters, caller-owned result arrays
        q2, _ = oracle.synth_fill(50_001, 320, fixed_len=300, seed=2)
        kw = dict(alpha=0.05, ambigs="ignore", maxerrors=4.0, round_=True)
        e2, n2, p2, _ = oracle.filter_batch(q2, fixed_len=300, threads=8, **kw)
        out = (np.zeros(60_000), np.zeros(60_000, np.int32), np.zeros(60_000, np.uint8))
        r2 = me.filter(q2, fixed_len=300, out=out, **kw)
        assert same(r2.ee, e2) and np.array_equal(r2.passed, p2.astype(bool)) and r2.ee.base is out[0]
        # fewer reads than contexts (empty shards), and small shards (the one-read-per-wave path inside a shard)
        for m in (1, 2, 5, 700):
            r3 = me.filter(q2[:m], fixed_len=300)
            e3, _, p3, _ = oracle.filter_batch(q2[:m], fixed_len=300)
            assert same(r3.ee, e3) and np.array_equal(r3.passed, p3.astype(bool))
        r0 = me.filter(q2[:0], fixed_len=300)
        assert len(r0.ee) == 0 and r0.n_pass == 0
        # --error_calc poisson over the same split
        import poisson_ref
        qp = q[:3000].copy()
        qp[qp == 255] = 17
        rp = me.filter_poisson(qp, lens=lens[:3000])
        from moira_amd.engine import Engine
        with Engine(0) as one:
            r1 = one.filter_poisson(qp, lens=lens[:3000])
        assert same(rp.ee, r1.ee) and np.array_equal(rp.passed, r1.passed) and np.array_equal(rp.ns, r1.ns)
        del poisson_ref


def test_multi_engine_reports_the_failing_shard(oracle):
    from moira_amd.shard import MultiEngine
    import ctypes as C
    from moira_amd import _lib as L
    with MultiEngine([0, 0, 0]) as me:
        q, lens = oracle.synth_fill(30_000, 320, min_len=50, max_len=300, seed=9)
        bad = lens.copy()
        bad[25_000] = 400                                  # does not fit the 320-byte row: shard 2's validation fails
        prm = me.params()
        ee, ns, ps = np.zeros(30_000), np.zeros(30_000, np.int32), np.zeros(30_000, np.uint8)
        rc = me.lib.mpb_filter_host_multi(me._ctxs, 3, q.ctypes.data, 30_000, 320, bad.ctypes.data, 0, C.byref(prm),
                                          ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, None, 0)
        assert rc == L.E_INVALID
        msg = me.lib.mpb_last_error().decode()
        assert "shard 2 of 3" in msg and "does not fit" in msg
        # a context listed twice is refused (calls on one context must not overlap)
        twice = (C.c_void_p * 2)(me.engines[0].ctx, me.engines[0].ctx)
        assert me.lib.mpb_filter_host_multi(twice, 2, q.ctypes.data, 100, 320, lens.ctypes.data, 0, C.byref(prm),
                                            ee.ctypes.data, ns.ctypes.data, ps.ctypes.data, None, 0) == L.E_INVALID
        # and the engines are still usable
        r = me.filter(q, lens=lens)
        e, _, p, _ = oracle.filter_batch(q, lens=lens, threads=8)
        assert same(r.ee, e)


def test_cli_device_list_reproduces_the_golden_files(tmp_path, golden_dir):
    """`--device 0,0`: every chunk of the run is split over two contexts; the reference's golden files still come out
    byte for byte (moira/test/test_moira.py:73-113)."""
    import gzip
    import os
    import shutil
    from moira_amd import cli
    from test_cli_golden import reference_args
    fq = tmp_path / "test1.fastq"
    with gzip.open(os.path.join(golden_dir, "test1.fastq.gz"), "rb") as f, open(fq, "wb") as g:
        shutil.copyfileobj(f, g)
    out = str(tmp_path / "forward")
    a = reference_args(paired=False, forward_fastq=str(fq), output_prefix=out, silent=True, device="0,0")
    assert cli.main(a, out=open(os.devnull, "w")) == 0
    base = os.path.join(golden_dir, "reference_test_results", "forward.qc.")
    for ext in ("good.fasta", "good.qual", "good.names", "bad.fasta", "bad.qual", "bad.names"):
        assert open(out + ".qc." + ext, "rb").read() == open(base + ext, "rb").read(), ext
    assert cli.parse_devices("all") == ["all"] and cli.parse_devices("1, 3") == [1, 3] and cli.parse_devices(2) == [2]
    with pytest.raises(ValueError):
        cli.parse_devices("gpu0")


def test_shard_threads_are_placed_next_to_their_gpu_and_the_caller_is_left_alone(oracle):
    """each shard thread of mpb_filter_host_multi restricts itself to the CPUs of its GPU's NUMA node for the
    duration of its pipeline.  The CALLING thread runs shard 0: its own affinity mask must be what it was when the call
    returns; the lookup the pinning uses must answer for this box's GPU; MOIRA_PB_NO_NUMA turns it off; results unchanged."""
    import ctypes as C
    import os
    from moira_amd import _lib as L
    from moira_amd.shard import MultiEngine
    lib = L.load()
    hip = C.CDLL("libamdhip64.so")
    bus = C.create_string_buffer(64)
    assert hip.hipDeviceGetPCIBusId(bus, 64, 0) == 0
    node, cpus = C.c_int32(-9), C.create_string_buffer(4096)
    assert lib.mpb_numa_cpulist_for_pci(b"", bus.value, C.byref(node), cpus, 4096) == 0
    assert node.value >= -1 and (node.value == -1 or len(cpus.value) > 0)
    print("GPU 0 at %s: NUMA node %d, CPUs %s" % (bus.value.decode(), node.value, cpus.value.decode() or "(no information)"))
    before = os.sched_getaffinity(0)
    q, _ = oracle.synth_fill(300_001, 320, fixed_len=300, seed=2)
    ee, ns, ps, _ = oracle.filter_batch(q, fixed_len=300, threads=8)
    for no_numa in (False, True):
        if no_numa:
            os.environ["MOIRA_PB_NO_NUMA"] = "1"
        try:
            with MultiEngine([0, 0]) as me:
                r = me.filter(q, fixed_len=300)
        finally:
            os.environ.pop("MOIRA_PB_NO_NUMA", None)
        assert same(r.ee, ee) and np.array_equal(r.passed, ps.astype(bool))
        assert os.sched_getaffinity(0) == before
