"""The drop-in boundary seen from the REFERENCE's side: moira.py's own parse_fastq -> process_data -> make_contig ->
write_results (a lib2to3 copy in a temp dir outside the repo, exactly as tests/golden/make_flag_matrix.py loads it), with
`nw` bound to moira_amd/dropin/nw_align.py instead of the reference's Cython extension (moira/moira.py:241-245, :794),
must write the very files the reference wrote with its own aligner (tests/golden/flag_matrix/, incl. the reference's
golden paired dataset).  Runs only where /root/reference exists (the build container); the `bernoulli` half of the
boundary needs a GPU and is tested there against the oracle and the same fixtures (tests/test_gpu_*.py)."""
import importlib.util
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference/moira"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference checkout exists in the build container only")


@pytest.fixture(scope="module")
def reference_with_dropin_aligner(oracle):
    sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
    import make_flag_matrix as FM
    import make_golden as MG
    ref = oracle.reference_module()
    if ref is None:
        pytest.skip("oracle/_ref/bernoulli.so is not built")
    M, tmp = MG.load_python_reference()
    spec = importlib.util.spec_from_file_location("nw_align_dropin", os.path.join(ROOT, "moira_amd", "dropin", "nw_align.py"))
    nw = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(nw)
    M.nw, M.Cy_nw_align = nw, True                                   # what `import nw_align as nw` binds with the drop-in first on the path
    M.bernoulli, M.Cbernoulli = FM.DefinedBernoulli(ref, M), True     # the reference's own extension (as the generator used it)
    yield M, FM, tmp
    shutil.rmtree(tmp, ignore_errors=True)


@pytest.mark.parametrize("case", ["pe_shipped_default", "pe_scores_2_-3_-1", "pe_sum_cap0_fastq", "pe_trim_overlap"])
def test_reference_pipeline_on_the_dropin_aligner_writes_the_references_files(case, reference_with_dropin_aligner):
    import golden_io as G
    M, FM, tmp = reference_with_dropin_aligner
    spec = G.flag_manifest()["cases"][case]
    if spec["input"] == "shipped":
        fwd, rev = os.path.join(REF, "test", "test1.fastq"), os.path.join(REF, "test", "test2.fastq")
    else:
        d1, d2 = G.derive_flag_inputs(G.read_fastq_records(os.path.join(G.GOLDEN, "test1.fastq.gz")),
                                      G.read_fastq_records(os.path.join(G.GOLDEN, "test2.fastq.bz2")))
        fwd, rev = os.path.join(tmp, "d1.fastq"), os.path.join(tmp, "d2.fastq")
        G.write_fastq(fwd, d1)
        G.write_fastq(rev, d2)
    files, processed, totals = FM.drive(M, FM.reference_args(**spec["flags"]), fwd, rev)
    want = G.flag_outputs()[case]
    assert processed == spec["processed"] and sorted(files) == sorted(want)
    for stem in want:
        assert files[stem] == want[stem], (case, stem)


def _serve_stub(path, name):
    import ctypes as C
    lib = C.CDLL(path)
    lib.mpb_broker_serve.argtypes = [C.c_void_p, C.c_char_p, C.c_int32, C.c_int32]
    os._exit(abs(lib.mpb_broker_serve(C.c_void_p(1), name.encode(), 8, 0)))


@pytest.mark.parametrize("case", ["se_shipped_default", "se_relabel_usearch_truncate", "se_alpha05_uncert02"])
def test_reference_pipeline_on_the_dropin_bernoulli_module(case, reference_with_dropin_aligner, oracle, tmp_path_factory, monkeypatch):
    """The other half of the boundary, as far as it goes without a GPU: moira.py's own process_data calling
    moira_amd/dropin/bernoulli.py (the real module, the real client side of libmoira_pb.so: packer, shared-memory slot,
    hand-over) -- with the broker process served by the CPU stand-in of tests/test_broker_protocol.py (mpb_broker.cpp
    compiled unchanged, the oracle for the arithmetic).  What this pins: the module's signature / marshalling / broker
    protocol inside the reference's pipeline; the kernels behind it are pinned on the GPU."""
    import ctypes as C
    import multiprocessing as mp
    import subprocess
    import time
    import golden_io as G
    M, FM, tmp = reference_with_dropin_aligner
    out = str(tmp_path_factory.mktemp("stub") / "libbroker_test.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-Wl,-Bsymbolic", os.path.join(ROOT, "moira_amd", "csrc", "mpb_broker.cpp"),
                           os.path.join(ROOT, "tests", "helpers", "broker_stub.cpp"), oracle._LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(oracle._LIB_PATH), "-o", out])
    name = "refside_%d_%s" % (os.getpid(), case[:12].replace("-", "_"))
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_serve_stub, args=(out, name))
    p.start()
    monkeypatch.setenv("MOIRA_PB_BROKER", "1")
    monkeypatch.setenv("MOIRA_PB_BROKER_NAME", name)
    try:
        from moira_amd import broker
        t0 = time.time()
        while broker.stats(name) is None or not broker.stats(name)["pid"]:
            assert p.is_alive() and time.time() - t0 < 30
            time.sleep(0.01)
        spec_ = importlib.util.spec_from_file_location("bernoulli_dropin_%s" % case[:8], os.path.join(ROOT, "moira_amd", "dropin", "bernoulli.py"))
        mod = importlib.util.module_from_spec(spec_)
        spec_.loader.exec_module(mod)
        saved = M.bernoulli
        M.bernoulli = mod                                     # what `import bernoulli` binds with the drop-in first on the path
        try:
            spec = G.flag_manifest()["cases"][case]
            if spec["input"] == "shipped":
                fwd = os.path.join(REF, "test", "test1.fastq")
            else:
                d1, _ = G.derive_flag_inputs(G.read_fastq_records(os.path.join(G.GOLDEN, "test1.fastq.gz")),
                                             G.read_fastq_records(os.path.join(G.GOLDEN, "test2.fastq.bz2")))
                fwd = os.path.join(tmp, "s1.fastq")
                G.write_fastq(fwd, d1)
            files, processed, totals = FM.drive(M, FM.reference_args(**spec["flags"]), fwd, None)
        finally:
            M.bernoulli = saved
        want = G.flag_outputs()[case]
        assert processed == spec["processed"] and sorted(files) == sorted(want)
        for stem in want:
            assert files[stem] == want[stem], (case, stem)
        assert broker.stats(name)["served"] >= processed      # every read went through the broker
    finally:
        lib = C.CDLL(out)
        lib.mpb_broker_shutdown.argtypes = [C.c_char_p]
        lib.mpb_broker_shutdown(name.encode())
        p.join(10)
        if p.is_alive():
            p.kill()


def test_dropin_survives_a_broker_that_dies(oracle, tmp_path_factory, monkeypatch):
    """The broker is killed (SIGKILL: no cleanup, its segment stays behind, still marked 'serving') while a worker is
    attached.  The worker's next call finds out within about a second (the broker's pid is gone), drops the attachment,
    attaches to the broker that now serves under the same name, and the call is answered -- once.  (On a GPU box the drop-in
    module would start that new broker itself; here, without a GPU, the test starts the CPU stand-in.)"""
    import ctypes as C
    import multiprocessing as mp
    import signal
    import subprocess
    import time
    out = str(tmp_path_factory.mktemp("stub2") / "libbroker_test.so")
    subprocess.check_call(["g++", "-O2", "-fPIC", "-shared", "-std=c++17", "-pthread", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include",
                           "-Wl,-Bsymbolic", os.path.join(ROOT, "moira_amd", "csrc", "mpb_broker.cpp"),
                           os.path.join(ROOT, "tests", "helpers", "broker_stub.cpp"), oracle._LIB_PATH,
                           "-Wl,-rpath," + os.path.dirname(oracle._LIB_PATH), "-o", out])
    from moira_amd import broker
    name = "dies_%d" % os.getpid()
    ctx = mp.get_context("spawn")
    monkeypatch.setenv("MOIRA_PB_BROKER", "1")
    monkeypatch.setenv("MOIRA_PB_BROKER_NAME", name)

    def serving():
        st = broker.stats(name)
        return st is not None and st["pid"] > 0

    def start():
        p = ctx.Process(target=_serve_stub, args=(out, name))
        p.start()
        t0 = time.time()
        while not serving():
            assert p.is_alive() and time.time() - t0 < 30
            time.sleep(0.01)
        return p
    a = start()
    spec_ = importlib.util.spec_from_file_location("bernoulli_dropin_dies", os.path.join(ROOT, "moira_amd", "dropin", "bernoulli.py"))
    mod = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(mod)
    read = ("ACGTNACGTA" * 9, [2 + (i * 5) % 39 for i in range(90)], 0.005)
    want = oracle.ee_rowwise(*read)[:2]
    assert mod.calculate_errors_PB(*read) == want
    os.kill(a.pid, signal.SIGKILL)
    a.join()
    assert not serving()                                  # the segment is still there, its broker is not
    b = start()                                           # (replaces the dead broker's segment under the same name)
    try:
        t0 = time.time()
        assert mod.calculate_errors_PB(*read) == want     # notices, re-attaches, is answered
        assert 0.5 < time.time() - t0 < 10
        assert mod.calculate_errors_PB(*read) == want and broker.stats(name)["served"] >= 2
    finally:
        lib = C.CDLL(out)
        lib.mpb_broker_shutdown.argtypes = [C.c_char_p]
        lib.mpb_broker_shutdown(name.encode())
        b.join(10)
        if b.is_alive():
            b.kill()
