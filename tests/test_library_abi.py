"""CPU-side checks of the C-ABI library: it loads, exports every symbol the header declares,
validates arguments, packs reads like the oracle, and refuses to run without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from moira_amd import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_symbols():
    src = open(os.path.join(ROOT, "include", "moira_pb.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mpb_[a-z_A-Z0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    lib = L.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), s
    assert sorted(L.PROTOTYPES) == syms          # the ctypes table covers the whole header
    assert b"gfx950" in lib.mpb_version()
    # ... and nothing else: the library is built with -fvisibility=hidden and a linker version script, so no mangled
    # launch wrapper, no compiler-generated id and no weak std:: template instance leaks out (hygiene)
    import shutil
    import subprocess
    if shutil.which("nm"):
        out = subprocess.run(["nm", "-D", "--defined-only", L.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = sorted(l.split()[-1] for l in out.splitlines() if l.strip())
        assert exported == syms, sorted(set(exported) ^ set(syms))


def test_no_cpu_fallback_without_device():
    lib = L.load()
    if lib.mpb_device_count() > 0:
        pytest.skip("a GPU is visible")
    from moira_amd.engine import Engine
    with pytest.raises(L.NoDeviceError):
        Engine(0)


def test_product_does_not_reference_the_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "moira_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pb_oracle" not in text and "oracle/" not in text.replace("the oracle", ""), f


def test_pack_matches_oracle(oracle):
    lib = L.load()
    rng = np.random.default_rng(3)
    for _ in range(50):
        n = int(rng.integers(0, 70))
        quals = rng.integers(0, 60, n).astype(np.int32)
        seq = "".join(rng.choice(list("ACGTNn"), n)) if n else ""
        row = np.full(80, 7, np.uint8)
        assert lib.mpb_pack_read(seq.encode(), quals.ctypes.data, n, row.ctypes.data, 80) == 0
        assert np.array_equal(row, oracle.pack_read(seq, quals, 80))
        asc = bytes(int(q) + 33 for q in quals)
        row2 = np.full(80, 9, np.uint8)
        assert lib.mpb_pack_read_ascii(seq.encode(), asc, n, 33, row2.ctypes.data, 80) == 0
        assert np.array_equal(row, row2)


def test_pack_rejects_bad_scores():
    lib = L.load()
    row = np.zeros(16, np.uint8)
    q = np.array([5, -1, 3], np.int32)
    assert lib.mpb_pack_read(b"ACG", q.ctypes.data, 3, row.ctypes.data, 16) == L.E_RANGE
    assert b"positive" in lib.mpb_last_error()
    q = np.array([5, 255, 3], np.int32)
    assert lib.mpb_pack_read(b"ACG", q.ctypes.data, 3, row.ctypes.data, 16) == L.E_RANGE
    q = np.arange(20, dtype=np.int32)
    assert lib.mpb_pack_read(None, q.ctypes.data, 20, row.ctypes.data, 16) == L.E_INVALID


def test_no_fma_in_exact_dp_kernel():
    """The bit-exact DP must not contain fused multiply-adds except inside the IEEE division
    expansion (3 v_fma_f64 + 2 v_fmac_f64 per division, one division per class body)."""
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    import subprocess
    import tempfile
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call([hipcc, "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                               "-fno-fast-math", "-x", "hip", "--cuda-device-only", "-S", "-o", out,
                               os.path.join(ROOT, "moira_amd", "csrc", "mpb_kernels.hip")],
                              stderr=subprocess.DEVNULL)
        text = open(out).read()
    # split into functions; exact-mode class bodies are dp_tile<R,G,false> = ...Lb0EEE in the mangled name
    bodies = re.split(r"\n(?=_ZN\S+:)", text)
    exact = [b for b in bodies if re.match(r"_ZN\S*dp_tilesILi\d+ELi\d+ELb0EE", b)]
    fast = [b for b in bodies if re.match(r"_ZN\S*dp_tilesILi\d+ELi\d+ELb1EE", b)]
    hdr = open(os.path.join(ROOT, "moira_amd", "csrc", "mpb_internal.h")).read()
    ncls = int(re.search(r"#define MPB_NCLS (\d+)", hdr).group(1))
    shapes = lambda macro: {(int(r), int(g)) for _, r, g in re.findall(r"X\((\d+), (\d+), (\d+)\)", re.search(
        r"#define %s\(X\)((?:.*\\\n)*.*)" % macro, hdr).group(1))}
    tile, thin = shapes("MPB_CLASSES"), shapes("MPB_THIN_CLASSES")       # thin: the latency bodies of k_small (round 4)
    assert len(tile) == ncls and len(thin) == 10 and len(thin - tile) == 9
    nbodies = len(tile | thin)
    assert len(exact) == nbodies and len(fast) == nbodies
    for b in exact:
        assert b.count("v_fma_f64") == 3 and b.count("v_fmac_f64") == 2, b.split(":")[0]
        assert b.count("v_div_fixup_f64") == 1
        # the only scratch traffic allowed is the callee-saved VGPR save/restore at entry/exit
        sc = [l for l in b.splitlines() if "scratch_" in l]
        assert all("Folded Spill" in l or "Folded Reload" in l for l in sc)
    assert all(b.count("v_fma_f64") + b.count("v_fmac_f64") > 5 for b in fast)
    # the natural-order narrow pass (k_narrow: LDS-DMA ring; k_narrow_rs: whole lines staged in registers), 2..4 rows, and
    # the resident one-read server: fused operations only inside their epilogues' IEEE divisions
    narrow = [b for b in bodies if re.match(r"_ZN\S*(k_narrowILi\d|k_narrow_rsILi\d)", b)]
    assert len(narrow) == 7, [b.split(":")[0] for b in narrow]      # k_narrow<2..4>, k_narrow_rs<2..4, aligned>, k_narrow_rs<2, general>
    for b in narrow:
        ndiv = b.count("v_div_fixup_f64")
        assert ndiv >= 1 and b.count("v_fma_f64") == 3 * ndiv and b.count("v_fmac_f64") == 2 * ndiv, b.split(":")[0]
        assert "scratch_" not in b, b.split(":")[0]                         # no spills
    # round 6: the narrow pass of ragged batches, pure (R rows everywhere) and with mixed rows (short groups with a row less): the same rule;
    # the forced four waves per SIMD cost R >= 3 a few spilled per-group values (scratch), never a fused cell
    ragged = [b for b in bodies if re.match(r"_ZN\S*k_narrow_rgILi\d", b)]
    assert sorted(re.search(r"k_narrow_rgILi(\d)ELi(\d)E", b).groups() for b in ragged) == [("2", "2"), ("3", "2"), ("3", "3"), ("4", "3"), ("4", "4")]
    for b in ragged:
        ndiv = b.count("v_div_fixup_f64")
        assert ndiv >= 1 and b.count("v_fma_f64") == 3 * ndiv and b.count("v_fmac_f64") == 2 * ndiv, b.split(":")[0]
    for name in ("k_narrow_rgILi2ELi2E", "k_narrow_rgILi3ELi2E", "k_narrow_rgILi3ELi3E", "k_narrow_rgILi4ELi3E", "k_narrow_rgILi4ELi4E"):
        m = re.search(r"\.amdhsa_kernel _ZN\S*%s\S*\n(?:.*\n)*?\s*\.amdhsa_next_free_vgpr (\d+)" % name, text)
        assert m and int(m.group(1)) <= 128, (name, m and m.group(1))
    # k_dp / k_small / k_serve call the class bodies: all three keep the 128-register budget (4 waves per SIMD) -- a caller
    # declared with looser launch bounds would let the shared bodies grow and halve k_dp's occupancy
    for name in ("k_dpILb0ELb0E", "k_smallILb0E", "k_serve", "k_narrow_rsILi2ELb1E", "k_narrow_rsILi3ELb1E", "k_narrow_rsILi4ELb1E", "k_narrow_rsILi2ELb0E"):
        m = re.search(r"\.amdhsa_kernel _ZN\S*%s\S*\n(?:.*\n)*?\s*\.amdhsa_next_free_vgpr (\d+)" % name, text)
        assert m and int(m.group(1)) <= 128, (name, m and m.group(1))


def test_pack_batch_ascii_matches_per_read_packer(oracle):
    lib = L.load()
    rng = np.random.default_rng(8)
    seqs, quals = [], []
    for _ in range(200):
        n = int(rng.integers(0, 90))
        seqs.append("".join(rng.choice(list("ACGTNn"), n)) if n else "")
        quals.append("".join(chr(int(v) + 33) for v in rng.integers(0, 42, n)))
    off = np.zeros(len(seqs) + 1, np.int64)
    off[1:] = np.cumsum([len(s) for s in seqs])
    for max_len in (0, 50):
        out = np.full((len(seqs), 96), 7, np.uint8)
        lens = np.zeros(len(seqs), np.int32)
        rc = lib.mpb_pack_batch_ascii("".join(seqs).encode(), "".join(quals).encode(), off.ctypes.data, len(seqs),
                                      33, max_len, 96, out.ctypes.data, lens.ctypes.data)
        assert rc == 0
        for i, (s, ql) in enumerate(zip(seqs, quals)):
            if max_len:
                s, ql = s[:max_len], ql[:max_len]
            assert lens[i] == len(s)
            assert np.array_equal(out[i], oracle.pack_read(s, [ord(c) - 33 for c in ql], 96))
    bad = "".join(quals)[:-1] + chr(20) if off[-1] else ""
    if bad:
        out = np.zeros((len(seqs), 96), np.uint8)
        assert lib.mpb_pack_batch_ascii("".join(seqs).encode(), bad.encode(), off.ctypes.data, len(seqs), 33, 0, 96,
                                        out.ctypes.data, None) == L.E_RANGE


def test_contig_library_exports_every_declared_symbol():
    import ctypes
    from moira_amd import contig as CT
    lib = CT.load()
    src = open(os.path.join(ROOT, "include", "moira_contig.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    syms = sorted(set(re.findall(r"\b(mct_[a-z_A-Z0-9]+)\s*\(", src)))
    assert set(syms) >= {"mct_contigs_batch", "mct_last_error", "mct_make_contig", "mct_nw_align", "mct_nw_align_scalar",
                         "mct_reverse_complement"}
    for s in syms:
        assert isinstance(getattr(lib, s), ctypes._CFuncPtr)


def test_word_wide_fastq_decode_equals_the_byte_rules(tmp_path):
    """The on-device FASTQ decode works on four bases per instruction (decode4 in mpb_kernels.hip); its byte-by-byte
    twin (decode4_bytes) states the reference's rules (moira/moira.py:1177, bernoullimodule.c:104-107,196).  Both are
    plain integer code, so they are cut out of the kernel source and compared on the host: 8 M random dwords."""
    import shutil
    import subprocess
    if not shutil.which("g++"):
        pytest.skip("no g++")
    src = open(os.path.join(ROOT, "moira_amd", "csrc", "mpb_kernels.hip")).read()
    a = src.index("__device__ __forceinline__ uint32_t decode4_bytes(")
    b = src.index("// 16 bases; `pos0` = position of the first base in the read")
    (tmp_path / "swar_funcs.h").write_text(src[a:b])
    exe = str(tmp_path / "check")
    subprocess.check_call(["g++", "-O2", "-I", str(tmp_path), os.path.join(ROOT, "tests", "helpers", "swar_decode_check.cpp"), "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True)
    assert out.returncode == 0 and "0 mismatches" in out.stdout, out.stdout[-2000:]


def test_pack_batch_coded_assigns_spare_codes(oracle):
    """Host-only half of the coded batch entry (round 4): scores above 254 get byte codes the batch does not use, from
    254 down, in order of first appearance; everything else packs as mpb_pack_read does."""
    lib = L.load()
    seqs = ["ACNTn", "GGG", "", "ACGTACGT"]
    quals = [[30, 300, 30, 0, 7], [5000, 300, 254], [], [253, 2 ** 31 - 1, 1, 2, 3, 4, 5, 6]]
    off = np.zeros(5, np.int64)
    off[1:] = np.cumsum([len(x) for x in quals])
    flat = np.array(sum(quals, []), np.int32)
    q = np.full((4, 16), 9, np.uint8)
    lens = np.empty(4, np.int32)
    codes = np.empty(256, np.int32)
    assert lib.mpb_pack_batch_coded("".join(seqs).encode(), flat.ctypes.data, off.ctypes.data, 4, 0, 16, q.ctypes.data,
                                    lens.ctypes.data, codes.ctypes.data) == 0
    assert lens.tolist() == [5, 3, 0, 8]
    changed = {i: int(c) for i, c in enumerate(codes) if c != i}
    assert changed == {252: 300, 251: 5000, 250: 2 ** 31 - 1}          # 254 and 253 are in use by the batch itself
    assert q[0, :5].tolist() == [30, 252, 0, 1, 255] and not q[0, 5:].any()
    assert q[1, :3].tolist() == [251, 252, 254] and q[3, :2].tolist() == [253, 250] and not q[2].any()
    # without big scores it is the plain packer
    plain = [[1, 2, 3], [40, 0, 41, 93]]
    off2 = np.array([0, 3, 7], np.int64)
    flat2 = np.array(sum(plain, []), np.int32)
    q2 = np.empty((2, 16), np.uint8)
    assert lib.mpb_pack_batch_coded(b"ACGNnAC", flat2.ctypes.data, off2.ctypes.data, 2, 0, 16, q2.ctypes.data,
                                    lens.ctypes.data, codes.ctypes.data) == 0
    assert codes.tolist() == list(range(256))
    assert np.array_equal(q2[0], oracle.pack_read("ACG", plain[0], 16)) and np.array_equal(q2[1], oracle.pack_read("NnAC", plain[1], 16))
    # max_len truncates (moira.py:806-807); a negative score and a batch without a free code are refused
    assert lib.mpb_pack_batch_coded(None, flat.ctypes.data, off.ctypes.data, 4, 2, 16, q.ctypes.data, lens.ctypes.data, codes.ctypes.data) == 0
    assert lens.tolist() == [2, 2, 0, 2]
    neg = np.array([3, -1], np.int32)
    assert lib.mpb_pack_batch_coded(None, neg.ctypes.data, np.array([0, 2], np.int64).ctypes.data, 1, 0, 16, q.ctypes.data,
                                    lens.ctypes.data, codes.ctypes.data) == L.E_RANGE
    full = np.array(list(range(1, 255)) + [300], np.int32)
    qq = np.empty((1, 256), np.uint8)
    assert lib.mpb_pack_batch_coded(None, full.ctypes.data, np.array([0, 255], np.int64).ctypes.data, 1, 0, 256, qq.ctypes.data,
                                    lens.ctypes.data, codes.ctypes.data) == L.E_RANGE
    assert b"distinct scores above 254" in lib.mpb_last_error()


def test_python2_form_of_the_dropin_parses_as_python2_and_checks_its_arguments():
    """moira.py is Python 2; this image has none.  moira_amd/dropin/py2/bernoulli.py is written for both versions: it must
    parse under lib2to3's Python-2 grammar (the one with the print statement) and under Python 3, use nothing but ctypes
    and the standard library, and raise the extension's exceptions (moira/bernoullimodule.c:74-90) before it needs a GPU."""
    import ast
    import importlib.util
    import lib2to3.pgen2.driver as D
    import lib2to3.pygram as G
    import lib2to3.pytree as T
    path = os.path.join(ROOT, "moira_amd", "dropin", "py2", "bernoulli.py")
    src = open(path).read()
    ast.parse(src)
    D.Driver(G.python_grammar, convert=T.convert).parse_string(src + "\n")
    imported = {n.names[0].name.split(".")[0] for n in ast.walk(ast.parse(src)) if isinstance(n, ast.Import)}
    assert imported <= {"array", "ctypes", "os", "subprocess", "sys", "time", "fcntl", "multiprocessing", "stat"}      # stdlib only
    tree = ast.parse(src)
    py3_only = (ast.JoinedStr, ast.AnnAssign, ast.Nonlocal, ast.NamedExpr, ast.AsyncFunctionDef, ast.Await, ast.YieldFrom,
                ast.MatMult)
    assert not [n for n in ast.walk(tree) if isinstance(n, py3_only)]                       # f-strings, annotations, ...
    for fn in (n for n in ast.walk(tree) if isinstance(n, (ast.FunctionDef, ast.Lambda))):
        a = fn.args
        assert not a.kwonlyargs and not a.posonlyargs and not any(x.annotation for x in a.args)
        assert isinstance(fn, ast.Lambda) or fn.returns is None
    spec = importlib.util.spec_from_file_location("bernoulli_py2form", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    with pytest.raises(TypeError):
        mod.calculate_errors_PB(5, [1], 0.005)
    with pytest.raises(TypeError):
        mod.calculate_errors_PB("A", (1,), 0.005)
    with pytest.raises(TypeError):
        mod.calculate_errors_PB("AC", [30, 1.5], 0.005)
    with pytest.raises(ValueError, match="Alpha must be between 0 and 1"):
        mod.calculate_errors_PB("A", [30], 1.0)
    with pytest.raises(ValueError, match="same length"):
        mod.calculate_errors_PB("AC", [30], 0.005)
    assert mod.calculate_errors is mod.calculate_errors_PB


def test_numa_lookup_parses_the_sysfs_files(tmp_path):
    """a shard thread of mpb_filter_host_multi pins itself to the CPUs of its GPU's NUMA node.  The lookup --
    PCI bus id (upper-case hex, as HIP prints it) -> numa_node -> cpulist -- against a copy of the two sysfs files."""
    lib = L.load()
    root = tmp_path
    for bus, node in (("0000:c1:00.0", "1\n"), ("0000:05:00.0", "0\n"), ("0000:85:00.0", "-1\n")):
        d = root / "sys/bus/pci/devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node)
    for node, cpus in ((0, "0-47,96-143\n"), (1, "48-95,144-191\n")):
        d = root / "sys/devices/system/node" / ("node%d" % node)
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)

    def look(bus, r=str(root)):
        node, buf = C.c_int32(-7), C.create_string_buffer(256)
        rc = lib.mpb_numa_cpulist_for_pci(r.encode(), bus.encode(), C.byref(node), buf, 256)
        return rc, node.value, buf.value.decode()

    assert look("0000:C1:00.0") == (0, 1, "48-95,144-191")
    assert look("0000:05:00.0") == (0, 0, "0-47,96-143")
    assert look("0000:85:00.0") == (0, -1, "")             # the platform does not say
    assert look("0000:FF:00.0") == (0, -1, "")             # no such device: no information, not an error
    (root / "sys/devices/system/node/node0/cpulist").write_text("0-47,x\n")
    rc, node, _ = look("0000:05:00.0")
    assert rc == L.E_INVALID and node == -1 and b"cannot parse" in lib.mpb_last_error()
    # the real tree: never an error, whatever this machine exposes
    rc, node, cpus = look("0000:00:00.0", "")
    assert rc == 0 and node >= -1
