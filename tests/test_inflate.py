"""libmoira_io's own gzip / DEFLATE decoder (csrc/inflate.cpp, written from RFC 1951 / 1952) against zlib: every
compression level and strategy, stored / fixed / dynamic blocks, multi-member files, padding, chunk boundaries at every
offset of input and output, truncated and corrupt input.  CPU only."""
import gzip
import io
import os
import zlib

import numpy as np
import pytest

from moira_amd import fastio as F


def fastq_text(rng, n):
    out = []
    for i in range(n):
        L = int(rng.integers(30, 260))
        out.append(b"@M0:%d:x%d\n" % (i % 7, i) + bytes(rng.choice(np.frombuffer(b"ACGTN", np.uint8), L, p=[.245, .245, .245, .245, .02]))
                   + b"\n+\n" + bytes((np.clip(38 - (np.arange(L) / L) ** 3 * rng.integers(4, 30) - rng.integers(0, 6, L), 2, 40) + 33).astype(np.uint8)) + b"\n")
    return b"".join(out)


def payloads():
    rng = np.random.default_rng(5)
    yield "fastq", fastq_text(rng, 4000)
    yield "random", bytes(rng.integers(0, 256, 300_000, dtype=np.uint8))
    yield "zeros", bytes(200_000)
    yield "runs", b"".join(bytes([int(rng.integers(0, 256))]) * int(rng.integers(1, 700)) for _ in range(2000))
    yield "short_period", (b"abcde" * 50_000) + (b"xy" * 70_000) + (b"0123456" * 30_000)
    yield "far_matches", b"".join([bytes(rng.integers(0, 256, 20_000, dtype=np.uint8))] * 12)        # distances ~20,000
    yield "tiny", b"A"
    yield "empty", b""
    yield "text", (b"The quick brown fox jumps over the lazy dog. " * 3000)
    yield "two_symbols", bytes(rng.integers(0, 2, 150_000, dtype=np.uint8))


def compress_variants(data):
    for level in range(0, 10):
        yield "level%d" % level, gzip.compress(data, compresslevel=level)
    for name, strat in (("filtered", zlib.Z_FILTERED), ("huffman_only", zlib.Z_HUFFMAN_ONLY), ("rle", zlib.Z_RLE), ("fixed", zlib.Z_FIXED)):
        c = zlib.compressobj(6, zlib.DEFLATED, 31, 9, strat)
        yield name, c.compress(data) + c.flush()
    c = zlib.compressobj(1, zlib.DEFLATED, 31, 1)                        # memLevel 1: many small blocks
    yield "small_blocks", c.compress(data) + c.flush()
    c = zlib.compressobj(6, zlib.DEFLATED, 31)                           # sync flushes: empty stored blocks in between
    parts = [c.compress(data[a:a + 10_000]) + c.flush(zlib.Z_SYNC_FLUSH) for a in range(0, len(data), 10_000)]
    yield "sync_flushes", b"".join(parts) + c.flush()


def decode(comp, read_size=1 << 20, in_block=1 << 16):
    r = F.GzipReader(io.BytesIO(comp), in_block=in_block)
    out = []
    while True:
        b = r.read(read_size)
        if not b:
            break
        out.append(b)
    r.close()
    return b"".join(out)


@pytest.mark.parametrize("name,data", list(payloads()), ids=[p[0] for p in payloads()])
def test_every_level_and_strategy(name, data):
    for vname, comp in compress_variants(data):
        assert zlib.decompress(comp, 31) == data
        got = decode(comp)
        assert got == data, (name, vname, len(got), len(data))


def test_chunk_boundaries_everywhere():
    """Input handed over in pieces of 4096..4200 bytes and output taken 1, 7, 300 or 65,537 bytes at a time: the decoder is
    stopped and resumed at (nearly) every phase and bit position."""
    rng = np.random.default_rng(9)
    data = fastq_text(rng, 600) + bytes(rng.integers(0, 256, 5000, dtype=np.uint8)) + b"z" * 9000
    for level in (0, 1, 6, 9):
        comp = gzip.compress(data, compresslevel=level)
        for in_block in (4096, 4097, 4133, 4200):
            for read_size in (1, 7, 300, 65_537):
                if read_size == 1 and in_block != 4096:
                    continue
                assert decode(comp, read_size, in_block) == data, (level, in_block, read_size)


def test_multi_member_and_padding():
    rng = np.random.default_rng(10)
    a, b, c = fastq_text(rng, 300), b"", bytes(rng.integers(0, 256, 70_000, dtype=np.uint8))
    comp = gzip.compress(a, 9) + gzip.compress(b) + gzip.compress(c, 1) + bytes(37)
    assert gzip.decompress(comp) == a + b + c
    assert decode(comp) == a + b + c
    assert decode(comp, read_size=11, in_block=4096) == a + b + c
    # header fields: name, comment, extra, header CRC
    buf = io.BytesIO()
    with gzip.GzipFile(filename="reads.fastq", mode="wb", fileobj=buf, mtime=12345) as g:
        g.write(a)
    assert decode(buf.getvalue()) == a
    raw = gzip.compress(a)
    with_fields = raw[:3] + bytes([4 | 8 | 16 | 2]) + raw[4:10] + b"\x05\x00hello" + b"name\x00" + b"a comment\x00" + b"\x12\x34" + raw[10:]
    assert decode(with_fields) == a


def test_corrupt_and_truncated_input_is_an_error_not_garbage():
    rng = np.random.default_rng(11)
    data = fastq_text(rng, 200)
    comp = gzip.compress(data, 6)
    for cut in list(range(0, 40)) + list(range(len(comp) - 20, len(comp))) + [len(comp) // 2, len(comp) // 3]:
        with pytest.raises(OSError):
            decode(comp[:cut])
    bad_crc = comp[:-8] + bytes([comp[-8] ^ 1]) + comp[-7:]
    with pytest.raises(OSError, match="CRC"):
        decode(bad_crc)
    bad_len = comp[:-1] + bytes([comp[-1] ^ 0x40])
    with pytest.raises(OSError, match="length"):
        decode(bad_len)
    with pytest.raises(OSError, match="not a gzip"):
        decode(b"plain text, not gzip at all" * 10)
    with pytest.raises(OSError):
        decode(b"")
    # flipped bits inside the stream: an error or a CRC failure, never a silent wrong result
    silent = 0
    for k in range(200):
        pos = int(rng.integers(12, len(comp) - 9))
        broken = comp[:pos] + bytes([comp[pos] ^ (1 << int(rng.integers(0, 8)))]) + comp[pos + 1:]
        try:
            out = decode(broken)
            silent += out != data
        except OSError:
            pass
    assert silent == 0


def test_crc32_is_zlibs():
    rng = np.random.default_rng(12)
    L = F.load()
    for n in (0, 1, 7, 8, 9, 63, 1000, 100_003):
        a = rng.integers(0, 256, n, dtype=np.uint8)
        assert L.mio_crc32(0, a.ctypes.data, n) == zlib.crc32(a.tobytes())
    a = rng.integers(0, 256, 5000, dtype=np.uint8)
    c = L.mio_crc32(0, a.ctypes.data, 1234)
    assert L.mio_crc32(c, a.ctypes.data + 1234, 5000 - 1234) == zlib.crc32(a.tobytes())
    # the carry-less-multiply form folds 64 bytes per step from 128 bytes on: every length around its boundaries, odd
    # starting offsets, non-zero starting values
    b = rng.integers(0, 256, 70_000, dtype=np.uint8)
    raw = b.tobytes()
    for n in list(range(100, 340)) + [4095, 4096, 4097, 65_535, 65_536, 69_990]:
        for off in (0, 1, 5):
            for init in (0, 0xDEADBEEF):
                assert L.mio_crc32(init, b.ctypes.data + off, n) == zlib.crc32(raw[off:off + n], init), (n, off, init)


def test_crc32_table_form_is_the_same():
    """MOIRA_CRC_TABLES=1 keeps the slice-by-8 form (what a CPU without PCLMULQDQ runs): same values, in a fresh process."""
    import subprocess
    import sys
    code = ("import zlib, numpy as np; from moira_amd import fastio as F; L = F.load(); "
            "a = np.random.default_rng(5).integers(0, 256, 300000, dtype=np.uint8); "
            "assert all(L.mio_crc32(0, a.ctypes.data + o, n) == zlib.crc32(a.tobytes()[o:o + n]) for n in (0, 5, 127, 128, 200, 299000) for o in (0, 3)); "
            "print('ok')")
    for env_extra in ({"MOIRA_CRC_TABLES": "1"}, {}):
        out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **env_extra), capture_output=True, text=True,
                             cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        assert out.returncode == 0 and out.stdout.strip() == "ok", out.stderr


def test_large_file_and_rate(tmp_path):
    """40 MB of FASTQ text through the reader exactly as the CLI uses it (32 MiB reads); also prints the rate against
    Python's gzip module (informational)."""
    import time
    rng = np.random.default_rng(13)
    block = fastq_text(rng, 20_000)
    data = b"".join(block[i:] + block[:i] for i in (0, 1001, 20_003, 77_777, 123_457, 300_001, 424_243, 999_331))
    path = tmp_path / "big.fastq.gz"
    with gzip.open(path, "wb", compresslevel=6) as f:
        f.write(data)
    t = time.perf_counter()
    with F.GzipReader(open(path, "rb")) as r:
        got = []
        while True:
            b = r.read(1 << 25)
            if not b:
                break
            got.append(b)
    dt = time.perf_counter() - t
    assert b"".join(got) == data
    t = time.perf_counter()
    with gzip.open(path, "rb") as f:
        ref = f.read()
    dz = time.perf_counter() - t
    assert ref == data
    print("\ninflate: %.0f MB/s (zlib through gzip.open: %.0f MB/s) on %d MB of FASTQ text" % (len(data) / dt / 1e6, len(data) / dz / 1e6, len(data) >> 20))


# ---- BGZF: members that state their size are inflated on several threads -------------------------------------------------

def decode_mt(comp, threads, read_size=1 << 20, in_block=1 << 17):
    r = F.GzipReader(io.BytesIO(comp), in_block=in_block, threads=threads)
    out = []
    while True:
        b = r.read(read_size)
        if not b:
            break
        out.append(b)
    batches = r.bgzf_batches
    r.close()
    return b"".join(out), batches


def test_bgzf_members_are_standard_gzip_and_state_their_size():
    from moira_amd.cli import BGZF_DATA, BGZF_EOF, bgzf_compress
    rng = np.random.default_rng(41)
    text = fastq_text(rng, 3000)
    z = bgzf_compress(text)
    assert gzip.decompress(z + BGZF_EOF) == text and gzip.decompress(BGZF_EOF) == b""
    pos = members = 0
    while pos < len(z):                                          # walk by the stated sizes alone
        assert z[pos:pos + 4] == b"\x1f\x8b\x08\x04" and z[pos + 10:pos + 16] == b"\x06\x00BC\x02\x00"
        size = int.from_bytes(z[pos + 16:pos + 18], "little") + 1
        assert size <= 65536
        piece = zlib.decompress(z[pos + 18:pos + size - 8], -15)
        assert piece == text[members * BGZF_DATA:(members + 1) * BGZF_DATA]
        assert int.from_bytes(z[pos + size - 4:pos + size], "little") == len(piece)
        pos += size
        members += 1
    assert pos == len(z) and members == -(-len(text) // BGZF_DATA)
    noise = rng.integers(0, 256, 200_000, dtype=np.uint8).tobytes()       # incompressible data still fits the size field
    assert gzip.decompress(bgzf_compress(noise)) == noise


@pytest.mark.parametrize("threads", [1, 2, 5])
def test_bgzf_input_on_several_threads(threads):
    from moira_amd.cli import BGZF_EOF, bgzf_compress
    rng = np.random.default_rng(42)
    text = fastq_text(rng, 6000)
    z = bgzf_compress(text) + BGZF_EOF
    for read_size in (1 << 20, 70_000, 999):
        got, batches = decode_mt(z, threads, read_size)
        assert got == text
        assert (batches > 0) == (threads > 1)
    # an input block barely larger than a member: members straddle every refill
    got, batches = decode_mt(z, threads, 1 << 16, in_block=(1 << 17) + 13)
    assert got == text
    # only the end marker / nothing after the last member / zero padding after it
    assert decode_mt(BGZF_EOF, threads) == (b"", 1 if threads > 1 else 0)
    assert decode_mt(bgzf_compress(text), threads)[0] == text
    assert decode_mt(z + b"\0" * 1000, threads)[0] == text
    with pytest.raises(OSError):
        decode_mt(b"", threads)


def test_bgzf_mixed_with_ordinary_members():
    """A file may change convention between members (cat of two files): the parallel path hands over to the one-stream
    decoder at the first member that is not BGZF and the text is the same."""
    from moira_amd.cli import bgzf_compress
    rng = np.random.default_rng(43)
    text = fastq_text(rng, 5000)
    a, b = len(text) // 3, 2 * len(text) // 3
    z = bgzf_compress(text[:a]) + gzip.compress(text[a:b]) + bgzf_compress(text[b:])
    got, batches = decode_mt(z, 4, 50_000)
    assert got == text and batches >= 1
    # an extra field that is not the BGZF one, and a BGZF-looking member with a second subfield in front
    c = zlib.compressobj(6, zlib.DEFLATED, -15)
    raw = c.compress(text[:5000]) + c.flush()
    trailer = zlib.crc32(text[:5000]).to_bytes(4, "little") + (5000).to_bytes(4, "little")
    other = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x06\x00XY\x02\x00\x07\x07" + raw + trailer
    assert decode_mt(other, 4) == (text[:5000], 0)
    two = b"\x1f\x8b\x08\x04\0\0\0\0\0\xff\x0b\x00XY\x01\x00\x07BC\x02\x00" + (len(raw) + 30).to_bytes(2, "little") + raw + trailer
    assert decode_mt(two + two, 4) == (text[:5000] * 2, 1)


def test_bgzf_damage_is_an_error_not_garbage():
    from moira_amd.cli import BGZF_EOF, bgzf_compress
    rng = np.random.default_rng(44)
    text = fastq_text(rng, 4000)
    z = bytearray(bgzf_compress(text) + BGZF_EOF)
    size0 = int.from_bytes(z[16:18], "little") + 1
    for what in ("data", "crc", "isize", "bsize_short", "bsize_long", "truncated"):
        bad = bytearray(z)
        if what == "data":
            bad[size0 + 200] ^= 0x55                              # inside the second member's deflate data
        elif what == "crc":
            bad[size0 - 8] ^= 1
        elif what == "isize":
            bad[size0 - 4] ^= 1
        elif what == "bsize_short":
            bad[16:18] = (size0 - 1 - 40).to_bytes(2, "little")
        elif what == "bsize_long":
            bad[16:18] = (size0 - 1 + 40).to_bytes(2, "little")
        else:
            bad = bad[:len(bad) - len(BGZF_EOF) - 11]
        for threads in (1, 4):
            try:
                got, _ = decode_mt(bytes(bad), threads)
            except OSError:
                continue
            # a wrong size FIELD around sound data: a decoder that never needed the field (one stream, or the parallel
            # path once it has handed over) still yields the text; anything else must have been refused
            assert what.startswith("bsize") and got == text, "%s (threads %d): decoded %d bytes without an error" % (what, threads, len(got))


def test_cli_reads_bgzf_input_like_plain_input(tmp_path, oracle):
    """The CLI on a BGZF-compressed FASTQ with -p 4 (members inflated on several threads) writes what it writes for the
    uncompressed file, and its own .gz outputs are BGZF: they decode on the parallel path."""
    from moira_amd import cli
    from moira_amd.cli import BGZF_EOF, bgzf_compress
    from test_cli_golden import oracle_backend, reference_args
    rng = np.random.default_rng(45)
    text = fastq_text(rng, 2500)
    (tmp_path / "r.fastq").write_bytes(text)
    (tmp_path / "z.fastq.gz").write_bytes(bgzf_compress(text) + BGZF_EOF)
    outs = {}
    for label, src, comp in (("plain", "r.fastq", "none"), ("bgzf", "z.fastq.gz", "gz")):
        pre = str(tmp_path / ("o_" + label))
        a = reference_args(paired=False, forward_fastq=str(tmp_path / src), output_prefix=pre, collapse=True,
                           output_compression=comp, processors=4)
        assert cli.main(a, backend=oracle_backend(oracle), out=open(os.devnull, "w")) == 0
        outs[label] = {}
        for f in sorted(os.listdir(tmp_path)):
            if f.startswith("o_" + label + "."):
                data = (tmp_path / f).read_bytes()
                if comp == "gz":
                    data, batches = decode_mt(data, 4)
                    assert batches >= 1
                outs[label][f.split(".", 1)[1].replace(".gz", "")] = data
    assert outs["plain"].keys() == outs["bgzf"].keys() and len(outs["plain"]) >= 6
    assert outs["plain"] == outs["bgzf"]


def test_end_of_file_found_just_after_a_full_input_block():
    """the decoder may stop short of a block header / trailer it cannot see whole until it is told
    that the input ends; when the read that discovers the end of the file comes right after such a stop, the reader must
    call the decoder once more with final = 1 instead of reporting a truncated file.  Sweep: compressed size = k x in_block
    + 1 .. 400 bytes, ending in a tiny last block (after a full flush), single-member and with a BGZF-style empty member
    at the end; every one is a VALID gzip file."""
    rng = np.random.default_rng(11)
    in_block = 4096
    body = bytes(rng.integers(0, 256, 3 * in_block, dtype=np.uint8))          # incompressible: the size is steerable
    empty_member = gzip.compress(b"")
    checked = 0
    for k in (1, 2):
        for extra in list(range(1, 60)) + list(range(60, 401, 7)):
            for tail_kind in ("tiny_block", "empty_member"):
                target = k * in_block + extra
                # a stored-ish first part sized so that the whole file lands on `target` bytes
                for n_body in range(target - 60, target - 10):
                    c = zlib.compressobj(1, zlib.DEFLATED, 31)
                    comp = c.compress(body[:n_body]) + c.flush(zlib.Z_FULL_FLUSH) + c.compress(b"xy") + c.flush()
                    if tail_kind == "empty_member":
                        comp += empty_member
                    if len(comp) == target:
                        break
                else:
                    continue
                want = body[:n_body] + b"xy"
                assert gzip.decompress(comp) == want
                assert decode(comp, read_size=1 << 16, in_block=in_block) == want, (target, tail_kind)
                checked += 1
    assert checked > 150
    # and a file that really is cut short still fails
    c = zlib.compressobj(1, zlib.DEFLATED, 31)
    comp = c.compress(body[:in_block]) + c.flush()
    with pytest.raises(OSError):
        decode(comp[:-5], in_block=in_block)
