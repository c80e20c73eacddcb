"""BGZF (blocked gzip) input: text rate of moira_amd.fastio.GzipReader by thread count, against the one-stream decoder
on the same text compressed as one member."""
import gzip, io, os, struct, sys, time, zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from moira_amd import fastio as F          # noqa: E402
from moira_amd.cli import bgzf_compress    # noqa: E402


def text(n, seed=1):
    rng = np.random.default_rng(seed)
    seq = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 150))]
    q = rng.integers(35, 74, (n, 150), dtype=np.uint8)
    recs = []
    for i in range(n):
        recs.append(b"@read%d/1\n%s\n+\n%s\n" % (i, seq[i].tobytes(), q[i].tobytes()))
    return b"".join(recs)


def drain(r):
    n = 0
    while True:
        d = r.read(1 << 25)
        if not d:
            return n
        n += len(d)


def main():
    n = int(os.environ.get("BGZF_READS", 1000000))
    txt = text(n)
    z = bgzf_compress(txt)
    one = gzip.compress(txt, 4)
    print("text %.0f MB, bgzf %.0f MB, one member %.0f MB" % (len(txt) / 1e6, len(z) / 1e6, len(one) / 1e6))
    for label, data, ths in (("one member", one, (1,)), ("bgzf", z, (1, 2, 4, 8, 16))):
        for th in ths:
            best = 1e9
            for _ in range(3):
                r = F.GzipReader(io.BytesIO(data), threads=th)
                t = time.perf_counter()
                got = drain(r)
                best = min(best, time.perf_counter() - t)
                assert got == len(txt)
            print("%-10s threads %2d: %.3f s  %.0f MB/s of text  %.2e reads/s" % (label, th, best, len(txt) / best / 1e6, n / best))


if __name__ == "__main__":
    main()
