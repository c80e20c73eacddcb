"""Host-side microbenchmark: gzip / zlib block compression against the thread count, and cli._BlockCompressedWriter."""
import sys, time, os, gzip, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
from moira_amd import cli
W = cli._BlockCompressedWriter
rng = np.random.default_rng(1)
n = 500000
bases = np.frombuffer(b"ACGT", np.uint8)[rng.integers(0, 4, (n, 250))]
quals = (rng.integers(2, 41, (n, 250)) + 33).astype(np.uint8)
rec = np.empty((n, 512), np.uint8)
rec[:, :7] = np.frombuffer(b"@r00000", np.uint8); rec[:, 7] = 10; rec[:, 8:258] = bases
rec[:, 258:261] = np.frombuffer(b"\n+\n", np.uint8); rec[:, 261:511] = quals; rec[:, 511] = 10
data = rec.tobytes()[:240 << 20]
print("pool threads", W.pool()._max_workers, "data MB", len(data) >> 20)
parts = [data[a:a + (4 << 20)] for a in range(0, len(data), 4 << 20)]
for th in (1, 4, 8, 16, 32):
    with ThreadPoolExecutor(th) as p:
        t = time.perf_counter(); out = list(p.map(lambda b: gzip.compress(b, compresslevel=4), parts)); dt = time.perf_counter() - t
    print("gzip.compress level 4 x%d threads: %.0f MB/s (ratio %.2f)" % (th, len(data) / dt / 1e6, sum(map(len, out)) / len(data)))
with ThreadPoolExecutor(16) as p:
    t = time.perf_counter(); out = list(p.map(lambda b: zlib.compress(b, 1), parts)); dt = time.perf_counter() - t
print("zlib level 1 x16: %.0f MB/s (ratio %.2f)" % (len(data) / dt / 1e6, sum(map(len, out)) / len(data)))
mv = memoryview(data)
t = time.perf_counter()
w = W('/tmp/w.gz', 'gz')
for a in range(0, len(data), 32 << 20):
    w.write(mv[a:a + (32 << 20)])
w.close()
dt = time.perf_counter() - t
print("writer: %.2f s = %.0f MB/s" % (dt, len(data) / dt / 1e6))
