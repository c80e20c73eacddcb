import sys, time, gzip, zlib, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__)))))
import numpy as np
from moira_amd import fastio as F
L = F.load()
data = open(sys.argv[1],'rb').read(200 << 20)
for lvl in (1, 6):
    comp = gzip.compress(data, lvl)
    cin = np.frombuffer(comp + bytes(8), np.uint8)
    out = np.empty(len(data) + 1024, np.uint8)
    used, made = C.c_int64(0), C.c_int64(0)
    best = 1e9
    for rep in range(3):
        st = L.mio_inflate_create()
        t = time.perf_counter()
        rc = L.mio_inflate_gzip(st, cin.ctypes.data, len(comp), 1, out.ctypes.data, 0, len(out), C.addressof(used), C.addressof(made))
        dt = time.perf_counter() - t
        L.mio_inflate_destroy(st)
        best = min(best, dt)
    assert rc == 2 and made.value == len(data) and out[:len(data)].tobytes() == data
    t = time.perf_counter(); z = zlib.decompress(comp, 31); dz = time.perf_counter() - t
    t = time.perf_counter(); c = zlib.crc32(data); dc = time.perf_counter() - t
    t = time.perf_counter(); c2 = L.mio_crc32(0, out.ctypes.data, len(data)); dc2 = time.perf_counter() - t
    print("level %d: ratio %.2f | ours %.0f MB/s | zlib %.0f MB/s | crc32 zlib %.0f MB/s, ours %.0f MB/s" % (lvl, len(comp)/len(data), len(data)/best/1e6, len(data)/dz/1e6, len(data)/dc/1e6, len(data)/dc2/1e6))
