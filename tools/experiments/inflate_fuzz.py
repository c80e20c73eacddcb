import sys, io, gzip, zlib
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path[:0] = [R, os.path.join(R, 'tests')]
import numpy as np
from moira_amd import fastio as F
import test_inflate as T
rng = np.random.default_rng(99)
pay = [p for _, p in T.payloads() if len(p) > 1000]
errs = ok = silent = 0
for it in range(4000):
    data = pay[it % len(pay)][:int(rng.integers(2000, 60000))]
    lvl = int(rng.integers(0, 10))
    comp = bytearray(gzip.compress(data, lvl))
    kind = it % 4
    if kind == 0:      # bit flips
        for _ in range(int(rng.integers(1, 4))):
            p = int(rng.integers(10, len(comp))); comp[p] ^= 1 << int(rng.integers(0, 8))
    elif kind == 1:    # random bytes overwrite
        p = int(rng.integers(10, len(comp) - 4)); comp[p:p + 4] = bytes(rng.integers(0, 256, 4, dtype=np.uint8))
    elif kind == 2:    # truncate
        comp = comp[:int(rng.integers(0, len(comp)))]
    else:              # garbage deflate stream after a valid header
        comp = comp[:10] + bytes(rng.integers(0, 256, int(rng.integers(1, 3000)), dtype=np.uint8))
    try:
        out = T.decode(bytes(comp), read_size=int(rng.choice([50, 1000, 1 << 20])), in_block=int(rng.choice([4096, 5000, 1 << 16])))
        if out == data: ok += 1
        else: silent += 1
    except OSError:
        errs += 1
print("inflate fuzz under ASan+UBSan: 4000 damaged streams: %d rejected, %d still decoded to the original (damage in a don't-care field), %d silently wrong" % (errs, ok, silent))
sys.exit(1 if silent else 0)
