#!/usr/bin/env python3
"""Calibration for the host-fed rate: what one plain hipMemcpy of a pinned buffer reaches on this box
(H2D alone, D2H alone), i.e. the ceiling mpb_filter_host's pipeline can approach."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

nbytes = 1 << 30
with Engine(0) as eng:
    pin = eng.host_alloc(nbytes, np.uint8)
    pin[:] = 7
    page = np.full(nbytes, 7, np.uint8)
    d = eng.alloc(nbytes)
    for name, h in (("pinned", pin), ("pageable", page)):
        for direction in ("h2d", "d2h"):
            best = 1e9
            for _ in range(5):
                t = time.perf_counter()
                if direction == "h2d":
                    eng.lib.mpb_memcpy_h2d(eng.ctx, d.ptr, h.ctypes.data, nbytes)
                else:
                    eng.lib.mpb_memcpy_d2h(eng.ctx, h.ctypes.data, d.ptr, nbytes)
                best = min(best, time.perf_counter() - t)
            print("%s %s: %.2f GB/s (1 GiB, best of 5)" % (name, direction, nbytes / best / 1e9), flush=True)
    d.free()
    eng.host_free(pin)
