import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import numpy as np
from moira_amd.engine import Engine
rng = np.random.default_rng(1)
with Engine(0) as eng:
    for L in (50, 150, 300, 600, 1000, 1024, 1025, 2000, 2047):
        seq = "".join(rng.choice(list("ACGT"), L))
        quals = [int(x) for x in np.clip(38 - (np.arange(L) / L) ** 3 * 20 - rng.integers(0, 6, L), 2, 40)]
        for _ in range(50): eng.calculate_errors_PB(seq, quals, 0.005)
        n = 2000
        t = time.perf_counter()
        for _ in range(n): eng.calculate_errors_PB(seq, quals, 0.005)
        dt = (time.perf_counter() - t) / n
        print("L = %4d: %.1f us per calculate_errors_PB call (in-process, resident server)" % (L, dt * 1e6))
