"""The N>1 path on CPU: world_size-2 (and 3) gloo processes run the host-side split/gather of
moira_amd.shard with the oracle standing in for the per-rank filter.  Checks the partition,
order-preserving gather and the totals all-reduce -- everything except the HIP kernel itself,
which the -m gpu tests cover."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

from moira_amd.shard import shard_bounds


def test_shard_bounds_partition():
    for n in (0, 1, 7, 64, 1000, 1001):
        for w in (1, 2, 3, 8):
            b = [shard_bounds(n, w, r) for r in range(w)]
            assert b[0][0] == 0 and b[-1][1] == n
            assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
            sizes = [hi - lo for lo, hi in b]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        shard_bounds(10, 2, 2)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for p in (root, os.path.join(root, "oracle")):
        if p not in sys.path:
            sys.path.insert(0, p)
    import torch.distributed as dist
    import pb_oracle as O
    from moira_amd.shard import filter_sharded
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, lens = O.synth_fill(n, 320, fixed_len=300, seed=7)

    def fn(qq, ll):
        ee, ns, ps, _ = O.filter_batch(qq, lens=ll, threads=1)
        return ee, ns, ps.astype(bool)

    ee, ns, passed, totals = filter_sharded(q, lens, fn, dist=dist, gather=True)
    ee_l, _, _, totals_l = filter_sharded(q, lens, fn, dist=dist, gather=False)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), ee=ee, ns=ns, passed=passed,
             totals=np.array(totals), local_n=len(ee_l), totals_l=np.array(totals_l))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_filter_matches_single_process(tmp_path, oracle, world):
    n = 1001
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    q, lens = oracle.synth_fill(n, 320, fixed_len=300, seed=7)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens)
    local = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(z["ee"], ee) and np.array_equal(z["ns"], ns)
        assert np.array_equal(z["passed"], ps.astype(bool))
        assert tuple(z["totals"]) == (int(ps.sum()), n - int(ps.sum())) == tuple(z["totals_l"])
        local += int(z["local_n"])
    assert local == n
