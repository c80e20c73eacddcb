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
    from moira_amd.shard import filter_sharded, filter_sharded_owned
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    q, lens = O.synth_fill(n, 320, fixed_len=300, seed=7)

    def fn(qq, ll):
        ee, ns, ps, _ = O.filter_batch(qq, lens=ll, threads=1)
        return ee, ns, ps.astype(bool)

    ee, ns, passed, totals = filter_sharded(q, lens, fn, dist=dist, gather=True)
    ee_l, _, _, totals_l = filter_sharded(q, lens, fn, dist=dist, gather=False)
    # owned form: this rank generates ONLY its own range (no rank ever holds the whole batch)
    asked = []

    def load(lo, hi):
        asked.append((lo, hi))
        return O.synth_fill(hi - lo, 320, fixed_len=300, seed=7, first_read=lo)

    ee_o, ns_o, ps_o, totals_o, (lo, hi) = filter_sharded_owned(n, load, fn, dist=dist, gather=False)
    ee_g, ns_g, ps_g, totals_g, _ = filter_sharded_owned(n, load, fn, dist=dist, gather=True)
    np.savez(os.path.join(out_dir, "r%d.npz" % rank), ee=ee, ns=ns, passed=passed,
             totals=np.array(totals), local_n=len(ee_l), totals_l=np.array(totals_l),
             asked=np.array(asked), lo=lo, hi=hi, ee_o=ee_o, ns_o=ns_o, ps_o=ps_o, totals_o=np.array(totals_o),
             ee_g=ee_g, ns_g=ns_g, ps_g=ps_g, totals_g=np.array(totals_g))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_filter_matches_single_process(tmp_path, oracle, world):
    n = 1001
    mp.spawn(_worker, args=(world, _free_port(), n, str(tmp_path)), nprocs=world, join=True)
    q, lens = oracle.synth_fill(n, 320, fixed_len=300, seed=7)
    ee, ns, ps, _ = oracle.filter_batch(q, lens=lens)
    want_totals = (int(ps.sum()), n - int(ps.sum()))
    local = 0
    for r in range(world):
        z = np.load(os.path.join(str(tmp_path), "r%d.npz" % r))
        assert np.array_equal(z["ee"], ee) and np.array_equal(z["ns"], ns)
        assert np.array_equal(z["passed"], ps.astype(bool))
        assert tuple(z["totals"]) == want_totals == tuple(z["totals_l"])
        local += int(z["local_n"])
        # owned form: the rank asked its loader for exactly its own range, both times, and nothing else
        lo, hi = shard_bounds(n, world, r)
        assert (int(z["lo"]), int(z["hi"])) == (lo, hi)
        assert [tuple(a) for a in z["asked"]] == [(lo, hi), (lo, hi)]
        assert np.array_equal(z["ee_o"], ee[lo:hi]) and np.array_equal(z["ns_o"], ns[lo:hi])
        assert np.array_equal(z["ps_o"], ps[lo:hi].astype(bool))
        assert tuple(z["totals_o"]) == want_totals == tuple(z["totals_g"])
        # padded fixed-width gather: full arrays in read order on every rank
        assert np.array_equal(z["ee_g"], ee) and np.array_equal(z["ns_g"], ns)
        assert np.array_equal(z["ps_g"], ps.astype(bool))
    assert local == n


def test_no_object_collectives_in_the_shard_module():
    """Results travel as fixed-width tensors (13 bytes per read), never as pickled Python objects."""
    import inspect
    import moira_amd.shard as S
    src = inspect.getsource(S)
    code = "\n".join(l for l in src.splitlines() if not l.lstrip().startswith("#"))
    assert "all_gather_object" not in code.split('"""', 2)[2] and "gather_object" not in code.split('"""', 2)[2]
    assert "all_gather_into_tensor" in code


def test_single_process_split_is_the_same_split(oracle):
    """mpb_filter_host_multi (one process, one host thread per GPU: MultiEngine) cuts a batch exactly as the
    process-per-GPU path does -- the C ABI's mpb_shard_bounds against shard.py's shard_bounds -- and filtering the
    shards one by one (the oracle standing in for the GPU) and laying the results side by side gives the unsplit
    batch's results, in read order, for every world size incl. more contexts than reads."""
    import ctypes as C
    from moira_amd import _lib as L
    from moira_amd.shard import filter_sharded, shard_bounds
    lib = L.load()
    lo, hi = C.c_int64(), C.c_int64()
    for n in (0, 1, 2, 7, 64, 1000, 1001, 12345, 10_000_000, 1_000_000_007):
        for world in (1, 2, 3, 5, 8, 13):
            covered = 0
            for r in range(world):
                assert lib.mpb_shard_bounds(n, world, r, C.byref(lo), C.byref(hi)) == 0
                assert (lo.value, hi.value) == shard_bounds(n, world, r)
                assert lo.value == covered
                covered = hi.value
            assert covered == n
    assert lib.mpb_shard_bounds(10, 0, 0, C.byref(lo), C.byref(hi)) == L.E_INVALID
    assert lib.mpb_shard_bounds(10, 2, 2, C.byref(lo), C.byref(hi)) == L.E_INVALID
    q, lens = oracle.synth_fill(2003, 608, min_len=50, max_len=600, seed=5)
    whole = oracle.filter_batch(q, lens=lens, threads=4)
    for world in (2, 3, 8):
        parts = []
        for r in range(world):
            a, b = shard_bounds(len(lens), world, r)
            e, s, p, _ = oracle.filter_batch(q[a:b], lens=lens[a:b], threads=2)
            parts.append((e, s, p))
        assert np.array_equal(np.concatenate([x[0] for x in parts]), whole[0])
        assert np.array_equal(np.concatenate([x[1] for x in parts]), whole[1])
        assert np.array_equal(np.concatenate([x[2] for x in parts]), whole[2])
