"""Drop-in for moira's CPython extension `bernoulli` (moira/bernoullimodule.c).

Put this directory on PYTHONPATH ahead of the compiled extension and moira.py's
`import bernoulli` (moira/moira.py:247-251) picks it up unchanged:

    bernoulli.calculate_errors_PB(contig, contig_quals, alpha) -> (expected_errors, Ns)

Same signature, same exceptions (moira/bernoullimodule.c:74-90); computed on the MI355X by
libmoira_pb.so.  `calculate_errors` is the alias BASELINE.json's north_star names.
A per-read call pays a kernel launch; throughput comes from the batch API
(moira_amd.engine.Engine.filter), which is what replaces moira's per-read Pool dispatch.

Under an unchanged `moira.py --processors P` the callers are the P worker processes of a multiprocessing.Pool
(moira/moira.py:398-399,431-454).  Those do not open P GPU contexts (which would time-share the card): a process that
was started by multiprocessing attaches to ONE GPU-owning broker process (moira_amd/broker.py; started on demand as a
fresh child before this process has touched the GPU), which micro-batches whatever the workers have pending into one
launch.  MOIRA_PB_BROKER=1 / 0 forces / forbids the broker; MOIRA_PB_DEVICE picks the GPU (default 0; a list such as
0,1,2,3 spreads the worker processes over one broker per GPU).
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from moira_amd.engine import default_engine as _default_engine  # noqa: E402

__doc__ = ("This module provides an interface for calculating the expected errors of a given sequence "
           "using a sum of Bernoulli random variables.")


_entry = {"pid": None, "fn": None}


def _choose():
    """The per-read entry of THIS process: the broker for a multiprocessing worker, a context of its own otherwise."""
    import multiprocessing
    mode = os.environ.get("MOIRA_PB_BROKER", "auto").lower()
    use_broker = mode in ("1", "on", "yes", "true") or (mode == "auto" and multiprocessing.parent_process() is not None)
    if not use_broker:
        return _default_engine().calculate_errors_PB
    from moira_amd import broker as _broker
    devices = [int(d) for d in os.environ.get("MOIRA_PB_DEVICE", "0").split(",") if d.strip() != ""] or [0]
    device = devices[os.getpid() % len(devices)]       # "0,1,...,7": the workers spread over one broker per GPU
    try:
        state = {"cl": _broker.client(device)}
    except MemoryError:                                # more worker processes than the broker has slots (64): this one
        return _default_engine().calculate_errors_PB   # takes a context of its own (slower under contention, still exact)

    def call(contig, contig_quals, alpha):
        try:
            return state["cl"].calculate_errors_PB(contig, contig_quals, alpha)
        except _broker.BrokerGone:                     # the broker died or left: start / find another one, once
            state["cl"].close()
            state["cl"] = _broker.client(device)
            return state["cl"].calculate_errors_PB(contig, contig_quals, alpha)
    return call


def calculate_errors_PB(contig, contig_quals, alpha):
    """This function returns the expected errors of a given sequence with a given confidence value
    using a sum of Bernoulli random variables."""
    pid = os.getpid()
    if _entry["pid"] != pid:                           # first call in this process (a forked worker decides for itself)
        _entry["fn"], _entry["pid"] = _choose(), pid
    return _entry["fn"](contig, contig_quals, alpha)


calculate_errors = calculate_errors_PB


def calculate_errors_PB_batch(quals, lens=None, alpha=0.005, fixed_len=None):
    """Batch form: `quals` is a C-contiguous uint8 (N x L_max) matrix in the packed
    encoding of include/moira_pb.h (Q0 already clamped to 1, 0 = 'N', 255 = 'n'), `lens` an optional
    int32 vector.  Returns (expected_errors float64[N], Ns int32[N]) -- per read exactly what
    calculate_errors_PB returns, one library call for the whole matrix."""
    r = _default_engine().filter(quals, lens=lens, fixed_len=fixed_len, alpha=alpha, ambigs="ignore", uncert=1.0)
    return r.ee, r.ns
