#!/usr/bin/env python3
"""BASELINE configs[2] at its STATED size, once: 100 chunks of 1 M synthetic 2 x 300-base pairs streamed through
the pipeline of bench.py's extras.config3_paired (FASTQ text in memory -> index -> NW + consensus on the host cores -> pack ->
GPU filter from host memory, index + contigs of chunk k+1 beside pack + filter of chunk k).  Prints the measured wall for
100 M pairs beside the projection the 4-chunk extra makes.  Not part of the driver-run bench (it takes over a minute).

    python tools/config3_full.py [chunks] [pairs_per_chunk]      -> one JSON object
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from moira_amd.engine import Engine  # noqa: E402

chunks = int(sys.argv[1]) if len(sys.argv) > 1 else 100
per = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
with Engine(0) as eng:
    short = bench.config3_paired_rate(eng, per, 4)
    sys.stderr.write("4 chunks: %.3g pairs/s pipelined, projected %.1f s for 100 M pairs\n"
                     % (short["pipelined"]["pairs_per_s"], short["projected_wall_s_for_100M_pairs"]))
    full = bench.config3_paired_rate(eng, per, chunks, stage_chunks=4)
print(json.dumps({"pairs": full["pairs"], "measured_wall_s": full["pipelined"]["wall_s"], "pairs_per_s": full["pipelined"]["pairs_per_s"],
                  "projection_from_4_chunks_s": short["projected_wall_s_for_100M_pairs"] * full["pairs"] / 1e8,
                  "host_threads": full["host_threads"], "contigs_kept": full["contigs_kept"],
                  "stage_pairs_per_s": full["stage_pairs_per_s"], "gpu_share_of_the_wall": full["gpu_share_of_the_pipelined_wall"],
                  "note": full["note"]}, indent=1))
