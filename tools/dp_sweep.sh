#!/bin/bash
# A/B of the DP chunk size in one session on one GPU (tools only; not part of the product)
for cfg in "8" "1" "2" "4" "6" "8" "12" "16" "4"; do
  MPB_DP_CHUNK=$cfg python bench.py --steps 6 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('chunk=$cfg', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items()})"
done
