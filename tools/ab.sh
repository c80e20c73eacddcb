#!/bin/bash
# interleaved A/B of an environment knob in one session on one GPU:  tools/ab.sh VAR valueA valueB [rounds]
VAR=$1; A=$2; B=$3; N=${4:-3}
for i in $(seq $N); do for v in $A $B; do
  env $VAR=$v python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$VAR=$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"
done; done
