#!/bin/bash
# Run ON THE GPU BOX: interleaved comparison of two prebuilt libraries:  tools/ab_so.sh old.so [rounds]
# (the tree's moira_amd/libmoira_pb.so is "new").
OLD=$1; N=${2:-3}
cp moira_amd/libmoira_pb.so /tmp/ab_new.so
for i in $(seq $N); do for v in old new; do
  if [ $v = old ]; then cp $OLD moira_amd/libmoira_pb.so; else cp /tmp/ab_new.so moira_amd/libmoira_pb.so; fi
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"
done; done
cp /tmp/ab_new.so moira_amd/libmoira_pb.so
