#!/bin/bash
# On the GPU box: rebuild the library with different DP register budgets and time each (experiment only).
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -x hip"
cp moira_amd/libmoira_pb.so /tmp/libmoira_pb.orig.so
for round in 1 2; do for W in 4 3 5 2; do
  /opt/rocm/bin/hipcc $FL -DMPB_DP_WAVES_PER_EU=$W moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp -o moira_amd/libmoira_pb.so 2>/dev/null
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('waves_per_eu=$W', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"
done; done
cp /tmp/libmoira_pb.orig.so moira_amd/libmoira_pb.so
