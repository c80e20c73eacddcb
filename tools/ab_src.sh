#!/bin/bash
# On the GPU box: interleaved A/B of two kernel sources (old = gpurun_tmp_old_*, new = tree).
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -x hip"
mkdir -p /tmp/ab/old/moira_amd/csrc /tmp/ab/old/include
cp include/*.h /tmp/ab/old/include/; cp moira_amd/csrc/mpb_api.cpp /tmp/ab/old/moira_amd/csrc/
cp gpurun_tmp_old_kernels.hip /tmp/ab/old/moira_amd/csrc/mpb_kernels.hip; cp gpurun_tmp_old_internal.h /tmp/ab/old/moira_amd/csrc/mpb_internal.h
/opt/rocm/bin/hipcc $FL /tmp/ab/old/moira_amd/csrc/mpb_kernels.hip /tmp/ab/old/moira_amd/csrc/mpb_api.cpp -o /tmp/ab/old.so 2>/dev/null
cp moira_amd/libmoira_pb.so /tmp/ab/new.so
for i in 1 2 3; do for v in old new; do
  cp /tmp/ab/$v.so moira_amd/libmoira_pb.so
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"
done; done
cp /tmp/ab/new.so moira_amd/libmoira_pb.so
