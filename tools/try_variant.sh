#!/bin/bash
# Run ON THE GPU BOX: build one compile-time variant in place, run the GPU parity tests against it, then
# time it and read the k_dp traffic counters; restores the tree's library afterwards.
#   tools/try_variant.sh "-DMPB_DP_PARK"
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -x hip"
cp moira_amd/libmoira_pb.so /tmp/tv_orig.so
trap 'cp /tmp/tv_orig.so moira_amd/libmoira_pb.so' EXIT
/opt/rocm/bin/hipcc $FL $1 moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp -o /tmp/tv_new.so 2>/dev/null || { echo "build failed"; exit 1; }
cp /tmp/tv_new.so moira_amd/libmoira_pb.so
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -3 || exit 1
for i in 1 2; do for v in orig new; do
  cp /tmp/tv_$v.so moira_amd/libmoira_pb.so
  python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print('$v', round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['kernels_ms_per_step'].items() if v})"
done; done
for v in orig new; do
  cp /tmp/tv_$v.so moira_amd/libmoira_pb.so
  echo "== $v"; tools/pmc_pass.sh tv_$v TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum 2>&1 | grep "k_dp<false, false>"
done
