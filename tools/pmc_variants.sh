#!/bin/bash
# Run ON THE GPU BOX: one --pmc pass per compile-time variant.  tools/pmc_variants.sh "COUNTERS..." name:"-Dflags" ...
CTRS=$1; shift
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -x hip"
cp moira_amd/libmoira_pb.so /tmp/pv_orig.so
for spec in "$@"; do
  name=${spec%%:*}; defs=${spec#*:}
  /opt/rocm/bin/hipcc $FL $defs moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp -o moira_amd/libmoira_pb.so 2>/dev/null || { echo "build $name failed"; exit 1; }
  echo "== $name"
  tools/pmc_pass.sh v_$name $CTRS 2>&1 | grep "k_dp<false, false>\|k_prepass"
done
cp /tmp/pv_orig.so moira_amd/libmoira_pb.so
