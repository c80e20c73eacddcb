#!/bin/bash
# Run ON THE GPU BOX: k_narrow time against resident workgroups and ring depth (no profiler attached).
export MPB_NAR_NO_RS=1          # the ring form (k_narrow): at this stride the library would take k_narrow_rs
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
for d in 2 3 4; do
  /opt/rocm/bin/hipcc $FL -DMPB_TUNING_KNOBS -DMPB_NAR_DEPTH=$d moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/knobs_d$d.so || exit 1
done
for rep in 1 2; do
for spec in "3 256" "3 384" "3 512" "3 640" "3 768" "2 512" "2 768" "2 1024" "4 256" "4 384" "4 512"; do
  set -- $spec
  export MOIRA_PB_LIB=/tmp/var/knobs_d$1.so MPB_NAR_GRID=$2
  echo "depth $1 grid $2: $(python3 tools/narrow_probe.py ${R:-2} 10000000 2>&1 | tail -1)"
done; done
