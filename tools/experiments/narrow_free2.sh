#!/bin/bash
# (written when the {p}-only table and a three-slot ring were the defaults: -DMPB_NAR_LUT128 is now a no-op, -DMPB_NAR_LUT64 /
# -DMPB_NAR_DEPTH=3 select the old forms)
# Run ON THE GPU BOX: the three {1-p,p}-table candidates of k_narrow at R = 2, 3, 4 (interleaved, three rounds)
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
build() { /opt/rocm/bin/hipcc $FL $2 moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$1.so 2>/tmp/var/$1.err || { tail -5 /tmp/var/$1.err; exit 1; }; }
build late_128 "-DMPB_NAR_LATE_FREE -DMPB_NAR_LUT128"; build early_128 "-DMPB_NAR_LUT128"; build early_d2_128 "-DMPB_NAR_DEPTH=2 -DMPB_NAR_LUT128"; build late_d2_128 "-DMPB_NAR_LATE_FREE -DMPB_NAR_DEPTH=2 -DMPB_NAR_LUT128"
for rep in 1 2 3; do for R in 2 3 4; do for v in late_128 early_128 early_d2_128 late_d2_128; do
  echo "$v: $(MOIRA_PB_LIB=/tmp/var/$v.so python3 tools/narrow_probe.py $R 10000000 2>&1 | tail -1)"
done; done; done
