#!/bin/bash
# Run ON THE GPU BOX: k_narrow's arithmetic alone (NODMA) and the whole kernel, for instruction-order / table variants
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
build() { /opt/rocm/bin/hipcc $FL $2 moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$1.so 2>/tmp/var/$1.err || { tail -5 /tmp/var/$1.err; exit 1; }; }
build plain "-DMPB_NAR_PLAIN_ORDER"; build plain_nodma "-DMPB_NAR_PLAIN_ORDER -DMPB_NAR_NODMA"
build ord ""; build ord_nodma "-DMPB_NAR_NODMA"
build ord128 "-DMPB_NAR_LUT128"; build ord128_nodma "-DMPB_NAR_LUT128 -DMPB_NAR_NODMA"
for rep in 1 2; do for v in plain plain_nodma ord ord_nodma ord128 ord128_nodma; do for R in ${RS:-2 3}; do
  echo "$v: $(MOIRA_PB_LIB=/tmp/var/$v.so python3 tools/narrow_probe.py $R 10000000 2>&1 | tail -1)"
done; done; done
