#!/bin/bash
# Run ON THE GPU BOX: where k_narrow's arithmetic-alone time goes: without the table reads, the panel reads, the epilogue
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
build() { /opt/rocm/bin/hipcc $FL -DMPB_NAR_NODMA $2 moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/$1.so 2>/tmp/var/$1.err || { tail -5 /tmp/var/$1.err; exit 1; }; }
build base ""; build nolut "-DMPB_NAR_X_NOLUT"; build notile "-DMPB_NAR_X_NOTILE"; build noepi "-DMPB_NAR_X_NOEPI"; build none "-DMPB_NAR_X_NOLUT -DMPB_NAR_X_NOTILE -DMPB_NAR_X_NOEPI"
for rep in 1 2; do for v in base nolut notile noepi none; do
  echo "$v: $(MOIRA_PB_LIB=/tmp/var/$v.so python3 tools/narrow_probe.py ${R:-2} 10000000 2>&1 | tail -1)"
done; done
