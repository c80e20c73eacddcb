#!/bin/bash
# Run ON THE GPU BOX: where a k_narrow wave's cycles go (in-kernel stamps; experiment build, never the shipped library)
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip"
mkdir -p /tmp/var
/opt/rocm/bin/hipcc $FL -DMPB_NAR_STAMPS $EXTRA moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/stamps.so 2>/tmp/var/stamps.err || { tail -5 /tmp/var/stamps.err; exit 1; }
MOIRA_PB_LIB=/tmp/var/stamps.so MPB_NAR_STAMPS_PRINT=1 python3 tools/narrow_probe.py ${1:-2} 10000000 2>&1 | tail -4
