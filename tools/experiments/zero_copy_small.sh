#!/bin/bash
# zero-copy threshold experiment for the small path + broker copies vs zero-copy
cd "$(dirname "$0")/../.."
FL="-O3 --offload-arch=gfx950 -fPIC -shared -std=c++17 -ffp-contract=off -fno-fast-math -pthread -Iinclude -x hip -DMPB_TUNING_KNOBS"
mkdir -p /tmp/var
/opt/rocm/bin/hipcc $FL moira_amd/csrc/mpb_kernels.hip moira_amd/csrc/mpb_api.cpp moira_amd/csrc/mpb_broker.cpp -o /tmp/var/knobs.so || exit 1
export MOIRA_PB_LIB=/tmp/var/knobs.so
for zc in 0 4096 65536 1048576; do
  echo "== MPB_SMALL_ZC_BYTES=$zc"; MPB_SMALL_ZC_BYTES=$zc timeout -k 10 120 python tools/per_read_latency.py 2>&1 | head -6
done
unset MOIRA_PB_LIB
echo "== broker, zero-copy lanes (default)"; timeout -k 10 200 python tools/per_read_concurrency.py 1 8 16
echo "== broker, copies (MPB_BROKER_COPIES=1)"; MPB_BROKER_COPIES=1 timeout -k 10 200 python tools/per_read_concurrency.py 1 8 16
