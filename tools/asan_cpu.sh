#!/bin/bash
# CPU-only: rebuild the two host libraries with AddressSanitizer + UBSan and run their tests
# (GPU sanitizers are not available on this pool; the HIP library is not touched).
set -e
cd "$(dirname "$0")/.."
cp moira_amd/libmoira_io.so /tmp/io_orig.so; cp moira_amd/libmoira_contig.so /tmp/contig_orig.so
trap 'cp /tmp/io_orig.so moira_amd/libmoira_io.so; cp /tmp/contig_orig.so moira_amd/libmoira_contig.so' EXIT
FL="-O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -fPIC -shared -std=c++17"
g++ $FL -pthread moira_amd/csrc/fastio.cpp moira_amd/csrc/inflate.cpp -o moira_amd/libmoira_io.so
g++ $FL -ffp-contract=off -pthread moira_amd/csrc/contig.cpp -o moira_amd/libmoira_contig.so
LD_PRELOAD=$(g++ -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 \
UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 \
python -m pytest tests/test_fastio.py tests/test_inflate.py tests/test_contig.py tests/test_cli_golden.py -x -q -m "not gpu" -p no:cacheprovider
