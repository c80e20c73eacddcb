#!/bin/bash
# CPU-only AddressSanitizer + UBSan run of the HOST-side code of the C ABI layer (mpb_api.cpp): packers, the Poisson
# tail (threaded), argument validation.  The kernels are not involved (GPU sanitizers are not available on this
# pool): the launch wrappers are stubbed as in tools/tsan_poisson.sh.
set -e
cd "$(dirname "$0")/.."
D=${TMPDIR:-/tmp}/mpb_asan; mkdir -p $D
sed -n '/^cat > \$D\/stubs.cpp/,/^CPP$/p' tools/tsan_poisson.sh | sed '1d;$d' > $D/stubs.cpp
cat > $D/main.cpp <<'CPP'
#include "moira_pb.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
int main()
{
    int bad = 0;
    // packers: every length from 0 to the row size, N / n / Q0, range errors
    for (int len = 0; len <= 48; len++) {
        std::vector<int32_t> q(len ? len : 1);
        std::vector<char> s(len + 1, 'A');
        s[len] = 0;
        for (int i = 0; i < len; i++) { q[i] = (i * 7) % 60; if (i % 11 == 3) s[i] = 'N'; if (i % 13 == 5) s[i] = 'n'; }
        std::vector<uint8_t> row(48, 0xAB);
        bad += mpb_pack_read(s.data(), q.data(), len, row.data(), 48) != MPB_OK;
        for (int i = 0; i < len; i++) {
            const uint8_t want = s[i] == 'N' ? 0 : s[i] == 'n' ? 255 : (q[i] ? q[i] : 1);
            bad += row[i] != want;
        }
        for (int i = len; i < 48; i++) bad += row[i] != 0;
        std::vector<char> qa(len + 1, 0);
        for (int i = 0; i < len; i++) qa[i] = (char)(33 + q[i]);
        bad += mpb_pack_read_ascii(s.data(), qa.data(), len, 33, row.data(), 48) != MPB_OK;
    }
    int32_t neg[3] = {3, -1, 4};
    uint8_t row[16];
    bad += mpb_pack_read("ACG", neg, 3, row, 16) != MPB_E_RANGE;
    int32_t big[1] = {255};
    bad += mpb_pack_read("A", big, 1, row, 16) != MPB_E_RANGE;
    bad += mpb_pack_read("ACGT", neg, 3, row, 2) != MPB_E_INVALID;
    // batch packer with truncation
    const char *seqs = "ACGTNNACGTACGTAC", *quals = "IIII##IIII!!IIII";
    int64_t off[4] = {0, 6, 6, 16};
    std::vector<uint8_t> out(3 * 16);
    int32_t lens[3];
    bad += mpb_pack_batch_ascii(seqs, quals, off, 3, 33, 8, 16, out.data(), lens) != MPB_OK;
    bad += !(lens[0] == 6 && lens[1] == 0 && lens[2] == 8);
    // the Poisson tail on a batch large enough to be split over threads, incl. lambda = 0, huge lambda (NaN), Ns, round
    const int n = 40000;
    std::vector<double> lam(n), ee(n);
    std::vector<int32_t> ns(n), len(n, 300);
    std::vector<uint8_t> ps(n);
    for (int i = 0; i < n; i++) { lam[i] = i < 10 ? 0.0 : (i % 97 == 0 ? 900.0 : 0.002 * i); ns[i] = i % 5 == 0; }
    mpb_filter_params p = {0.005, 0.01, NAN, MPB_AMBIG_TREAT_AS_ERRORS, MPB_FLAG_ROUND};
    bad += mpb_poisson_finish_host(lam.data(), ns.data(), len.data(), 0, n, &p, ee.data(), ps.data()) != MPB_OK;
    bad += !(ee[1] == 0.0 && std::isnan(ee[97]) && ps[97] == 0);
    mpb_filter_params q0 = {1.5, 0.01, NAN, 0, 0};
    bad += mpb_poisson_finish_host(lam.data(), ns.data(), nullptr, 300, n, &q0, ee.data(), ps.data()) != MPB_E_INVALID;
    bad += std::strstr(mpb_last_error(), "Alpha must be between 0 and 1") == nullptr;
    // round 4: the coded batch packer (scores above 254 get spare byte codes), incl. truncation, the full-table refusal, n = 0
    {
        const char *sq = "ACNTnGGGACGTACGT";
        int32_t ql[16] = {30, 300, 30, 0, 7, 5000, 300, 254, 253, 2147483647, 1, 2, 3, 4, 5, 6};
        int64_t off2[4] = {0, 5, 8, 16};
        std::vector<uint8_t> qo(3 * 16, 0xCD);
        int32_t ln[3], codes[256];
        bad += mpb_pack_batch_coded(sq, ql, off2, 3, 0, 16, qo.data(), ln, codes) != MPB_OK;
        bad += !(ln[0] == 5 && ln[1] == 3 && ln[2] == 8 && codes[252] == 300 && codes[251] == 5000 && codes[250] == 2147483647);
        bad += !(qo[0] == 30 && qo[1] == 252 && qo[2] == 0 && qo[3] == 1 && qo[4] == 255 && qo[5] == 0);
        bad += mpb_pack_batch_coded(nullptr, ql, off2, 3, 2, 16, qo.data(), ln, codes) != MPB_OK;
        bad += !(ln[0] == 2 && ln[1] == 2 && ln[2] == 2);
        bad += mpb_pack_batch_coded(sq, ql, off2, 0, 0, 16, qo.data(), ln, codes) != MPB_OK;
        std::vector<int32_t> full(255);
        for (int i = 0; i < 254; i++) full[i] = i + 1;
        full[254] = 300;
        int64_t off3[2] = {0, 255};
        std::vector<uint8_t> wide(256);
        bad += mpb_pack_batch_coded(nullptr, full.data(), off3, 1, 0, 256, wide.data(), ln, codes) != MPB_E_RANGE;
        int32_t negq[2] = {3, -1};
        int64_t off4[2] = {0, 2};
        bad += mpb_pack_batch_coded(nullptr, negq, off4, 1, 0, 16, qo.data(), ln, codes) != MPB_E_RANGE;
        bad += mpb_pack_batch_coded(nullptr, ql, off2, 3, 0, 8, qo.data(), ln, codes) != MPB_E_INVALID;      // stride not a multiple of 16
    }
    std::printf("asan_api: %d failed checks, version %s\n", bad, mpb_version());
    return bad != 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-omit-frame-pointer -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ \
    -I/opt/rocm/include -Iinclude -Imoira_amd/csrc moira_amd/csrc/mpb_api.cpp $D/stubs.cpp $D/main.cpp \
    -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o $D/asan_api
ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 $D/asan_api
