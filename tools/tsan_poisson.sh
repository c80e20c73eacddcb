#!/bin/bash
# CPU-only ThreadSanitizer run of the Poisson host tail (mpb_poisson_finish_host): two threads reach
# factorial_table() for the first time concurrently, as two contexts on two threads can.  The HIP kernels are
# not involved (GPU sanitizers are not available on this pool): mpb_api.cpp is built with g++ against the HIP
# headers, the launch wrappers are stubbed, and only the host function is called.
set -e
cd "$(dirname "$0")/.."
D=${TMPDIR:-/tmp}/mpb_tsan; mkdir -p $D
cat > $D/stubs.cpp <<'CPP'
#include "mpb_internal.h"
#include <cstdlib>
#define STUB { abort(); }
void mpb_launch_prepass(const uint8_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, int32_t *, double *, uint8_t *, hipStream_t, const int32_t *) STUB
void mpb_launch_small(const uint8_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, int32_t *, double *, uint8_t *, hipStream_t, const MpbSmallHost *) STUB
void mpb_launch_scan(int64_t, const int32_t *, const MpbWorkspace &, hipStream_t) STUB
void mpb_launch_scatter(int64_t, const int32_t *, const int32_t *, const MpbDevParams &, const MpbWorkspace &, hipStream_t, const int32_t *) STUB
void mpb_launch_dp(const uint8_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, const int32_t *, double *, uint8_t *, hipStream_t) STUB
void mpb_launch_overflow(const uint8_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, const int32_t *, double *, uint8_t *, hipStream_t) STUB
void mpb_launch_lambda(const uint8_t *, int64_t, int64_t, const int32_t *, int32_t, const double2 *, double *, int32_t *, int32_t *, hipStream_t) STUB
void mpb_launch_decode(const uint8_t *, const uint8_t *, int64_t, int64_t, const int32_t *, int32_t, int32_t, uint8_t *, int32_t *, hipStream_t) STUB
void mpb_launch_wide(const uint8_t *, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, const int32_t *, double *, uint8_t *, hipStream_t) STUB
void mpb_launch_decode_classify(const uint8_t *, const uint8_t *, int32_t, uint8_t *, int32_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, int32_t *, double *, uint8_t *, hipStream_t) STUB
void mpb_launch_encode(const uint8_t *, int64_t, int64_t, int32_t, uint8_t *, uint8_t *, hipStream_t) STUB
void mpb_launch_count(const uint8_t *, int64_t, const MpbWorkspace &, hipStream_t) STUB
void mpb_launch_synth(uint8_t *, int64_t, int64_t, int32_t, int32_t, int32_t, int32_t *, uint64_t, int64_t, hipStream_t, int) STUB
void mpb_launch_narrow(int, const uint8_t *, int64_t, int64_t, int32_t, const MpbDevParams &, const MpbWorkspace &, double *, int32_t *, uint8_t *, int32_t *, int, hipStream_t) STUB
void mpb_launch_sample(const uint8_t *, int64_t, int64_t, int32_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, int, hipStream_t) STUB
void mpb_launch_narrow_ragged(int, int, const uint8_t *, int64_t, int64_t, const int32_t *, const MpbDevParams &, const MpbWorkspace &, double *, int32_t *, uint8_t *, int32_t *, int, hipStream_t) STUB
void mpb_launch_serve(const MpbServeBox &, const double2 *, uint32_t, uint32_t, hipStream_t) STUB
int mpb_narrow_lds_bytes() { return 1 << 15; }
int mpb_narrow_rs_lds_bytes() { return 1 << 15; }
int mpb_narrow_rs_reads_per_lane(int64_t, int) { return 0; }
CPP
cat > $D/main.cpp <<'CPP'
#include "moira_pb.h"
#include <cmath>
#include <cstdio>
#include <thread>
#include <vector>
int main()
{
    const int n = 2000;
    std::vector<double> lam(n), ee1(n), ee2(n);
    std::vector<int32_t> ns(n, 0);
    std::vector<uint8_t> p1(n), p2(n);
    for (int i = 0; i < n; i++) lam[i] = 0.05 * i;
    mpb_filter_params prm = {0.005, 0.01, NAN, MPB_AMBIG_IGNORE, 0};
    int rc1 = -1, rc2 = -1;
    std::thread a([&] { rc1 = mpb_poisson_finish_host(lam.data(), ns.data(), nullptr, 300, n, &prm, ee1.data(), p1.data()); });
    std::thread b([&] { rc2 = mpb_poisson_finish_host(lam.data(), ns.data(), nullptr, 300, n, &prm, ee2.data(), p2.data()); });
    a.join(); b.join();
    int same = 1;
    for (int i = 0; i < n; i++) same &= (ee1[i] == ee2[i]) || (std::isnan(ee1[i]) && std::isnan(ee2[i]));
    std::printf("tsan_poisson: rc %d %d, results identical: %d, ee[100] = %.17g\n", rc1, rc2, same, ee1[100]);
    return (rc1 || rc2 || !same) ? 1 : 0;
}
CPP
g++ -std=c++17 -O1 -g -fsanitize=thread -ffp-contract=off -pthread -D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude -Imoira_amd/csrc \
    moira_amd/csrc/mpb_api.cpp $D/stubs.cpp $D/main.cpp -L/opt/rocm/lib -lamdhip64 -Wl,-rpath,/opt/rocm/lib -o $D/tsan_poisson
TSAN_OPTIONS="halt_on_error=1" $D/tsan_poisson
