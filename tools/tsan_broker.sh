#!/bin/bash
# CPU-only sanitizer runs of the broker's protocol (moira_amd/csrc/mpb_broker.cpp compiled UNCHANGED against
# tests/helpers/broker_stub.cpp: HIP calls on host memory, the oracle for the arithmetic).  One process: thread 0 serves,
# 12 client threads attach (each claims a slot of its own) and make 4,000 calls each with mixed lengths / alphas, a
# shutdown arrives while two more clients are still calling.  ThreadSanitizer sees every hand-over of a slot's plain
# fields (length, alpha, row bytes, results) across the state word, AddressSanitizer + UBSan every access of both sides.
# GPU sanitizers are not available on this pool; the kernels are not involved here.
set -e
cd "$(dirname "$0")/.."
D=${TMPDIR:-/tmp}/mpb_broker_san; mkdir -p $D
make -C oracle -s >/dev/null 2>&1 || true
cat > $D/main.cpp <<'CPP'
#include "moira_pb.h"
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>
#include <unistd.h>
#include <vector>
extern "C" {
typedef struct pbo_params { double alpha, uncert, maxerrors; int32_t ambig_mode; uint32_t flags; } pbo_params;
int pbo_filter_batch(const uint8_t *, int64_t, int64_t, const int32_t *, int32_t, const pbo_params *, int, int, double *, int32_t *, uint8_t *, int32_t *);
int pbo_pack_read(const char *, const int32_t *, int32_t, uint8_t *, int32_t);
}
static std::atomic<int> bad{0}, done{0};
static void client(const char *name, int id, int calls, bool expect_cut)
{
    mpb_broker_client *cl = nullptr;
    if (mpb_broker_attach(name, 20000, &cl) != MPB_OK) { bad++; return; }
    unsigned s = 12345u * (id + 1);
    auto rnd = [&] { s = s * 1664525u + 1013904223u; return s >> 8; };
    for (int k = 0; k < calls; k++) {
        const int len = rnd() % 330;
        std::vector<int32_t> q(len ? len : 1);
        std::vector<char> seq(len + 1, 'A');
        seq[len] = 0;
        for (int i = 0; i < len; i++) { q[i] = 2 + rnd() % 40; if (rnd() % 61 == 0) seq[i] = 'N'; }
        const double alpha = (rnd() & 3) ? 0.005 : 0.05;
        double ee = -1; int32_t ns = -1;
        const int rc = mpb_broker_call(cl, seq.data(), q.data(), len, alpha, &ee, &ns);
        if (rc == MPB_E_HIP && expect_cut) break;            // the shutdown arrived
        if (rc != MPB_OK) { bad++; continue; }
        std::vector<uint8_t> row(336, 0);
        pbo_pack_read(seq.data(), q.data(), len, row.data(), 336);
        pbo_params p{alpha, 1.0, NAN, 1, 0};
        double we; int32_t wn; uint8_t wp;
        pbo_filter_batch(row.data(), 1, 336, &len, 0, &p, 0, 1, &we, &wn, &wp, nullptr);
        if (!(we == ee && wn == ns)) bad++;
    }
    mpb_broker_detach(cl);
    done++;
}
int main()
{
    char name[64];
    snprintf(name, sizeof(name), "san_%d", (int)getpid());
    std::thread srv([&] { if (mpb_broker_serve((mpb_ctx *)1, name, 16, 0) != MPB_OK) bad++; });
    std::vector<std::thread> th;
    for (int i = 0; i < 12; i++) th.emplace_back(client, name, i, 4000, false);
    for (auto &t : th) t.join();
    std::thread late1(client, name, 100, 1 << 30, true), late2(client, name, 101, 1 << 30, true);
    usleep(200000);
    int64_t served = 0, batches = 0, solo = 0; int32_t pid = 0, att = 0;
    if (mpb_broker_stats(name, &served, &batches, &solo, &pid, &att) != MPB_OK || served < 48000 || att != 2) bad++;
    mpb_broker_shutdown(name);
    srv.join(); late1.join(); late2.join();
    std::printf("broker sanitizer run: %d failed checks; %lld reads in %lld launches + %lld alone; %d clients finished\n",
                bad.load(), (long long)served, (long long)batches, (long long)solo, done.load());
    return bad.load() != 0;
}
CPP
SRC="moira_amd/csrc/mpb_broker.cpp tests/helpers/broker_stub.cpp $D/main.cpp"
INC="-D__HIP_PLATFORM_AMD__ -I/opt/rocm/include -Iinclude"
gcc -O1 -g -fPIC -c oracle/pb_oracle.c -o $D/oracle.o -lm
for san in thread address,undefined; do
  g++ -O1 -g -std=c++17 -pthread -fsanitize=$san -fno-omit-frame-pointer $INC $SRC $D/oracle.o -lm -o $D/run_${san%%,*}
  echo "== -fsanitize=$san, resident server serving the slots themselves (the stub's host thread keeps k_serve's side of the mailbox)"
  TSAN_OPTIONS="halt_on_error=0" ASAN_OPTIONS="detect_leaks=1" $D/run_${san%%,*}
  echo "== -fsanitize=$san, resident server behind the broker's copies (MPB_BROKER_DIRECT=0)"
  MPB_BROKER_DIRECT=0 TSAN_OPTIONS="halt_on_error=0" ASAN_OPTIONS="detect_leaks=1" $D/run_${san%%,*}
  for rt in 0 1; do                # the launch-per-micro-batch lanes: the serving loop alone, and with its optional retire thread
    echo "== -fsanitize=$san, lanes (MPB_BROKER_SERVER=0), retire thread $rt"
    MPB_BROKER_SERVER=0 MPB_BROKER_RETIRE_THREAD=$rt TSAN_OPTIONS="halt_on_error=0" ASAN_OPTIONS="detect_leaks=1" $D/run_${san%%,*}
  done
done
