// What the broker itself can serve, without Python in the callers: P client THREADS of one process (each attaches and
// claims a slot of its own) call mpb_broker_call in a loop with 300-base reads.  Build and run ON THE GPU BOX:
//   g++ -O2 -std=c++17 -pthread -Iinclude tools/experiments/broker_c_clients.cpp moira_amd/libmoira_pb.so -Wl,-rpath,$PWD/moira_amd -o /tmp/bcc
//   python -m moira_amd.broker --name cc --idle-exit 30 &   ;   /tmp/bcc cc 1 2 4 8 16 32
#include "moira_pb.h"
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>
int main(int argc, char **argv)
{
    const char *name = argv[1];
    for (int a = 2; a < argc; a++) {
        const int P = atoi(argv[a]), N = 20000;
        std::atomic<int> ready{0}, go{0}, bad{0};
        std::vector<double> secs(P);
        std::vector<std::thread> th;
        for (int k = 0; k < P; k++)
            th.emplace_back([&, k] {
                mpb_broker_client *cl = nullptr;
                if (mpb_broker_attach(name, 60000, &cl) != MPB_OK) { bad++; ready++; return; }
                char seq[301]; int32_t q[300];
                for (int i = 0; i < 300; i++) { seq[i] = "ACGT"[(i + k) & 3]; q[i] = 38 - (i * i * 20) / 90000 - ((i * 7 + k) % 6); }
                seq[300] = 0;
                double ee; int32_t ns;
                for (int i = 0; i < 50; i++) bad += mpb_broker_call(cl, seq, q, 300, 0.005, &ee, &ns) != MPB_OK;
                ready++;
                while (!go.load()) std::this_thread::yield();
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < N; i++) bad += mpb_broker_call(cl, seq, q, 300, 0.005, &ee, &ns) != MPB_OK;
                secs[k] = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
                mpb_broker_detach(cl);
            });
        while (ready.load() < P) std::this_thread::yield();
        go = 1;
        for (auto &t : th) t.join();
        double mx = 0;
        for (double s : secs) mx = s > mx ? s : mx;
        int64_t served = 0, batches = 0, solo = 0;
        mpb_broker_stats(name, &served, &batches, &solo, nullptr, nullptr);
        printf("%2d C client threads: %.1f us per call in a thread, %.3e calls/s in all (errors %d; broker totals: %lld reads, %lld launches)\n",
               P, mx / N * 1e6, P * (double)N / mx, bad.load(), (long long)served, (long long)batches);
        fflush(stdout);
    }
    return 0;
}
