// Host-side micro-benchmark of the Needleman-Wunsch aligner alone (csrc/contig.cpp): us per 2 x L alignment on one core.
//   g++ -O3 -std=c++17 -ffp-contract=off -pthread -Iinclude moira_amd/csrc/contig.cpp tools/experiments/nw_bench.cpp -o /tmp/nw_bench && /tmp/nw_bench 250
#include "moira_contig.h"
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
int main(int argc, char **argv)
{
    const int L = argc > 1 ? atoi(argv[1]) : 250, N = 20000;
    unsigned s = 1;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<std::string> a(N), b(N);
    for (int k = 0; k < N; k++) {
        std::string frag; const int F = L + 60 + rnd() % 100;       // fragment: overlap L*2 - F
        for (int i = 0; i < F; i++) frag += "ACGT"[rnd() % 4];
        a[k] = frag.substr(0, L);
        std::string tail = frag.substr(F - L);
        for (auto &c : tail) if (rnd() % 50 == 0) c = "ACGT"[rnd() % 4];
        b[k] = tail;       // (already in forward orientation: the aligner sees seq1 vs revcomp'd mate)
    }
    std::vector<char> o1(2 * L + 8), o2(2 * L + 8);
    int32_t alen, score; long long tot = 0;
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < N; k++) { mct_nw_align(a[k].data(), L, b[k].data(), L, 1, -1, -2, o1.data(), o2.data(), &alen, &score); tot += score; }
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("L=%d: %.2f us per alignment (checksum %lld)\n", L, dt / N * 1e6, tot);
    return 0;
}
