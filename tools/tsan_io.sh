#!/bin/bash
# CPU-only ThreadSanitizer run of the threaded parts of libmoira_io (round 3): the parallel FASTQ indexer
# (mio_fastq_index_mt), the concurrent preads (mio_pread_mt) and the sharded collapse (mio_collapse_add with 8 threads,
# several chunks, export) and the BGZF members inflated on 8 threads (mio_bgzf_inflate_mt).  Pure C++ harness, no Python in the process.
set -e
cd "$(dirname "$0")/.."
D=${TMPDIR:-/tmp}/mio_tsan; mkdir -p $D
cat > $D/main.cpp <<'CPP'
#include "moira_io.h"
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <fcntl.h>
#include <unistd.h>
int main()
{
    // 60,000 records, heavy duplication of sequences, quality lines that start with '@' now and then
    std::string buf;
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
    std::vector<std::string> pool;
    for (int k = 0; k < 3000; k++) { std::string q; int L = 30 + rnd() % 80; for (int i = 0; i < L; i++) q += "ACGT"[rnd() % 4]; pool.push_back(q); }
    const int n = 60000;
    for (int i = 0; i < n; i++) {
        const std::string &sq = pool[(rnd() % 3000) * (rnd() % 3000) / 3000];
        std::string ql(sq.size(), 'I');
        if (i % 11 == 0) ql[0] = '@';
        buf += "@r:" + std::to_string(i) + "\n" + sq + "\n+\n" + ql + "\n";
    }
    std::vector<int64_t> a((size_t)(n + 1) * MIO_IDX_COLS), b((size_t)(n + 1) * MIO_IDX_COLS);
    int64_t ca = 0, cb = 0; int32_t ba = 0, bb = 0;
    const int64_t na = mio_fastq_index(buf.data(), (int64_t)buf.size(), 1, n, a.data(), &ca, &ba);
    const int64_t nb = mio_fastq_index_mt(buf.data(), (int64_t)buf.size(), 1, n, b.data(), &cb, &bb, 8);
    if (na != n || nb != n || ca != cb || ba != bb || memcmp(a.data(), b.data(), (size_t)n * MIO_IDX_COLS * 8)) { printf("index mismatch\n"); return 1; }
    // concurrent preads
    char path[] = "/tmp/mio_tsan_XXXXXX";
    int fd = mkstemp(path);
    if (write(fd, buf.data(), buf.size()) != (ssize_t)buf.size()) return 2;
    std::vector<char> back(buf.size() + 4096);
    const int64_t got = mio_pread_mt(fd, 0, back.data(), (int64_t)back.size(), 8);
    close(fd); unlink(path);
    if (got != (int64_t)buf.size() || memcmp(back.data(), buf.data(), buf.size())) { printf("pread mismatch\n"); return 3; }
    // sharded collapse: 8 threads against 1 thread, three chunks each
    std::vector<double> ee((size_t)n);
    for (int i = 0; i < n; i++) ee[(size_t)i] = (rnd() % 30) / 10.0;
    std::vector<std::vector<char>> outs;
    for (int threads : {1, 8}) {
        mio_collapse *c = mio_collapse_create();
        mio_collapse_set_threads(c, threads);
        for (int lo = 0; lo < n; lo += 20000)
            if (mio_collapse_add(c, buf.data(), a.data() + (size_t)lo * MIO_IDX_COLS, 20000, 0, ee.data() + lo, nullptr, nullptr)) return 4;
        const int64_t g = mio_collapse_count(c);
        std::vector<double> gee((size_t)g); std::vector<int64_t> glen((size_t)g), gsize((size_t)g), sel((size_t)g);
        if (mio_collapse_export(c, gee.data(), glen.data(), gsize.data(), nullptr, nullptr)) return 5;
        for (int64_t k = 0; k < g; k++) sel[(size_t)k] = k;
        std::vector<char> out(64 << 20);
        int64_t need = 0;
        const int64_t w = mio_collapse_format(c, sel.data(), g, MIO_FMT_NAMES, 33, 33, 1, nullptr, 0, nullptr, nullptr, nullptr, out.data(), (int64_t)out.size(), &need);
        if (w < 0) return 6;
        out.resize((size_t)w);
        outs.push_back(out);
        mio_collapse_destroy(c);
    }
    if (outs[0] != outs[1]) { printf("collapse mismatch\n"); return 7; }
    // BGZF members (stored deflate blocks built by hand) inflated on 8 threads
    {
        std::string z;
        const size_t piece = 60000;
        for (size_t pos = 0; pos < buf.size(); pos += piece) {
            const size_t len = std::min(piece, buf.size() - pos);
            const unsigned size = 18 + 5 + (unsigned)len + 8;
            const unsigned char head[18] = {0x1f, 0x8b, 8, 4, 0, 0, 0, 0, 0, 0xff, 6, 0, 'B', 'C', 2, 0, (unsigned char)((size - 1) & 0xff), (unsigned char)((size - 1) >> 8)};
            z.append((const char *)head, 18);
            const unsigned char sb[5] = {1, (unsigned char)(len & 0xff), (unsigned char)(len >> 8), (unsigned char)(~len & 0xff), (unsigned char)((~len >> 8) & 0xff)};
            z.append((const char *)sb, 5);
            z.append(buf.data() + pos, len);
            const uint32_t crc = mio_crc32(0, (const uint8_t *)buf.data() + pos, (int64_t)len);
            for (uint32_t v : {crc, (uint32_t)len}) for (int k = 0; k < 4; k++) z += (char)((v >> (8 * k)) & 0xff);
        }
        const size_t zl = z.size();
        z.append(8, '\0');                                              // readable slack
        std::vector<int64_t> offs(4096), oo(4097); std::vector<int32_t> sizes(4096);
        int32_t why = 0;
        const int64_t nbk = mio_bgzf_scan((const uint8_t *)z.data(), (int64_t)zl, 4096, (int64_t)1 << 40, offs.data(), sizes.data(), oo.data(), &why);
        if (nbk <= 0 || oo[(size_t)nbk] != (int64_t)buf.size()) { printf("bgzf scan mismatch\n"); return 8; }
        std::vector<uint8_t> text(buf.size());
        if (mio_bgzf_inflate_mt((const uint8_t *)z.data(), offs.data(), sizes.data(), oo.data(), nbk, text.data(), 8)) { printf("bgzf: %s\n", mio_inflate_error()); return 9; }
        if (memcmp(text.data(), buf.data(), buf.size())) { printf("bgzf text mismatch\n"); return 10; }
        printf("tsan harness: %lld BGZF members inflated on 8 threads\n", (long long)nbk);
    }
    printf("tsan harness: index_mt == index (%d records), pread_mt ok, collapse(8 threads) == collapse(1 thread): %zu bytes of names\n", n, outs[0].size());
    return 0;
}
CPP
g++ -O1 -g -fsanitize=thread -std=c++17 -pthread -Iinclude moira_amd/csrc/fastio.cpp moira_amd/csrc/inflate.cpp $D/main.cpp -o $D/tsan_io
TSAN_OPTIONS=halt_on_error=1 $D/tsan_io
echo "ThreadSanitizer: no data race reported"
