// libmoira_io.so -- byte-level FASTQ indexing, packing and record formatting for the CLI
// (C ABI in include/moira_io.h).  Host only; built with g++.
//
// Behaviour follows the reference's text handling (moira/moira.py:1152-1204 parse_fastq,
// :842-970 write_results) as restated in moira_amd/cli.py; tests/test_fastio.py compares the two
// paths byte for byte.
#include "../../include/moira_io.h"

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <memory>
#include <new>
#include <thread>
#include <vector>

#include <unistd.h>

namespace {

thread_local char g_err[256] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

// what str.strip() removes from an ASCII string
inline bool is_space(unsigned char c)
{
    return c == ' ' || (c >= 0x09 && c <= 0x0d) || (c >= 0x1c && c <= 0x1f);
}

struct Span { int64_t off, len; };

inline Span strip(const char *buf, int64_t a, int64_t b)       // [a, b) -> stripped
{
    while (a < b && is_space((unsigned char)buf[a])) a++;
    while (b > a && is_space((unsigned char)buf[b - 1])) b--;
    return {a, b - a};
}

// decimal text of 1..254 (the qualities a .qual line can show), built once
struct QualText {
    char txt[256][4];
    unsigned char len[256];
    QualText()
    {
        for (int q = 0; q < 256; q++) len[q] = (unsigned char)snprintf(txt[q], 4, "%d", q > 999 ? 999 : q);
    }
};
const QualText g_qt;

struct Out {
    char *p; int64_t cap, n;
    inline void put(const char *s, int64_t k)
    {
        if (n + k <= cap) memcpy(p + n, s, (size_t)k);
        n += k;
    }
    inline void ch(char c)
    {
        if (n < cap) p[n] = c;
        n++;
    }
};

}  // namespace

// ---- collapse of identical sequences (ref: moira/moira.py:459-475, :490-493) --------------------

// CPython 2.7's dict as far as iteration order goes (PyDict_MINSIZE 8; insertdict_clean into the first empty slot of the
// probe sequence; dictresize to 4 x used, 2 x above 50000 used, when the table is 2/3 full; no deletions)
struct Py2DictOrder {
    struct Slot { uint64_t hash; int64_t id; };              // id < 0: empty
    std::vector<Slot> slots;
    uint64_t mask = 7, used = 0;
    Py2DictOrder() : slots(8, Slot{0, -1}) {}
    void place(uint64_t h, int64_t id)
    {
        uint64_t i = h & mask, perturb = h;
        for (;;) {
            Slot &sl = slots[i & mask];
            if (sl.id < 0) { sl.hash = h; sl.id = id; return; }
            i = (i << 2) + i + perturb + 1;
            perturb >>= 5;
        }
    }
    void insert_new(uint64_t h, int64_t id)
    {
        place(h, id);
        used++;
        if (used * 3 < (mask + 1) * 2) return;
        const uint64_t minused = (used > 50000 ? 2 : 4) * used;      // dictresize(): 4 x used, 2 x above 50000
        uint64_t newsize = 8;
        while (newsize <= minused) newsize <<= 1;
        std::vector<Slot> old;
        old.swap(slots);
        slots.assign(newsize, Slot{0, -1});
        mask = newsize - 1;
        // re-inserted in old slot order; the target line of the entry 16 slots ahead is requested now (a table of
        // millions of keys is one cache miss per placement otherwise -- and a run re-places every key about 4 times)
        const size_t no = old.size();
        for (size_t k = 0; k < no; k++) {
            if (k + 16 < no && old[k + 16].id >= 0) __builtin_prefetch(&slots[old[k + 16].hash & mask], 1);
            if (old[k].id >= 0) place(old[k].hash, old[k].id);
        }
    }
    void prefetch(uint64_t h) const { __builtin_prefetch(&slots[h & mask], 1); }
};

// The reference's `uniques` is ONE Python-2 dict, and its iteration order (the slot order of CPython 2.7's open
// addressing, which depends on the hashes, the insertion order of the distinct keys and the resize history) decides the
// output order of groups of equal abundance.  Only that ORDER is sequential; the work per read -- hash the sequence, find
// its group, compare, copy a new sequence, link the header -- is not.  So the groups live in 64 shards picked by hash,
// each a plain table with its own storage that one thread fills (reads in file order, so "first seen wins ties" and the
// order of names_info are kept shard by shard), and every group remembers the global number of the read that created
// it; after the shards of a chunk are filled, the chunk's NEW keys are replayed in that order into a simulated
// CPython-2.7 dict (insertdict / dictresize: ~0.1 us per new key -- the one sequential step, and it runs on the writer
// thread while the next chunks are read, packed and filtered); mio_collapse_export reads the slot order off it.
struct mio_collapse {
    struct Uniq {
        int64_t seq_off, qual_off;       // arena offsets; the rep's qualities are re-pointed on a better rep
        int32_t len;
        uint8_t flags;
        double ee;
        int32_t aux[3];                  // overlap_length, gaps, mismatches of the representative
        int64_t size;                    // len(names_info)
        int64_t front, back_head, back_tail;   // names: front-inserted (newest first), then first seen + appended
        uint64_t hash;                   // CPython-2.7 hash(str) of the sequence
        int64_t first_seen;              // global number of the read that created the group
    };
    struct Name { int64_t off; int32_t len; int64_t next; };
    struct Shard {
        // append-only byte store in 1 MiB blocks (no reallocation copies; an item never straddles blocks)
        static constexpr int ARENA_SHIFT = 20;
        std::vector<std::unique_ptr<char[]>> blocks;
        int64_t block_used = 0;
        std::vector<Uniq> uniq;
        std::vector<Name> names;
        std::vector<int64_t> tab;        // open addressing, linear probing: uid or -1 (order irrelevant here)
        uint64_t tmask = 0;

        char *reserve(int64_t n, int64_t *off)
        {
            const int64_t cap = (int64_t)1 << ARENA_SHIFT;
            if (n > cap) {                                    // a header or sequence line longer than a block: a block of its own
                blocks.emplace_back(new char[(size_t)n]);     // (this used to end as "out of memory while collapsing")
                *off = (int64_t)(blocks.size() - 1) << ARENA_SHIFT;
                block_used = cap;                             // the next item opens a fresh block
                return blocks.back().get();
            }
            if (blocks.empty() || block_used + n > cap) {
                blocks.emplace_back(new char[(size_t)cap]);
                block_used = 0;
            }
            *off = ((int64_t)(blocks.size() - 1) << ARENA_SHIFT) | block_used;
            char *p = blocks.back().get() + block_used;
            block_used += n;
            return p;
        }
        int64_t put(const char *p, int64_t n)
        {
            int64_t off;
            memcpy(reserve(n, &off), p, (size_t)n);
            return off;
        }
        const char *at(int64_t off) const { return blocks[(size_t)(off >> ARENA_SHIFT)].get() + (off & (((int64_t)1 << ARENA_SHIFT) - 1)); }
        void grow()
        {
            const uint64_t size = tab.empty() ? 1024 : (tmask + 1) * 2;
            std::vector<int64_t> nt(size, -1);
            for (int64_t uid = 0; uid < (int64_t)uniq.size(); uid++) {
                uint64_t i = (uniq[(size_t)uid].hash * 0x9E3779B97F4A7C15ull >> 20) & (size - 1);
                while (nt[i] >= 0) i = (i + 1) & (size - 1);
                nt[i] = uid;
            }
            tab.swap(nt);
            tmask = size - 1;
        }
        // uid of the group whose sequence is key[0, len), or -1 with *slot = where it would go
        int64_t find(uint64_t h, const char *key, int64_t len, uint64_t *slot)
        {
            if (tab.empty() || (uniq.size() + 1) * 2 > tmask + 1) grow();
            uint64_t i = (h * 0x9E3779B97F4A7C15ull >> 20) & tmask;
            for (;; i = (i + 1) & tmask) {
                const int64_t uid = tab[i];
                if (uid < 0) { *slot = i; return -1; }
                const Uniq &u = uniq[(size_t)uid];
                if (u.hash == h && u.len == len && memcmp(at(u.seq_off), key, (size_t)len) == 0) return uid;
            }
        }
    };
    static constexpr int NSHARD = 64;
    static int shard_of(uint64_t h) { return (int)((h * 0x9E3779B97F4A7C15ull) >> 58); }
    Shard shard[NSHARD];
    int threads = 1;
    int64_t reads_seen = 0;
    Py2DictOrder dict;                   // ids = positions in `created`
    struct Ref { int64_t local; uint8_t sid; };
    std::vector<Ref> created;            // the groups in the order the reference's dict met their keys
    // output order, built by mio_collapse_export: group k is shard[g_shard[k]].uniq[g_local[k]]
    std::vector<uint8_t> g_shard;
    std::vector<int64_t> g_local;
    bool exported = false;
    int64_t count() const { int64_t n = 0; for (const auto &sh : shard) n += (int64_t)sh.uniq.size(); return n; }
};

namespace {

uint64_t py2_hash(const unsigned char *s, int64_t L)
{
    if (L <= 0) return 0;
    uint64_t x = (uint64_t)s[0] << 7;
    for (int64_t i = 0; i < L; i++) x = (1000003ull * x) ^ s[i];
    x ^= (uint64_t)L;
    if (x == ~0ull) x = ~0ull - 1;
    return x;
}

void put_quals(Out &o, const unsigned char *ql, int64_t L, int32_t fastq_offset, int floor_q)
{
    if (o.n + 4 * L <= o.cap) {                               // ' '.join(map(str, quals)); fast when it fits
        char *w = o.p + o.n;
        for (int64_t i = 0; i < L; i++) {
            int q = (int)ql[i] - fastq_offset;
            q = q <= floor_q ? floor_q : q > 255 ? 255 : q;   // (packing has already rejected q < 0 / > 254)
            memcpy(w, g_qt.txt[q], 4);
            w += g_qt.len[q];
            *w++ = ' ';
        }
        if (L > 0) w--;                                       // no trailing separator
        o.n = w - o.p;
    } else {
        for (int64_t i = 0; i < L; i++) {
            int q = (int)ql[i] - fastq_offset;
            q = q <= floor_q ? floor_q : q > 255 ? 255 : q;
            o.n += g_qt.len[q] + (i + 1 < L ? 1 : 0);
        }
    }
}

void put_qual_string(Out &o, const unsigned char *ql, int64_t L, int32_t fastq_offset, int32_t out_offset, int floor_q)
{
    // chr(q + offset) of the clamped qualities; with equal offsets that is the input string with Q0
    // shown as Q1
    if (o.n + L <= o.cap) {
        char *w = o.p + o.n;
        if (out_offset == fastq_offset) {
            memcpy(w, ql, (size_t)L);
            if (floor_q > 0)
            for (char *z = (char *)memchr(w, fastq_offset, (size_t)L); z; z = (char *)memchr(z, fastq_offset, (size_t)(w + L - z)))
                *z++ = (char)(fastq_offset + 1);
        } else {
            for (int64_t i = 0; i < L; i++) {
                int q = (int)ql[i] - fastq_offset;
                q = q <= floor_q ? floor_q : q;
                w[i] = (char)(q + out_offset);
            }
        }
    }
    o.n += L;
}

}  // namespace

extern "C" {

const char *mio_version(void) { return "moira_io 0.3"; }
const char *mio_last_error(void) { return g_err; }

int64_t mio_fastq_index(const char *buf, int64_t len, int32_t final, int64_t max_records,
                        int64_t *idx, int64_t *consumed, int32_t *bad_kind)
{
    if (!buf || len < 0 || max_records < 0 || !idx || !consumed || !bad_kind)
        return fail(MIO_E_INVALID, "mio_fastq_index: bad arguments");
    *consumed = 0;
    *bad_kind = MIO_REC_OK;
    int64_t n = 0, pos = 0;
    Span line[4];
    while (n < max_records) {
        int64_t p = pos;
        int k = 0;
        for (; k < 4; k++) {
            if (p >= len) break;
            const char *nl = (const char *)memchr(buf + p, '\n', (size_t)(len - p));
            int64_t e;
            if (nl) e = nl - buf;
            else if (final) e = len;
            else break;
            // a "\r" that is not the one before "\n" would be a line break of its own
            const char *cr = (const char *)memchr(buf + p, '\r', (size_t)(e - p));
            if (cr && !(nl && cr == nl - 1)) return fail(MIO_E_UNSUPPORTED, "lone carriage return");
            unsigned char any = 0;
            for (int64_t t = p; t < e; t++) any |= (unsigned char)buf[t];
            if (any & 0x80) return fail(MIO_E_UNSUPPORTED, "non-ASCII byte");
            line[k] = strip(buf, p, e);
            p = nl ? e + 1 : e;
        }
        if (k < 4) break;                                   // incomplete record: wait for more input
        // header: replace('\t',' ').split(' ')[0].lstrip('@')                  moira.py:1175
        int64_t h0 = line[0].off, h1 = h0;
        const int64_t hend = line[0].off + line[0].len;
        while (h1 < hend && buf[h1] != ' ' && buf[h1] != '\t') h1++;
        while (h0 < h1 && buf[h0] == '@') h0++;
        int64_t *r = idx + n * MIO_IDX_COLS;
        r[MIO_HDR_OFF] = h0; r[MIO_HDR_LEN] = h1 - h0;
        r[MIO_SEQ_OFF] = line[1].off; r[MIO_SEQ_LEN] = line[1].len;
        r[MIO_QUAL_OFF] = line[3].off; r[MIO_QUAL_LEN] = line[3].len;
        int bad = MIO_REC_OK;                                // moira.py:1178-1183, in this order
        if (line[1].len == 0) bad = MIO_REC_EMPTY_SEQ;
        else if (line[3].len == 0) bad = MIO_REC_EMPTY_QUAL;
        else if (line[1].len != line[3].len) bad = MIO_REC_LENGTH_MISMATCH;
        if (bad) { *bad_kind = bad; break; }                 // idx row n describes the offending record
        n++;
        pos = p;
        *consumed = pos;
    }
    return n;
}

// ---- the same index, built by several threads ------------------------------------------------------------------
// "Every 4 lines are one record whatever they contain" (moira/moira.py:1166-1174) makes a record boundary a LINE COUNT
// property, not a content property ('@' may start a quality line), so the buffer cannot be cut by looking for '@'.
// Pass 1 counts the newlines of T byte slices in parallel; the running total gives the line number at every slice
// start, hence each slice's first record start (the first line start whose number is a multiple of 4) and the row of
// the index it belongs to.  Pass 2 runs the sequential indexer on [record start of slice t, record start of slice
// t + 1), writing its rows where they belong.  Results -- rows, consumed, bad_kind, MIO_E_UNSUPPORTED -- are those of
// mio_fastq_index on the whole buffer (tests/test_fastio.py compares them on quirky inputs).
int64_t mio_fastq_index_mt(const char *buf, int64_t len, int32_t final, int64_t max_records,
                           int64_t *idx, int64_t *consumed, int32_t *bad_kind, int32_t threads)
{
    if (!buf || len < 0 || max_records < 0 || !idx || !consumed || !bad_kind)
        return fail(MIO_E_INVALID, "mio_fastq_index_mt: bad arguments");
    int T = threads;
    if (T > 64) T = 64;
    if (T <= 1 || len < (int64_t)T * (64 << 10)) return mio_fastq_index(buf, len, final, max_records, idx, consumed, bad_kind);
    struct Part {
        int64_t a = 0, lines = 0;          // slice start, newlines in the slice
        int64_t rec = 0, row0 = 0;         // first record start at or after `a`, and its row of the index
        int64_t n = 0, used = 0; int32_t bad = MIO_REC_OK; char err[sizeof(g_err)] = "";
    };
    std::vector<Part> P((size_t)T + 1);
    for (int t = 0; t <= T; t++) P[(size_t)t].a = len / T * t + (t == T ? len % T : 0);
    auto run = [&](auto &&fn) {            // fn(t) for t = 0..T-1, one thread each (the caller's thread takes t = 0)
        std::vector<std::thread> th;
        int started = 1;
        try {
            th.reserve((size_t)T);
            for (int t = 1; t < T; t++) { th.emplace_back(fn, t); started = t + 1; }
        } catch (...) {}                   // no thread to be had: the rest runs here
        fn(0);
        for (int t = started; t < T; t++) fn(t);
        for (auto &x : th) x.join();
    };
    run([&](int t) {
        const char *p = buf + P[(size_t)t].a, *e = buf + P[(size_t)t + 1].a;
        int64_t c = 0;
        for (; p < e; p++) c += *p == '\n';                          // vectorised by the compiler
        P[(size_t)t].lines = c;
    });
    int64_t before = 0;                                                // newlines before the slice
    for (int t = 0; t <= T; t++) {
        Part &S = P[(size_t)t];
        if (t == T) { S.rec = len; S.row0 = (before + 3) / 4; break; }
        int64_t pos = S.a, line = before;                              // `line` = number of the line that holds byte `pos`
        bool at_start = pos == 0 || buf[pos - 1] == '\n';
        for (;;) {
            if (at_start && line % 4 == 0) break;
            const char *nl = pos < len ? (const char *)memchr(buf + pos, '\n', (size_t)(len - pos)) : nullptr;
            if (!nl) { pos = len; break; }                             // no further line start in the buffer
            pos = nl - buf + 1;
            line++;
            at_start = true;
        }
        S.rec = pos;
        S.row0 = pos >= len ? -1 : line / 4;
        before += S.lines;
    }
    for (int t = T - 1; t >= 0; t--)                                    // slices with no record start take the next one's
        if (P[(size_t)t].row0 < 0) { P[(size_t)t].row0 = P[(size_t)t + 1].row0; P[(size_t)t].rec = P[(size_t)t + 1].rec; }
    run([&](int t) {
        Part &S = P[(size_t)t];
        const Part &N = P[(size_t)t + 1];
        const bool last = N.rec >= len;                                // nothing starts after this range: it runs to the end
        int64_t cap = max_records - S.row0;
        if (!last && N.row0 - S.row0 < cap) cap = N.row0 - S.row0;
        if (cap <= 0 || S.rec >= len || (!last && N.rec <= S.rec)) return;
        const int64_t end = last ? len : N.rec;
        int64_t used = 0;
        int32_t bad = MIO_REC_OK;
        int64_t *rows = idx + S.row0 * MIO_IDX_COLS;
        const int64_t n = mio_fastq_index(buf + S.rec, end - S.rec, last ? final : 0, cap, rows, &used, &bad);
        S.n = n; S.used = used; S.bad = bad;
        if (n < 0) { snprintf(S.err, sizeof(S.err), "%s", g_err); return; }
        const int64_t fix = n + (bad != MIO_REC_OK ? 1 : 0);           // the row after the last holds the offending record
        for (int64_t k = 0; k < fix; k++) {
            rows[k * MIO_IDX_COLS + MIO_HDR_OFF] += S.rec;
            rows[k * MIO_IDX_COLS + MIO_SEQ_OFF] += S.rec;
            rows[k * MIO_IDX_COLS + MIO_QUAL_OFF] += S.rec;
        }
    });
    *consumed = 0;
    *bad_kind = MIO_REC_OK;
    int64_t n = 0;
    for (int t = 0; t < T; t++) {
        const Part &S = P[(size_t)t], &N = P[(size_t)t + 1];
        if (t > 0 && S.rec == P[(size_t)t - 1].rec) continue;           // shared a range with the slice before
        if (S.rec >= len || S.row0 >= max_records) break;
        if (S.n < 0) return fail((int)S.n, "%s", S.err);
        n = S.row0 + S.n;
        *consumed = S.rec + S.used;
        if (S.bad != MIO_REC_OK) { *bad_kind = S.bad; break; }
        const bool last = N.rec >= len;
        if (!last && S.n < N.row0 - S.row0) break;                      // stopped at max_records
    }
    return n;
}

// Fill dst[0, len) from a regular file at `offset`, the range cut over `threads` preads (page-cache copies and the
// first-touch faults of a fresh destination scale with the threads).  Returns the bytes read (short only at the end
// of the file), or -1.
int64_t mio_pread_mt(int32_t fd, int64_t offset, char *dst, int64_t len, int32_t threads)
{
    if (fd < 0 || offset < 0 || len < 0 || (len > 0 && !dst)) return fail(MIO_E_INVALID, "mio_pread_mt: bad arguments");
    int T = threads < 1 ? 1 : (threads > 64 ? 64 : threads);
    if (len < (int64_t)T * (1 << 20)) T = 1;
    std::vector<int64_t> got((size_t)T, 0);
    auto work = [&](int t) {
        const int64_t a = len / T * t, b = t == T - 1 ? len : len / T * (t + 1);
        int64_t done = 0;
        while (a + done < b) {
            const ssize_t r = pread(fd, dst + a + done, (size_t)(b - a - done), (off_t)(offset + a + done));
            if (r < 0) { got[(size_t)t] = -1; return; }
            if (r == 0) break;
            done += r;
        }
        got[(size_t)t] = done;
    };
    std::vector<std::thread> th;
    int started = 1;
    try {
        th.reserve((size_t)T);
        for (int t = 1; t < T; t++) { th.emplace_back(work, t); started = t + 1; }
    } catch (...) {}
    work(0);
    for (int t = started; t < T; t++) work(t);
    for (auto &x : th) x.join();
    int64_t total = 0;
    for (int t = 0; t < T; t++) {
        if (got[(size_t)t] < 0) return fail(MIO_E_INVALID, "pread failed");
        total += got[(size_t)t];
        const int64_t a = len / T * t, b = t == T - 1 ? len : len / T * (t + 1);
        if (got[(size_t)t] < b - a) break;                              // end of file inside this slice
    }
    return total;
}

int32_t mio_pack(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel,
                 int32_t fastq_offset, int32_t max_len, int32_t lower_n_is_base, int64_t row_stride,
                 uint8_t *out, int32_t *lens_out, uint8_t *flags_out, int64_t *bad_record)
{
    if (!buf || !idx || nsel < 0 || row_stride <= 0 || (nsel > 0 && (!out || !lens_out)))
        return fail(MIO_E_INVALID, "mio_pack: bad arguments");
    for (int64_t k = 0; k < nsel; k++) {
        const int64_t *r = idx + (sel ? sel[k] : k) * MIO_IDX_COLS;
        int64_t L = r[MIO_QUAL_LEN];
        if (max_len > 0 && L > max_len) L = max_len;
        if (L > row_stride) {
            if (bad_record) *bad_record = k;
            return fail(MIO_E_INVALID, "read of %lld bases does not fit a %lld-byte row", (long long)L, (long long)row_stride);
        }
        const unsigned char *sq = (const unsigned char *)buf + r[MIO_SEQ_OFF];
        const unsigned char *ql = (const unsigned char *)buf + r[MIO_QUAL_OFF];
        uint8_t *row = out + k * row_stride;
        // branch-free per-byte form (the compiler turns it into byte-wide SIMD): qualities first, then
        // the ambiguity markers over them
        int lo = 0, hi = 0;
        unsigned char any_n = 0;
        if (fastq_offset >= 0 && fastq_offset <= 255) {
            const unsigned char off = (unsigned char)fastq_offset;
            unsigned char below = 0, above = 0;
            for (int64_t i = 0; i < L; i++) {
                const unsigned char c = ql[i];                       // q = c - off            moira.py:1177
                below |= (unsigned char)(c < off);
                above |= (unsigned char)(c > (unsigned)off + 254u);   // only possible for off == 0
                unsigned char q = (unsigned char)(c - off);
                q = q == 0 ? 1 : q;                                  // moira.py:814, bernoullimodule.c:104-107
                const unsigned char b = sq[i];
                const unsigned char up = (unsigned char)(b == 'N');                             // bernoullimodule.c:196
                const unsigned char low = (unsigned char)((b == 'n') & (lower_n_is_base == 0));
                any_n |= up;
                row[i] = up ? 0 : low ? 255 : q;
            }
            lo = below ? -1 : 0;
            hi = above ? -1 : 0;
        } else {
            for (int64_t i = 0; i < L; i++) {
                int q = (int)ql[i] - fastq_offset;
                lo |= q;                                         // sign bit set <=> some q < 0
                hi |= 254 - q;                                   // sign bit set <=> some q > 254
                q = q == 0 ? 1 : q;
                const unsigned char b = sq[i];
                const bool up = b == 'N', low = b == 'n' && !lower_n_is_base;
                any_n |= (unsigned char)up;
                row[i] = up ? 0 : low ? 255 : (uint8_t)q;
            }
        }
        if ((lo | hi) < 0) {
            if (bad_record) *bad_record = k;
            if (lo < 0) return fail(MIO_E_RANGE, "Qualities must have positive values.");
            return fail(MIO_E_RANGE, "quality exceeds the encodable maximum 254");
        }
        memset(row + L, 0, (size_t)(row_stride - L));
        lens_out[k] = (int32_t)L;
        if (flags_out) flags_out[k] = any_n;
    }
    return MIO_OK;
}

// one line of buf starting at p: [a, b) stripped; returns the position after the line, or -1 when the
// line is not complete yet, or -2 for content the byte-level path declines
static int64_t take_line(const char *buf, int64_t len, int64_t p, int final, Span *out)
{
    if (p >= len) return -1;
    const char *nl = (const char *)memchr(buf + p, '\n', (size_t)(len - p));
    int64_t e;
    if (nl) e = nl - buf;
    else if (final) e = len;
    else return -1;
    const char *cr = (const char *)memchr(buf + p, '\r', (size_t)(e - p));
    if (cr && !(nl && cr == nl - 1)) return -2;
    unsigned char any = 0;
    for (int64_t t = p; t < e; t++) any |= (unsigned char)buf[t];
    if (any & 0x80) return -2;
    *out = strip(buf, p, e);
    return nl ? e + 1 : e;
}

int64_t mio_fasta_qual_index(const char *fbuf, int64_t flen, const char *qbuf, int64_t qlen, int32_t final,
                             int64_t max_records, char *out, int64_t out_cap, int64_t *out_idx,
                             int64_t *f_consumed, int64_t *q_consumed, int64_t *out_used)
{
    if (!fbuf || !qbuf || flen < 0 || qlen < 0 || max_records < 0 || !out || !out_idx || !f_consumed || !q_consumed || !out_used)
        return fail(MIO_E_INVALID, "mio_fasta_qual_index: bad arguments");
    *f_consumed = *q_consumed = *out_used = 0;
    int64_t n = 0, fp = 0, qp = 0, w = 0;
    while (n < max_records) {
        Span fh, fs, qh, ql;
        int64_t a = take_line(fbuf, flen, fp, final, &fh);
        int64_t b = a >= 0 ? take_line(fbuf, flen, a, final, &fs) : a;
        int64_t c = take_line(qbuf, qlen, qp, final, &qh);
        int64_t d = c >= 0 ? take_line(qbuf, qlen, c, final, &ql) : c;
        if (a == -2 || b == -2 || c == -2 || d == -2) return fail(MIO_E_UNSUPPORTED, "lone carriage return or non-ASCII byte");
        if (b < 0 || d < 0) {
            // at the end of the files anything but "both exhausted" is the line parser's business
            if (final && !(fp >= flen && qp >= qlen)) return fail(MIO_E_UNSUPPORTED, "truncated record at the end of the files");
            break;
        }
        // header tokens: replace('\t',' ').split(' ')[0].lstrip('>')                 moira.py:1121-1123
        Span tok[2];
        const char *bufs[2] = {fbuf, qbuf};
        const Span hd[2] = {fh, qh};
        for (int k = 0; k < 2; k++) {
            int64_t h0 = hd[k].off, h1 = h0;
            const int64_t hend = hd[k].off + hd[k].len;
            while (h1 < hend && bufs[k][h1] != ' ' && bufs[k][h1] != '\t') h1++;
            while (h0 < h1 && bufs[k][h0] == '>') h0++;
            tok[k] = {h0, h1 - h0};
        }
        if (tok[0].len != tok[1].len || memcmp(fbuf + tok[0].off, qbuf + tok[1].off, (size_t)tok[0].len) != 0)
            return fail(MIO_E_UNSUPPORTED, "fasta and qual headers differ");          // NameMismatchError: line parser
        if (fs.len == 0 || ql.len == 0) return fail(MIO_E_UNSUPPORTED, "empty sequence or quality line");
        if (w + tok[0].len + 2 * fs.len > out_cap) break;                             // caller's buffer is full
        char *slot = out + w;
        memcpy(slot, fbuf + tok[0].off, (size_t)tok[0].len);
        memcpy(slot + tok[0].len, fbuf + fs.off, (size_t)fs.len);
        // qualities: decimal integers 0..254 separated by single spaces or tabs (moira.py:1124: map(int, ...split(' ')))
        unsigned char *qo = (unsigned char *)slot + tok[0].len + fs.len;
        int64_t nq = 0;
        const char *t = qbuf + ql.off, *tend = t + ql.len;
        while (t < tend) {
            if (*t < '0' || *t > '9') return fail(MIO_E_UNSUPPORTED, "quality token the byte-level path declines");
            int v = 0;
            while (t < tend && *t >= '0' && *t <= '9' && v <= 254) v = v * 10 + (*t++ - '0');
            if (v > 254) return fail(MIO_E_UNSUPPORTED, "quality above 254");
            if (nq >= fs.len) return fail(MIO_E_UNSUPPORTED, "sequence and qualities differ in length");
            qo[nq++] = (unsigned char)v;
            if (t < tend) {
                if (*t != ' ' && *t != '\t') return fail(MIO_E_UNSUPPORTED, "quality token the byte-level path declines");
                t++;
                if (t == tend) return fail(MIO_E_UNSUPPORTED, "trailing separator");
            }
        }
        if (nq != fs.len) return fail(MIO_E_UNSUPPORTED, "sequence and qualities differ in length");
        int64_t *r = out_idx + n * MIO_IDX_COLS;
        r[MIO_HDR_OFF] = w; r[MIO_HDR_LEN] = tok[0].len;
        r[MIO_SEQ_OFF] = w + tok[0].len; r[MIO_SEQ_LEN] = fs.len;
        r[MIO_QUAL_OFF] = w + tok[0].len + fs.len; r[MIO_QUAL_LEN] = fs.len;
        w += tok[0].len + 2 * fs.len;
        n++;
        fp = b; qp = d;
        *f_consumed = fp; *q_consumed = qp; *out_used = w;
    }
    return n;
}

int64_t mio_first_header_mismatch(const char *fbuf, const int64_t *fidx, const char *rbuf, const int64_t *ridx, int64_t n)
{
    for (int64_t k = 0; k < n; k++) {
        const int64_t *f = fidx + k * MIO_IDX_COLS, *r = ridx + k * MIO_IDX_COLS;
        if (f[MIO_HDR_LEN] != r[MIO_HDR_LEN] ||
            memcmp(fbuf + f[MIO_HDR_OFF], rbuf + r[MIO_HDR_OFF], (size_t)f[MIO_HDR_LEN]) != 0)
            return k;
    }
    return -1;
}

int32_t mio_py2_hash(const char *buf, const int64_t *idx, int64_t n, int32_t max_len, uint64_t *out)
{
    if (!buf || !idx || n < 0 || (n > 0 && !out)) return fail(MIO_E_INVALID, "mio_py2_hash: bad arguments");
    for (int64_t k = 0; k < n; k++) {
        const int64_t *r = idx + k * MIO_IDX_COLS;
        int64_t L = r[MIO_SEQ_LEN];
        if (max_len > 0 && L > max_len) L = max_len;
        const unsigned char *s = (const unsigned char *)buf + r[MIO_SEQ_OFF];
        out[k] = py2_hash(s, L);                              // CPython 2.7 Objects/stringobject.c string_hash
    }
    return MIO_OK;
}

int64_t mio_format(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel, int32_t kind,
                   int32_t fastq_offset, int32_t out_offset, int32_t clamp_q0, int32_t max_len, const char *relabel,
                   const int64_t *relabel_index,
                   const double *ee, const char *const *labels, const int32_t *label_id,
                   char *out, int64_t cap, int64_t *needed)
{
    if (!buf || !idx || nsel < 0 || kind < MIO_FMT_FASTA || kind > MIO_FMT_FASTQ || cap < 0 || (cap > 0 && !out) ||
        (relabel && !relabel_index) || (label_id && !labels))
        return fail(MIO_E_INVALID, "mio_format: bad arguments");
    Out o{out, cap, 0};
    const int64_t relabel_len = relabel ? (int64_t)strlen(relabel) : 0;
    char num[64];
    for (int64_t k = 0; k < nsel; k++) {
        const int64_t *r = idx + (sel ? sel[k] : k) * MIO_IDX_COLS;
        o.ch(kind == MIO_FMT_FASTQ ? '@' : '>');
        if (relabel) {                                       // moira.py:854-855
            o.put(relabel, relabel_len);
            o.put(num, snprintf(num, sizeof(num), "%lld", (long long)relabel_index[k]));
        } else {
            const char *h = buf + r[MIO_HDR_OFF];
            const int64_t hl = r[MIO_HDR_LEN];
            if (o.n + hl <= o.cap)
                for (int64_t i = 0; i < hl; i++) o.p[o.n + i] = h[i] == ':' ? '_' : h[i];   // moira.py:1175
            o.n += hl;
        }
        if (ee) o.put(num, snprintf(num, sizeof(num), ";ee=%.2f;size=1;", ee[k]));           // moira.py:858-863
        if (label_id && label_id[k] >= 0) {
            const char *lab = labels[label_id[k]];
            o.ch('\t');
            o.put(lab, (int64_t)strlen(lab));
        }
        o.ch('\n');
        int64_t L = r[MIO_SEQ_LEN];
        if (max_len > 0 && L > max_len) L = max_len;
        if (kind != MIO_FMT_QUAL) {
            o.put(buf + r[MIO_SEQ_OFF], L);
            o.ch('\n');
        }
        if (kind == MIO_FMT_FASTQ) { o.ch('+'); o.ch('\n'); }
        if (kind != MIO_FMT_FASTA) {
            const unsigned char *ql = (const unsigned char *)buf + r[MIO_QUAL_OFF];
            if (kind == MIO_FMT_FASTQ) put_qual_string(o, ql, L, fastq_offset, out_offset, clamp_q0 ? 1 : 0);
            else put_quals(o, ql, L, fastq_offset, clamp_q0 ? 1 : 0);
            o.ch('\n');
        }
    }
    if (needed) *needed = o.n;
    if (o.n > cap) return fail(MIO_E_SPACE, "output needs %lld bytes", (long long)o.n);
    return o.n;
}


int64_t mio_format_report(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel,
                          const char *relabel, const int64_t *relabel_index, const double *ee,
                          const int32_t *aux, char *out, int64_t cap, int64_t *needed)
{
    if (!buf || !idx || nsel < 0 || !aux || cap < 0 || (cap > 0 && !out) || (relabel && !relabel_index))
        return fail(MIO_E_INVALID, "mio_format_report: bad arguments");
    Out o{out, cap, 0};
    const int64_t relabel_len = relabel ? (int64_t)strlen(relabel) : 0;
    char num[96];
    for (int64_t k = 0; k < nsel; k++) {
        const int64_t rec = sel ? sel[k] : k;
        const int64_t *r = idx + rec * MIO_IDX_COLS;
        if (relabel) {
            o.put(relabel, relabel_len);
            o.put(num, snprintf(num, sizeof(num), "%lld", (long long)relabel_index[k]));
        } else {
            const char *h = buf + r[MIO_HDR_OFF];
            const int64_t hl = r[MIO_HDR_LEN];
            if (o.n + hl <= o.cap)
                for (int64_t i = 0; i < hl; i++) o.p[o.n + i] = h[i] == ':' ? '_' : h[i];
            o.n += hl;
        }
        if (ee) o.put(num, snprintf(num, sizeof(num), ";ee=%.2f;size=1;", ee[k]));
        o.put(num, snprintf(num, sizeof(num), "\t1\t%d\t%d\t%d\n", aux[3 * rec], aux[3 * rec + 1], aux[3 * rec + 2]));   // moira.py:868
    }
    if (needed) *needed = o.n;
    if (o.n > cap) return fail(MIO_E_SPACE, "output needs %lld bytes", (long long)o.n);
    return o.n;
}

mio_collapse *mio_collapse_create(void) { return new (std::nothrow) mio_collapse(); }
void mio_collapse_destroy(mio_collapse *c) { delete c; }
int64_t mio_collapse_count(const mio_collapse *c) { return c ? c->count() : 0; }
int32_t mio_collapse_set_threads(mio_collapse *c, int32_t threads)
{
    if (!c) return fail(MIO_E_INVALID, "mio_collapse_set_threads: bad arguments");
    c->threads = threads < 1 ? 1 : (threads > mio_collapse::NSHARD ? mio_collapse::NSHARD : threads);
    return MIO_OK;
}

int32_t mio_collapse_add(mio_collapse *c, const char *buf, const int64_t *idx, int64_t n, int32_t max_len,
                         const double *ee, const uint8_t *flags, const int32_t *aux)
{
    if (!c || !buf || !idx || n < 0 || (n > 0 && !ee)) return fail(MIO_E_INVALID, "mio_collapse_add: bad arguments");
    c->exported = false;
    std::vector<uint64_t> hashes;
    try { hashes.resize((size_t)n); } catch (const std::bad_alloc &) { return fail(MIO_E_INVALID, "out of memory while collapsing"); }
    auto seq_of = [&](int64_t k, int64_t *L) {
        const int64_t *r = idx + k * MIO_IDX_COLS;
        *L = r[MIO_SEQ_LEN];
        if (max_len > 0 && *L > max_len) *L = max_len;
        return (const unsigned char *)buf + r[MIO_SEQ_OFF];
    };
    // the string hash is one dependent multiply-xor per byte: four records' chains run interleaved
    auto hash_range = [&](int64_t lo, int64_t hi) {
        int64_t k4 = lo;
        for (; k4 + 4 <= hi; k4 += 4) {
            const unsigned char *p[4];
            int64_t len4[4], common = INT64_MAX;
            for (int j = 0; j < 4; j++) { p[j] = seq_of(k4 + j, &len4[j]); common = len4[j] < common ? len4[j] : common; }
            if (common <= 0) { for (int j = 0; j < 4; j++) hashes[(size_t)(k4 + j)] = py2_hash(p[j], len4[j]); continue; }
            uint64_t x0 = (uint64_t)p[0][0] << 7, x1 = (uint64_t)p[1][0] << 7, x2 = (uint64_t)p[2][0] << 7, x3 = (uint64_t)p[3][0] << 7;
            for (int64_t i = 0; i < common; i++) {
                x0 = (1000003ull * x0) ^ p[0][i]; x1 = (1000003ull * x1) ^ p[1][i];
                x2 = (1000003ull * x2) ^ p[2][i]; x3 = (1000003ull * x3) ^ p[3][i];
            }
            uint64_t x[4] = {x0, x1, x2, x3};
            for (int j = 0; j < 4; j++) {
                for (int64_t i = common; i < len4[j]; i++) x[j] = (1000003ull * x[j]) ^ p[j][i];
                x[j] ^= (uint64_t)len4[j];
                hashes[(size_t)(k4 + j)] = x[j] == ~0ull ? ~0ull - 1 : x[j];
            }
        }
        for (; k4 < hi; k4++) { int64_t L; const unsigned char *q = seq_of(k4, &L); hashes[(size_t)k4] = py2_hash(q, L); }
    };
    const int T = n < 8192 ? 1 : c->threads;
    std::vector<int> oom((size_t)T, 0);
    const int64_t base = c->reads_seen;
    // reads of the shards s with s % T == t, in file order (moira.py:461-475)
    auto fill = [&](int t) {
        try {
            for (int64_t k = 0; k < n; k++) {
                const uint64_t h = hashes[(size_t)k];
                const int sid = mio_collapse::shard_of(h);
                if (sid % T != t) continue;
                mio_collapse::Shard &S = c->shard[sid];
                const int64_t *r = idx + k * MIO_IDX_COLS;
                int64_t L = r[MIO_SEQ_LEN];
                if (max_len > 0 && L > max_len) L = max_len;
                const char *seq = buf + r[MIO_SEQ_OFF];
                const int64_t hl = r[MIO_HDR_LEN];
                int64_t hoff;
                char *hw = S.reserve(hl, &hoff);
                const char *hs = buf + r[MIO_HDR_OFF];
                for (int64_t i = 0; i < hl; i++) hw[i] = hs[i] == ':' ? '_' : hs[i];      // moira.py:1175
                const int64_t name_id = (int64_t)S.names.size();
                S.names.push_back({hoff, (int32_t)hl, -1});
                uint64_t slot = 0;
                const int64_t hit = S.find(h, seq, L, &slot);
                if (hit < 0) {                                                          // moira.py:461-464
                    mio_collapse::Uniq u;
                    u.seq_off = S.put(seq, L);
                    u.qual_off = S.put(buf + r[MIO_QUAL_OFF], L);
                    u.len = (int32_t)L; u.flags = flags ? flags[k] : 0; u.ee = ee[k]; u.size = 1;
                    for (int a = 0; a < 3; a++) u.aux[a] = aux ? aux[3 * k + a] : 0;
                    u.front = -1; u.back_head = u.back_tail = name_id;
                    u.hash = h; u.first_seen = base + k;
                    S.tab[slot] = (int64_t)S.uniq.size();
                    S.uniq.push_back(u);
                } else {
                    mio_collapse::Uniq &u = S.uniq[(size_t)hit];
                    u.size++;
                    if (ee[k] < u.ee) {                                                 // moira.py:466-471: strict <
                        u.ee = ee[k];
                        for (int a = 0; a < 3; a++) u.aux[a] = aux ? aux[3 * k + a] : 0;
                        u.qual_off = S.put(buf + r[MIO_QUAL_OFF], L);
                        S.names[(size_t)name_id].next = u.front;                        // names_info.insert(0, header)
                        u.front = name_id;
                    } else {                                                            // names_info.append(header)
                        S.names[(size_t)u.back_tail].next = name_id;
                        u.back_tail = name_id;
                    }
                }
            }
        } catch (const std::bad_alloc &) {
            oom[(size_t)t] = 1;
        }
    };
    auto run = [&](auto &&fn) {
        std::vector<std::thread> th;
        int started = 1;
        try {
            th.reserve((size_t)T);
            for (int t = 1; t < T; t++) { th.emplace_back(fn, t); started = t + 1; }
        } catch (...) {}
        fn(0);
        for (int t = started; t < T; t++) fn(t);
        for (auto &x : th) x.join();
    };
    int64_t before[mio_collapse::NSHARD];
    for (int sid = 0; sid < mio_collapse::NSHARD; sid++) before[sid] = (int64_t)c->shard[sid].uniq.size();
    run([&](int t) { hash_range(n / T * t, t == T - 1 ? n : n / T * (t + 1)); });
    run(fill);
    c->reads_seen += n;
    for (int t = 0; t < T; t++)
        if (oom[(size_t)t]) return fail(MIO_E_INVALID, "out of memory while collapsing");
    // the keys this chunk created, in the order the reads came: into the simulated Python-2 dict
    try {
        struct Key { int64_t first_seen; int64_t local; uint8_t sid; };
        std::vector<Key> fresh;
        for (int sid = 0; sid < mio_collapse::NSHARD; sid++)
            for (int64_t u = before[sid]; u < (int64_t)c->shard[sid].uniq.size(); u++)
                fresh.push_back({c->shard[sid].uniq[(size_t)u].first_seen, u, (uint8_t)sid});
        std::sort(fresh.begin(), fresh.end(), [](const Key &a, const Key &b) { return a.first_seen < b.first_seen; });
        std::vector<uint64_t> hs(fresh.size());
        for (size_t k = 0; k < fresh.size(); k++) hs[k] = c->shard[fresh[k].sid].uniq[(size_t)fresh[k].local].hash;
        for (size_t k = 0; k < fresh.size(); k++) {
            if (k + 12 < fresh.size()) c->dict.prefetch(hs[k + 12]);
            c->dict.insert_new(hs[k], (int64_t)c->created.size());
            c->created.push_back({fresh[k].local, fresh[k].sid});
        }
    } catch (const std::bad_alloc &) {
        return fail(MIO_E_INVALID, "out of memory while collapsing");
    }
    return MIO_OK;
}

int32_t mio_collapse_export(mio_collapse *c, double *ee, int64_t *len, int64_t *size, uint8_t *flags, int32_t *aux)
{
    if (!c) return fail(MIO_E_INVALID, "mio_collapse_export: bad arguments");
    try {
        auto group = [&](int64_t g) -> const mio_collapse::Uniq & {
            const auto &r = c->created[(size_t)g];
            return c->shard[r.sid].uniq[(size_t)r.local];
        };
        // sorted(uniques, key=abundance, reverse=True): stable, on dict iteration (= slot) order  moira.py:492
        std::vector<int64_t> order;
        order.reserve(c->created.size());
        for (const auto &sl : c->dict.slots)
            if (sl.id >= 0) order.push_back(sl.id);
        std::vector<int64_t> sizes(order.size());
        int64_t top = 0;
        for (size_t k = 0; k < order.size(); k++) { sizes[k] = group(order[k]).size; top = sizes[k] > top ? sizes[k] : top; }
        std::vector<int64_t> sorted(order.size());
        if (top <= (int64_t)(4 * order.size() + 1024)) {
            // a counting sort by abundance, descending, stable: most groups of a real run are small
            std::vector<int64_t> start((size_t)top + 2, 0);
            for (size_t k = 0; k < order.size(); k++) start[(size_t)(top - sizes[k]) + 1]++;
            for (size_t v = 1; v < start.size(); v++) start[v] += start[v - 1];
            for (size_t k = 0; k < order.size(); k++) sorted[(size_t)start[(size_t)(top - sizes[k])]++] = order[k];
        } else {
            std::vector<size_t> pos(order.size());
            for (size_t k = 0; k < pos.size(); k++) pos[k] = k;
            std::stable_sort(pos.begin(), pos.end(), [&](size_t a, size_t b) { return sizes[a] > sizes[b]; });
            for (size_t k = 0; k < pos.size(); k++) sorted[k] = order[pos[k]];
        }
        c->g_shard.resize(sorted.size());
        c->g_local.resize(sorted.size());
        for (size_t k = 0; k < sorted.size(); k++) {
            const auto &r = c->created[(size_t)sorted[k]];
            c->g_shard[k] = r.sid;
            c->g_local[k] = r.local;
            const auto &u = group(sorted[k]);
            if (ee) ee[k] = u.ee;
            if (len) len[k] = u.len;
            if (size) size[k] = u.size;
            if (flags) flags[k] = u.flags;
            if (aux) for (int a = 0; a < 3; a++) aux[3 * k + a] = u.aux[a];
        }
        c->exported = true;
    } catch (const std::bad_alloc &) {
        return fail(MIO_E_INVALID, "out of memory while ordering the groups");
    }
    return MIO_OK;
}

int64_t mio_collapse_format(const mio_collapse *c, const int64_t *sel, int64_t nsel, int32_t kind,
                            int32_t fastq_offset, int32_t out_offset, int32_t clamp_q0, const char *relabel, int32_t usearch,
                            const char *const *labels, const int32_t *label_id, const uint8_t *lstrip_gt,
                            char *out, int64_t cap, int64_t *needed)
{
    if (!c || nsel < 0 || (nsel > 0 && !sel) || kind < MIO_FMT_FASTA || kind > MIO_FMT_REPORT || cap < 0 ||
        (cap > 0 && !out) || (label_id && !labels))
        return fail(MIO_E_INVALID, "mio_collapse_format: bad arguments");
    if (!c->exported) return fail(MIO_E_INVALID, "call mio_collapse_export first");
    Out o{out, cap, 0};
    const int64_t relabel_len = relabel ? (int64_t)strlen(relabel) : 0;
    char num[96];
    std::vector<char> hdr;
    for (int64_t k = 0; k < nsel; k++) {
        if (sel[k] < 0 || sel[k] >= (int64_t)c->g_local.size()) return fail(MIO_E_INVALID, "selection out of range");
        const mio_collapse::Shard &S = c->shard[c->g_shard[(size_t)sel[k]]];
        const auto &u = S.uniq[(size_t)c->g_local[(size_t)sel[k]]];
        const auto &rep = S.names[(size_t)(u.front >= 0 ? u.front : u.back_head)];  // names_info[0] == rep_header
        hdr.clear();
        if (relabel) {                                                              // moira.py:854-855, index from 1
            hdr.insert(hdr.end(), relabel, relabel + relabel_len);
            const int m = snprintf(num, sizeof(num), "%lld", (long long)(sel[k] + 1));
            hdr.insert(hdr.end(), num, num + m);
        } else {
            hdr.insert(hdr.end(), S.at(rep.off), S.at(rep.off) + rep.len);
        }
        if (usearch) {                                                              // moira.py:858-863
            const int m = snprintf(num, sizeof(num), ";ee=%.2f;size=%lld;", u.ee, (long long)u.size);
            hdr.insert(hdr.end(), num, num + m);
        }
        if (kind == MIO_FMT_REPORT) {                                               // moira.py:866
            o.put(hdr.data(), (int64_t)hdr.size());
            o.put(num, snprintf(num, sizeof(num), "\t%lld\t%d\t%d\t%d\n", (long long)u.size, u.aux[0], u.aux[1], u.aux[2]));
            continue;
        }
        if (kind == MIO_FMT_NAMES) {                                                // "%s\t%s\n" % (header, ",".join(names_info))
            size_t h0 = 0;
            if (lstrip_gt && lstrip_gt[k])
                while (h0 < hdr.size() && hdr[h0] == '>') h0++;
            o.put(hdr.data() + h0, (int64_t)(hdr.size() - h0));
            o.ch('\t');
            bool first = true;
            for (int pass = 0; pass < 2; pass++)
                for (int64_t id = pass == 0 ? u.front : u.back_head; id >= 0; id = S.names[(size_t)id].next) {
                    if (!first) o.ch(',');
                    first = false;
                    o.put(S.at(S.names[(size_t)id].off), S.names[(size_t)id].len);
                }
            o.ch('\n');
            continue;
        }
        o.ch(kind == MIO_FMT_FASTQ ? '@' : '>');
        o.put(hdr.data(), (int64_t)hdr.size());
        if (label_id && label_id[k] >= 0) {
            const char *lab = labels[label_id[k]];
            o.ch('\t');
            o.put(lab, (int64_t)strlen(lab));
        }
        o.ch('\n');
        if (kind != MIO_FMT_QUAL) { o.put(S.at(u.seq_off), u.len); o.ch('\n'); }
        if (kind == MIO_FMT_FASTQ) { o.ch('+'); o.ch('\n'); put_qual_string(o, (const unsigned char *)S.at(u.qual_off), u.len, fastq_offset, out_offset, clamp_q0 ? 1 : 0); o.ch('\n'); }
        if (kind == MIO_FMT_QUAL) { put_quals(o, (const unsigned char *)S.at(u.qual_off), u.len, fastq_offset, clamp_q0 ? 1 : 0); o.ch('\n'); }
    }
    if (needed) *needed = o.n;
    if (o.n > cap) return fail(MIO_E_SPACE, "output needs %lld bytes", (long long)o.n);
    return o.n;
}

}  // extern "C"
