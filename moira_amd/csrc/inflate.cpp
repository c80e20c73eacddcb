// inflate.cpp -- gzip (RFC 1952) / DEFLATE (RFC 1951) decoder of libmoira_io.so.
//
// Why it exists: with the text path at 10^7 reads/s, a gzip-compressed input -- the usual form of a FASTQ file -- is
// bounded by ONE inflating stream, and zlib's inflate runs at about 250 MB/s of text on the GPU boxes' hosts
// (4.7 x 10^5 reads/s).  The reference reads such files through Python's gzip module (moira/moira.py:1058-1090).
// A DEFLATE stream cannot be split across threads, so the stream itself has to get faster: 64-bit bit buffer refilled
// eight bytes at a time, 11-bit primary decode tables whose entries carry symbol, extra-bit count and code length in
// one word, literals stored without a branch per byte, matches copied eight bytes at a time.
//
// Written from the two RFCs; no code from zlib or any other inflate implementation.  Checked against zlib (Python's
// zlib / gzip modules) on every compression level and strategy, stored / fixed / dynamic blocks, multi-member files,
// chunk boundaries at every offset, truncated and corrupt input (tests/test_inflate.py).
#include "../../include/moira_io.h"

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <atomic>
#include <cstdlib>
#include <initializer_list>
#include <mutex>
#include <new>
#include <thread>
#include <vector>
#if defined(__x86_64__)
#include <immintrin.h>
#endif

namespace {

// ---- CRC-32 (RFC 1952 section 8): eight bytes per step by tables, 64 bytes per step by carry-less multiplication where the
// CPU has it (the check of every member's CRC was a fifth of the decoder's time with the tables alone) -----------------------
struct CrcTables {
    uint32_t t[8][256];
    CrcTables()
    {
        for (uint32_t n = 0; n < 256; n++) {
            uint32_t c = n;
            for (int k = 0; k < 8; k++) c = (c & 1) ? 0xedb88320u ^ (c >> 1) : c >> 1;
            t[0][n] = c;
        }
        for (uint32_t n = 0; n < 256; n++)
            for (int k = 1; k < 8; k++) t[k][n] = t[0][t[k - 1][n] & 0xff] ^ (t[k - 1][n] >> 8);
    }
};
const CrcTables g_crc;

// the register form (no pre/post inversion): state c after the bytes p[0, n)
uint32_t crc32_tables(uint32_t c, const uint8_t *p, size_t n);

#if defined(__x86_64__)
// ---- the same CRC by carry-less multiplication (PCLMULQDQ), 64 bytes per step -------------------------------------------
// A 128-bit register holds 16 message bytes as a polynomial over GF(2) in the bit order of the CRC (bit k of the
// register = coefficient of x^(127-k): the first byte's least significant bit is the highest power).  A register A that
// is followed by D bytes of message contributes A * x^(8D); with A = A_H x^64 + A_L that is congruent (mod P) to
// A_H * (x^(8D+64) mod P) + A_L * (x^(8D) mod P), two 64 x 32-bit products that fit 128 bits again.  PCLMULQDQ on two
// bit-reflected operands yields the reflected product times x, so the constants are x^(e-1) mod P, held bit-reflected
// in the upper half of a 64-bit operand.  Four registers are folded 64 bytes ahead per step (independent multiplies), then
// into one; the last 16 bytes of state and the tail go through the table form (state 0 over the register's bytes is
// exactly "times x^32 mod P").  The constants are computed here from P, and the routine is checked against the table
// form on a test pattern before first use (crc32_pick).
struct ClmulConsts { uint64_t k64[2], k16[2]; };
inline uint32_t xpow_mod_p(int e)                       // x^e mod P, normal bit order (bit t = x^t), P = 0x104C11DB7
{
    uint32_t r = 1;
    for (int i = 0; i < e; i++) r = (r << 1) ^ ((r & 0x80000000u) ? 0x04C11DB7u : 0u);
    return r;
}
inline uint64_t refl_hi(uint32_t v)                     // bit-reflected into the upper half of a 64-bit operand
{
    uint32_t r = 0;
    for (int i = 0; i < 32; i++) r |= ((v >> i) & 1u) << (31 - i);
    return (uint64_t)r << 32;
}
inline ClmulConsts clmul_consts()
{
    ClmulConsts k;
    k.k64[0] = refl_hi(xpow_mod_p(512 + 64 - 1)); k.k64[1] = refl_hi(xpow_mod_p(512 - 1));
    k.k16[0] = refl_hi(xpow_mod_p(128 + 64 - 1)); k.k16[1] = refl_hi(xpow_mod_p(128 - 1));
    return k;
}

__attribute__((target("pclmul,sse4.1")))
uint32_t crc32_clmul(uint32_t c, const uint8_t *p, size_t n)     // register form, like crc32_tables
{
    if (n < 128) return crc32_tables(c, p, n);
    static const ClmulConsts K = clmul_consts();
    const __m128i k64 = _mm_set_epi64x((long long)K.k64[1], (long long)K.k64[0]);
    const __m128i k16 = _mm_set_epi64x((long long)K.k16[1], (long long)K.k16[0]);
    __m128i x0 = _mm_loadu_si128((const __m128i *)p), x1 = _mm_loadu_si128((const __m128i *)(p + 16));
    __m128i x2 = _mm_loadu_si128((const __m128i *)(p + 32)), x3 = _mm_loadu_si128((const __m128i *)(p + 48));
    x0 = _mm_xor_si128(x0, _mm_cvtsi32_si128((int)c));               // the state meets the first four bytes
    p += 64; n -= 64;
#define fold(x, k, d) _mm_xor_si128(_mm_xor_si128(_mm_clmulepi64_si128(x, k, 0x00), _mm_clmulepi64_si128(x, k, 0x11)), d)
    while (n >= 64) {
        x0 = fold(x0, k64, _mm_loadu_si128((const __m128i *)p));
        x1 = fold(x1, k64, _mm_loadu_si128((const __m128i *)(p + 16)));
        x2 = fold(x2, k64, _mm_loadu_si128((const __m128i *)(p + 32)));
        x3 = fold(x3, k64, _mm_loadu_si128((const __m128i *)(p + 48)));
        p += 64; n -= 64;
    }
    x1 = fold(x0, k16, x1);
    x2 = fold(x1, k16, x2);
    x3 = fold(x2, k16, x3);
#undef fold
    uint8_t last[16];
    _mm_storeu_si128((__m128i *)last, x3);
    return crc32_tables(crc32_tables(0, last, 16), p, n);
}
#endif

typedef uint32_t (*crc_fn)(uint32_t, const uint8_t *, size_t);
crc_fn crc32_pick()
{
#if defined(__x86_64__)
    __builtin_cpu_init();
    if (__builtin_cpu_supports("pclmul") && __builtin_cpu_supports("sse4.1") && !getenv("MOIRA_CRC_TABLES")) {
        uint8_t pat[1500];
        uint32_t s = 0x9e3779b9u;
        for (auto &b : pat) { s = s * 1664525u + 1013904223u; b = (uint8_t)(s >> 24); }
        bool same = true;
        for (size_t off : {(size_t)0, (size_t)3}) for (size_t len : {(size_t)128, (size_t)129, (size_t)191, (size_t)192, (size_t)1000, (size_t)1497})
            same = same && crc32_clmul(0xffffffffu, pat + off, len) == crc32_tables(0xffffffffu, pat + off, len)
                        && crc32_clmul(0x12345678u, pat + off, len) == crc32_tables(0x12345678u, pat + off, len);
        if (same) return crc32_clmul;
    }
#endif
    return crc32_tables;
}

uint32_t crc32_update(uint32_t crc, const uint8_t *p, size_t n)
{
    static const crc_fn fn = crc32_pick();
    return ~fn(~crc, p, n);
}

uint32_t crc32_tables(uint32_t c, const uint8_t *p, size_t n)
{
    while (n && ((uintptr_t)p & 7)) { c = g_crc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8); n--; }
    while (n >= 8) {
        uint64_t v;
        memcpy(&v, p, 8);
        v ^= c;
        c = g_crc.t[7][v & 0xff] ^ g_crc.t[6][(v >> 8) & 0xff] ^ g_crc.t[5][(v >> 16) & 0xff] ^ g_crc.t[4][(v >> 24) & 0xff] ^
            g_crc.t[3][(v >> 32) & 0xff] ^ g_crc.t[2][(v >> 40) & 0xff] ^ g_crc.t[1][(v >> 48) & 0xff] ^ g_crc.t[0][v >> 56];
        p += 8; n -= 8;
    }
    while (n--) c = g_crc.t[0][(c ^ *p++) & 0xff] ^ (c >> 8);
    return c;
}

// ---- decode tables ---------------------------------------------------------------------------------------------------------
// An entry is one 32-bit word:
//   bits  0..7   code length in bits (for a sub-table pointer: the primary width) -- what to drop from the bit buffer
//   bits  8..15  kind / extra:  literal: 0x80 | 0;  end of block: 0x40;  sub-table pointer: 0x20 | sub-table bits;
//                length or distance base: number of extra bits (0..13)
//   bits 16..31  literal byte, length / distance base, or sub-table start index
constexpr int LIT_BITS = 11, DIST_BITS = 8;
constexpr uint32_t K_LITERAL = 0x8000u, K_EOB = 0x4000u, K_SUB = 0x2000u;
// Two literals in one entry (both codes fit the 11 index bits -- the usual case in text, where codes are 3..7 bits):
// K_LITERAL | K_LIT2, bits 0..7 = both code lengths together, bits 8..11 = the first code's length (for a decoder that has
// room or bits for one byte only), bits 16..23 / 24..31 = first / second byte.  A symbol costs a dependent load-shift-mask
// chain of ~7 cycles however clever the rest is; this halves the chains per byte.
constexpr uint32_t K_LIT2 = 0x1000u;

struct Tables {
    uint32_t lit[(1 << LIT_BITS) + 2048];      // primary + sub-tables (worst case for 288 symbols of <= 15 bits)
    uint32_t dist[(1 << DIST_BITS) + 512];
};

const uint16_t LEN_BASE[29] = {3, 4, 5, 6, 7, 8, 9, 10, 11, 13, 15, 17, 19, 23, 27, 31, 35, 43, 51, 59, 67, 83, 99, 115, 131, 163, 195, 227, 258};
const uint8_t LEN_EXTRA[29] = {0, 0, 0, 0, 0, 0, 0, 0, 1, 1, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 5, 5, 5, 5, 0};
const uint16_t DIST_BASE[30] = {1, 2, 3, 4, 5, 7, 9, 13, 17, 25, 33, 49, 65, 97, 129, 193, 257, 385, 513, 769, 1025, 1537, 2049, 3073, 4097, 6145,
                                8193, 12289, 16385, 24577};
const uint8_t DIST_EXTRA[30] = {0, 0, 0, 0, 1, 1, 2, 2, 3, 3, 4, 4, 5, 5, 6, 6, 7, 7, 8, 8, 9, 9, 10, 10, 11, 11, 12, 12, 13, 13};

inline uint32_t reverse_bits(uint32_t v, int n)
{
    uint32_t r = 0;
    for (int i = 0; i < n; i++) { r = (r << 1) | (v & 1); v >>= 1; }
    return r;
}

// Canonical Huffman code of `n` symbols with lengths len[] (0 = unused) -> decode table.  `primary_bits` wide primary
// table, sub-tables behind it for longer codes.  value_of(sym) gives the entry's upper 24 bits for a symbol.
// Returns false for an over-subscribed or (other than the one-code special case) incomplete code.
template <typename F>
bool build_table(const uint8_t *len, int n, int primary_bits, uint32_t *table, int table_cap, F value_of)
{
    int count[16] = {0};
    for (int i = 0; i < n; i++) count[len[i]]++;
    count[0] = 0;
    int used = 0;
    for (int l = 1; l <= 15; l++) used += count[l];
    const int psize = 1 << primary_bits;
    if (used == 0) {                                   // no code at all: every lookup is an error (entry 0 = invalid)
        for (int i = 0; i < psize; i++) table[i] = 0;
        return true;
    }
    // Kraft sum
    int64_t left = 1;
    for (int l = 1; l <= 15; l++) {
        left <<= 1;
        left -= count[l];
        if (left < 0) return false;                    // over-subscribed
    }
    if (left > 0 && !(used == 1 && count[1] == 1)) return false;     // incomplete (a single 1-bit code is allowed, RFC 1951 3.2.7)
    uint32_t next_code[16];
    uint32_t code = 0;
    for (int l = 1; l <= 15; l++) { code = (code + (uint32_t)count[l - 1]) << 1; next_code[l] = code; }
    for (int i = 0; i < psize; i++) table[i] = 0;
    // sub-table sizing: for each primary prefix of long codes, the longest code under it
    int sub_next = psize;
    // first pass: short codes fill the primary table; long codes record the needed sub-table width per prefix
    static thread_local uint8_t sub_bits[1 << LIT_BITS];
    for (int i = 0; i < psize; i++) sub_bits[i] = 0;
    uint32_t codes[320];
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        if (!l) continue;
        const uint32_t c = reverse_bits(next_code[l]++, l);        // bits arrive LSB first
        codes[s] = c;
        if (l <= primary_bits) {
            const uint32_t e = (value_of(s) << 8) | (uint32_t)l;
            for (uint32_t k = c; k < (uint32_t)psize; k += 1u << l) table[k] = e;
        } else {
            const uint32_t prefix = c & (uint32_t)(psize - 1);
            if (l - primary_bits > sub_bits[prefix]) sub_bits[prefix] = (uint8_t)(l - primary_bits);
        }
    }
    for (int p = 0; p < psize; p++) {
        if (!sub_bits[p]) continue;
        const int size = 1 << sub_bits[p];
        if (sub_next + size > table_cap) return false;
        table[p] = ((uint32_t)sub_next << 16) | K_SUB | ((uint32_t)sub_bits[p] << 8) | (uint32_t)primary_bits;
        for (int k = 0; k < size; k++) table[sub_next + k] = 0;
        sub_next += size;
    }
    for (int s = 0; s < n; s++) {
        const int l = len[s];
        if (l <= primary_bits) continue;
        const uint32_t c = codes[s];
        const uint32_t prefix = c & (uint32_t)(psize - 1);
        const uint32_t start = table[prefix] >> 16;
        const int sb = sub_bits[prefix];
        const uint32_t e = (value_of(s) << 8) | (uint32_t)(l - primary_bits);
        for (uint32_t k = c >> primary_bits; k < (1u << sb); k += 1u << (l - primary_bits)) table[start + k] = e;
    }
    if (used == 1 && count[1] == 1) {
        // the one-code case: the other 1-bit pattern stays invalid (entry 0)
    }
    return true;
}

// literal pairs over a finished literal/length table (primary part only)
void add_literal_pairs(uint32_t *table)
{
    static thread_local uint32_t single[1 << LIT_BITS];
    memcpy(single, table, sizeof(single));
    for (uint32_t i = 0; i < (1u << LIT_BITS); i++) {
        const uint32_t e1 = single[i];
        if ((e1 & (K_LITERAL | K_SUB)) != K_LITERAL) continue;
        const uint32_t l1 = e1 & 0xff;
        if (l1 >= LIT_BITS) continue;
        const uint32_t e2 = single[i >> l1];
        if ((e2 & (K_LITERAL | K_SUB)) != K_LITERAL) continue;
        const uint32_t l2 = e2 & 0xff;
        if (l1 + l2 > LIT_BITS) continue;                           // the second code must be decided by the index bits alone
        table[i] = (e2 & 0x00ff0000u) << 8 | (e1 & 0x00ff0000u) | K_LITERAL | K_LIT2 | (l1 << 8) | (l1 + l2);
    }
}

inline uint32_t lit_value(int s)
{
    if (s < 256) return ((uint32_t)s << 8) | (K_LITERAL >> 8);
    if (s == 256) return K_EOB >> 8;
    if (s > 285) return 0;                                          // 286, 287 (fixed code only): never valid in data
    return ((uint32_t)LEN_BASE[s - 257] << 8) | LEN_EXTRA[s - 257];
}
inline uint32_t dist_value(int s)
{
    if (s > 29) return 0;                                           // 30, 31 (fixed code only): never valid in data
    return ((uint32_t)DIST_BASE[s] << 8) | DIST_EXTRA[s];
}
// No flag and a zero value: "no such code" (entry 0) or a symbol that must not occur -- a length base is >= 3, a
// distance base >= 1, literals / end-of-block / sub-table pointers carry a flag.
inline bool bad_entry(uint32_t e) { return (e & 0xffffe000u) == 0; }

}  // namespace

struct mio_inflate {
    enum Phase { GZ_HEADER, BLOCK_HEADER, STORED, HUFF, GZ_TRAILER, BETWEEN, DONE } phase = GZ_HEADER;
    uint64_t bitbuf = 0;
    int bitcnt = 0;
    bool last_block = false;
    uint32_t stored_left = 0;
    uint32_t crc = 0;
    uint64_t isize = 0;
    bool any_member = false;
    Tables tb;
    bool fixed_ready = false;
    Tables fixed;
};

namespace {

thread_local char g_ierr[200] = "";
int ifail(int code, const char *msg) { snprintf(g_ierr, sizeof(g_ierr), "%s", msg); return code; }

struct In {
    const uint8_t *p, *end;      // end = real end of the data (8 readable slack bytes follow)
};

inline void refill(uint64_t &bb, int &bc, In &in)     // needs bc <= 63 (the byte-wise fills stop at 63 for that reason)
{
    // top up to >= 56 bits when 8 bytes are readable (slack guarantees readability; bytes past `end` are never consumed
    // as data because every consumer checks what it used against the real end)
    uint64_t v;
    memcpy(&v, in.p, 8);
    bb |= v << bc;
    const int take = (63 - bc) >> 3;
    in.p += take;
    bc += take * 8;
}

}  // namespace

extern "C" {

mio_inflate *mio_inflate_create(void) { return new (std::nothrow) mio_inflate(); }
void mio_inflate_destroy(mio_inflate *s) { delete s; }
const char *mio_inflate_error(void) { return g_ierr; }

uint32_t mio_crc32(uint32_t crc, const uint8_t *p, int64_t n) { return crc32_update(crc, p, (size_t)n); }

// See include/moira_io.h.
int32_t mio_inflate_gzip(mio_inflate *S, const uint8_t *in_buf, int64_t in_len, int32_t final, uint8_t *out, int64_t hist,
                         int64_t out_cap, int64_t *in_used, int64_t *out_used)
{
    if (!S || !in_buf || in_len < 0 || !out || hist < 0 || out_cap < hist || !in_used || !out_used)
        return ifail(MIO_E_INVALID, "mio_inflate_gzip: bad arguments");
    In in{in_buf, in_buf + in_len};
    uint8_t *const out_begin = out + hist;
    uint8_t *op = out_begin;
    uint8_t *const out_end = out + out_cap;
    uint64_t bb = S->bitbuf;
    int bc = S->bitcnt;
    uint8_t *crc_from = op;                 // output not yet folded into the CRC
    int rc = 0;

    // bytes of real input still unread, counting whole bytes parked in the bit buffer
    auto avail = [&]() -> int64_t { return (in.end - in.p) + (bc >> 3); };
    auto fold_crc = [&]() {
        if (op > crc_from) { S->crc = crc32_update(S->crc, crc_from, (size_t)(op - crc_from)); S->isize += (uint64_t)(op - crc_from); crc_from = op; }
    };
    // give the bytes parked in the bit buffer back to the input (byte-aligned phases)
    auto unread_bits = [&]() { in.p -= bc >> 3; bb = 0; bc = 0; };

    for (;;) {
        switch (S->phase) {
        case mio_inflate::GZ_HEADER: {
            // byte aligned; needs the whole header (bounded: we ask for 64 KiB + 18 of lookahead at most, else need input)
            unread_bits();
            if (S->any_member)                     // zero padding after a member is ignored (as gzip does)
                while (in.p < in.end && *in.p == 0) in.p++;
            const uint8_t *p = in.p;
            const int64_t n = in.end - p;
            if (n == 0 && final) {
                if (!S->any_member) { rc = ifail(MIO_E_INVALID, "empty input is not a gzip file"); goto out; }
                S->phase = mio_inflate::DONE;
                break;
            }
            if (n < 10) { if (final) { rc = ifail(MIO_E_INVALID, "truncated gzip header"); goto out; } rc = 0; goto need_input; }
            if (p[0] != 0x1f || p[1] != 0x8b) { rc = ifail(MIO_E_INVALID, "not a gzip file"); goto out; }
            if (p[2] != 8) { rc = ifail(MIO_E_INVALID, "unknown gzip compression method"); goto out; }
            const int flg = p[3];
            if (flg & 0xe0) { rc = ifail(MIO_E_INVALID, "reserved gzip flag bits set"); goto out; }
            int64_t pos = 10;
            bool short_in = false;
            if (flg & 4) {                                           // FEXTRA
                if (n < pos + 2) short_in = true;
                else { const int64_t xl = p[pos] | (p[pos + 1] << 8); pos += 2 + xl; if (n < pos) short_in = true; }
            }
            for (int bit : {8, 16}) {                                // FNAME, FCOMMENT: zero-terminated
                if (short_in || !(flg & bit)) continue;
                const void *z = memchr(p + pos, 0, (size_t)(n - pos));
                if (!z) short_in = true; else pos = (const uint8_t *)z - p + 1;
            }
            if (!short_in && (flg & 2)) { pos += 2; if (n < pos) short_in = true; }     // FHCRC (not verified)
            if (short_in) {
                if (final) { rc = ifail(MIO_E_INVALID, "truncated gzip header"); goto out; }
                // the optional fields (FEXTRA <= 64 KiB + 2; FNAME / FCOMMENT: zero-terminated, no limit in RFC 1952) are
                // parsed from one contiguous view of the input: a header that has not ended after 1 MiB is refused by name
                // instead of surfacing as "no progress" in the caller
                if (n > (1 << 20)) { rc = ifail(MIO_E_INVALID, "gzip header field (FNAME / FCOMMENT) longer than 1 MiB is not supported"); goto out; }
                rc = 0; goto need_input;
            }
            in.p = p + pos;
            S->crc = 0; S->isize = 0; S->any_member = true;
            fold_crc();                       // (nothing pending: crc_from == op here)
            crc_from = op;
            S->phase = mio_inflate::BLOCK_HEADER;
            break;
        }
        case mio_inflate::BLOCK_HEADER: {
            // a dynamic header is at most 3 + 14 + 19*3 + 320*7 bits < 300 bytes: ask for that much (or the end of the input)
            if (!final && avail() < 320) { rc = 0; goto need_input; }
            if (in.end - in.p >= 8) refill(bb, bc, in);
            else { while (bc <= 55 && in.p < in.end) { bb |= (uint64_t)*in.p++ << bc; bc += 8; } }
            auto need = [&](int nbits) -> bool {                      // make nbits available; false: input exhausted
                if (bc >= nbits) return true;
                if (in.end - in.p >= 8) refill(bb, bc, in);
                else while (bc <= 55 && in.p < in.end) { bb |= (uint64_t)*in.p++ << bc; bc += 8; }
                return bc >= nbits;
            };
            auto take = [&](int nbits) -> uint32_t { const uint32_t v = (uint32_t)(bb & ((1ull << nbits) - 1)); bb >>= nbits; bc -= nbits; return v; };
            if (!need(3)) { rc = ifail(MIO_E_INVALID, "truncated deflate stream"); goto out; }
            S->last_block = take(1) != 0;
            const uint32_t type = take(2);
            if (type == 0) {
                // stored: skip to the byte boundary, LEN, NLEN
                const int drop = bc & 7;
                bb >>= drop; bc -= drop;
                if (!need(32)) { rc = ifail(MIO_E_INVALID, "truncated stored block"); goto out; }
                const uint32_t len = take(16), nlen = take(16);
                if ((len ^ nlen) != 0xffffu) { rc = ifail(MIO_E_INVALID, "stored block length check failed"); goto out; }
                S->stored_left = len;
                S->phase = mio_inflate::STORED;
            } else if (type == 1) {
                if (!S->fixed_ready) {
                    uint8_t ll[288], dl[32];
                    for (int i = 0; i < 144; i++) ll[i] = 8;
                    for (int i = 144; i < 256; i++) ll[i] = 9;
                    for (int i = 256; i < 280; i++) ll[i] = 7;
                    for (int i = 280; i < 288; i++) ll[i] = 8;
                    for (int i = 0; i < 32; i++) dl[i] = 5;
                    build_table(ll, 288, LIT_BITS, S->fixed.lit, (int)(sizeof(S->fixed.lit) / 4), lit_value);
                    build_table(dl, 32, DIST_BITS, S->fixed.dist, (int)(sizeof(S->fixed.dist) / 4), dist_value);
                    add_literal_pairs(S->fixed.lit);
                    S->fixed_ready = true;
                }
                memcpy(&S->tb, &S->fixed, sizeof(Tables));
                S->phase = mio_inflate::HUFF;
            } else if (type == 2) {
                if (!need(14)) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; }
                const int hlit = (int)take(5) + 257, hdist = (int)take(5) + 1, hclen = (int)take(4) + 4;
                if (hlit > 286 || hdist > 30) { rc = ifail(MIO_E_INVALID, "too many length or distance codes"); goto out; }
                static const uint8_t order[19] = {16, 17, 18, 0, 8, 7, 9, 6, 10, 5, 11, 4, 12, 3, 13, 2, 14, 1, 15};
                uint8_t cl[19] = {0};
                for (int i = 0; i < hclen; i++) {
                    if (!need(3)) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; }
                    cl[order[i]] = (uint8_t)take(3);
                }
                uint32_t pre[128 + 64];
                if (!build_table(cl, 19, 7, pre, 128 + 64, [](int s) { return (uint32_t)s << 8; })) {
                    rc = ifail(MIO_E_INVALID, "invalid code-length code"); goto out;
                }
                uint8_t lens[320];
                int k = 0;
                const int total = hlit + hdist;
                while (k < total) {
                    if (!need(7 + 7)) { if (bc < 1) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; } }
                    const uint32_t e = pre[bb & 127];
                    if (e == 0 || (int)(e & 0xff) > bc) { rc = ifail(MIO_E_INVALID, "invalid code length symbol"); goto out; }
                    bb >>= (e & 0xff); bc -= (int)(e & 0xff);
                    const int sym = (int)(e >> 16);
                    if (sym < 16) { lens[k++] = (uint8_t)sym; continue; }
                    int rep, val = 0;
                    if (sym == 16) {
                        if (k == 0) { rc = ifail(MIO_E_INVALID, "repeat with no previous length"); goto out; }
                        if (!need(2)) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; }
                        val = lens[k - 1]; rep = 3 + (int)take(2);
                    } else if (sym == 17) {
                        if (!need(3)) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; }
                        rep = 3 + (int)take(3);
                    } else {
                        if (!need(7)) { rc = ifail(MIO_E_INVALID, "truncated dynamic block header"); goto out; }
                        rep = 11 + (int)take(7);
                    }
                    if (k + rep > total) { rc = ifail(MIO_E_INVALID, "code lengths overrun"); goto out; }
                    while (rep--) lens[k++] = (uint8_t)val;
                }
                if (lens[256] == 0) { rc = ifail(MIO_E_INVALID, "no end-of-block code"); goto out; }
                if (!build_table(lens, hlit, LIT_BITS, S->tb.lit, (int)(sizeof(S->tb.lit) / 4), lit_value) ||
                    !build_table(lens + hlit, hdist, DIST_BITS, S->tb.dist, (int)(sizeof(S->tb.dist) / 4), dist_value)) {
                    rc = ifail(MIO_E_INVALID, "invalid literal/length or distance code"); goto out;
                }
                add_literal_pairs(S->tb.lit);
                S->phase = mio_inflate::HUFF;
            } else {
                rc = ifail(MIO_E_INVALID, "invalid block type");
                goto out;
            }
            break;
        }
        case mio_inflate::STORED: {
            unread_bits();
            while (S->stored_left) {
                int64_t n = S->stored_left;
                if (n > in.end - in.p) n = in.end - in.p;
                if (n > out_end - op) n = out_end - op;
                if (n == 0) {
                    if (op == out_end) { rc = 1; goto out; }
                    if (final) { rc = ifail(MIO_E_INVALID, "truncated stored block"); goto out; }
                    rc = 0; goto need_input;
                }
                memcpy(op, in.p, (size_t)n);
                op += n; in.p += n; S->stored_left -= (uint32_t)n;
            }
            S->phase = S->last_block ? mio_inflate::GZ_TRAILER : mio_inflate::BLOCK_HEADER;
            break;
        }
        case mio_inflate::HUFF: {
            const uint32_t *lit = S->tb.lit, *dst = S->tb.dist;
            const uint32_t lit_mask = (1u << LIT_BITS) - 1, dist_mask = (1u << DIST_BITS) - 1;
            // ---- fast loop: >= 16 input bytes and >= 320 output bytes to spare, so nothing inside checks bounds -------------
            while (in.end - in.p >= 16 && out_end - op >= 320) {
                refill(bb, bc, in);                                  // >= 56 bits
                uint32_t e = lit[bb & lit_mask];
                // up to three primary lookups per refill (each consumes at most 15 bits; a pair entry at most 11): one or two
                // literals each.  Two bytes are always stored (the second is overwritten when the entry holds one literal).
#define MIO_LITERAL_STEP                                                                  \
                if ((e & (K_LITERAL | K_SUB)) == K_LITERAL) {                             \
                    bb >>= (e & 0xff); bc -= (int)(e & 0xff);                             \
                    op[0] = (uint8_t)(e >> 16); op[1] = (uint8_t)(e >> 24);               \
                    op += 1 + ((e >> 12) & 1);                                            \
                    e = lit[bb & lit_mask];                                               \
                } else goto general;
                MIO_LITERAL_STEP
                MIO_LITERAL_STEP
                MIO_LITERAL_STEP
#undef MIO_LITERAL_STEP
                continue;                                            // (the entry just looked up is looked up again after the refill)
            general:
                if (e & K_SUB) { const uint32_t nb = (e >> 8) & 0x1f; bb >>= LIT_BITS; bc -= LIT_BITS; e = lit[(e >> 16) + (bb & ((1u << nb) - 1))]; }
                if (e & K_LITERAL) {                                 // a literal with a code longer than 11 bits
                    bb >>= (e & 0xff); bc -= (int)(e & 0xff);
                    *op++ = (uint8_t)(e >> 16);
                    continue;
                }
                if (bad_entry(e)) { rc = ifail(MIO_E_INVALID, "invalid literal/length code"); goto out; }
                bb >>= (e & 0xff); bc -= (int)(e & 0xff);
                if (e & K_EOB) { S->phase = S->last_block ? mio_inflate::GZ_TRAILER : mio_inflate::BLOCK_HEADER; goto next_phase; }
                {
                    // length: base + extra bits (<= 5); at least 56 - 3*15 = 11 bits are left, refill to be safe for the distance
                    const uint32_t xl = (e >> 8) & 0x1f;
                    uint32_t length = (e >> 16) + (uint32_t)(bb & ((1u << xl) - 1));
                    bb >>= xl; bc -= (int)xl;
                    if (bc < 32) refill(bb, bc, in);
                    uint32_t d = dst[bb & dist_mask];
                    if (d & K_SUB) { const uint32_t nb = (d >> 8) & 0x1f; bb >>= DIST_BITS; bc -= DIST_BITS; d = dst[(d >> 16) + (bb & ((1u << nb) - 1))]; }
                    if (bad_entry(d)) { rc = ifail(MIO_E_INVALID, "invalid distance code"); goto out; }
                    bb >>= (d & 0xff); bc -= (int)(d & 0xff);
                    const uint32_t xd = (d >> 8) & 0x1f;
                    const uint32_t distance = (d >> 16) + (uint32_t)(bb & ((1u << xd) - 1));
                    bb >>= xd; bc -= (int)xd;
                    if ((int64_t)distance > op - out) { rc = ifail(MIO_E_INVALID, "distance too far back"); goto out; }
                    const uint8_t *src = op - distance;
                    uint8_t *const stop = op + length;
                    if (distance >= 8) {
                        // eight bytes at a time (may write up to 7 bytes past `stop`: the 320-byte margin covers it)
                        do { uint64_t v; memcpy(&v, src, 8); memcpy(op, &v, 8); op += 8; src += 8; } while (op < stop);
                    } else if (distance == 1) {
                        memset(op, *src, length);
                    } else {
                        do { *op++ = *src++; } while (op < stop);
                    }
                    op = stop;
                }
            }
            // ---- careful loop: near the end of the input or of the output, one symbol at a time, every bound checked -------
            for (;;) {
                while (bc <= 55 && in.p < in.end) { bb |= (uint64_t)*in.p++ << bc; bc += 8; }
                // a whole symbol (length code + extra + distance code + extra) is at most 15 + 5 + 15 + 13 = 48 bits
                if (!final && bc < 48 && in.p == in.end) { rc = 0; goto need_input; }
                if (in.end - in.p >= 16 && out_end - op >= 320) break;                   // margins are back: fast loop again
                uint32_t e = lit[bb & lit_mask];
                int usedb = 0;
                uint64_t b2 = bb;
                if (e & K_SUB) { const uint32_t nb = (e >> 8) & 0x1f; b2 >>= LIT_BITS; usedb += LIT_BITS; e = lit[(e >> 16) + (b2 & ((1u << nb) - 1))]; }
                if (bad_entry(e)) { rc = ifail(MIO_E_INVALID, bc == 0 ? "truncated deflate stream" : "invalid literal/length code"); goto out; }
                usedb += (int)(e & 0xff);
                if (e & K_LIT2) {
                    // a pair: both bytes when there is room and there are bits for both, else only the first one
                    if (op == out_end) { rc = 1; goto out; }
                    if (usedb <= bc && out_end - op >= 2) {
                        op[0] = (uint8_t)(e >> 16); op[1] = (uint8_t)(e >> 24); op += 2;
                        bb >>= usedb; bc -= usedb;
                        continue;
                    }
                    const int l1 = (int)((e >> 8) & 0xf);
                    if (l1 > bc) { rc = ifail(MIO_E_INVALID, "truncated deflate stream"); goto out; }
                    *op++ = (uint8_t)(e >> 16);
                    bb >>= l1; bc -= l1;
                    continue;
                }
                if (usedb > bc) { rc = ifail(MIO_E_INVALID, "truncated deflate stream"); goto out; }
                b2 >>= (e & 0xff);
                if (e & K_LITERAL) {
                    if (op == out_end) { rc = 1; goto out; }
                    *op++ = (uint8_t)(e >> 16);
                    bb = b2; bc -= usedb;
                    continue;
                }
                if (e & K_EOB) {
                    bb = b2; bc -= usedb;
                    S->phase = S->last_block ? mio_inflate::GZ_TRAILER : mio_inflate::BLOCK_HEADER;
                    goto next_phase;
                }
                const uint32_t xl = (e >> 8) & 0x1f;
                const uint32_t length = (e >> 16) + (uint32_t)(b2 & ((1u << xl) - 1));
                b2 >>= xl; usedb += (int)xl;
                uint32_t d = dst[b2 & dist_mask];
                if (d & K_SUB) { const uint32_t nb = (d >> 8) & 0x1f; b2 >>= DIST_BITS; usedb += DIST_BITS; d = dst[(d >> 16) + (b2 & ((1u << nb) - 1))]; }
                if (bad_entry(d)) { rc = ifail(MIO_E_INVALID, usedb >= bc ? "truncated deflate stream" : "invalid distance code"); goto out; }
                usedb += (int)(d & 0xff);
                b2 >>= (d & 0xff);
                const uint32_t xd = (d >> 8) & 0x1f;
                const uint32_t distance = (d >> 16) + (uint32_t)(b2 & ((1u << xd) - 1));
                b2 >>= xd; usedb += (int)xd;
                if (usedb > bc) { rc = ifail(MIO_E_INVALID, "truncated deflate stream"); goto out; }
                if ((int64_t)distance > op - out) { rc = ifail(MIO_E_INVALID, "distance too far back"); goto out; }
                if ((int64_t)length > out_end - op) { rc = 1; goto out; }               // the match does not fit: nothing consumed
                bb = b2; bc -= usedb;
                const uint8_t *src = op - distance;
                for (uint32_t k = 0; k < length; k++) op[k] = src[k];
                op += length;
            }
            break;
        }
        case mio_inflate::GZ_TRAILER: {
            unread_bits();
            if (in.end - in.p < 8) { if (final) { rc = ifail(MIO_E_INVALID, "truncated gzip trailer"); goto out; } rc = 0; goto need_input; }
            fold_crc();
            const uint32_t want_crc = in.p[0] | (in.p[1] << 8) | (in.p[2] << 16) | ((uint32_t)in.p[3] << 24);
            const uint32_t want_size = in.p[4] | (in.p[5] << 8) | (in.p[6] << 16) | ((uint32_t)in.p[7] << 24);
            in.p += 8;
            if (want_crc != S->crc) { rc = ifail(MIO_E_INVALID, "CRC check failed"); goto out; }
            if (want_size != (uint32_t)S->isize) { rc = ifail(MIO_E_INVALID, "length check failed"); goto out; }
            S->phase = mio_inflate::GZ_HEADER;                     // another member, padding, or the end
            break;
        }
        case mio_inflate::BETWEEN:
        case mio_inflate::DONE:
            rc = 2;
            goto out;
        }
    next_phase:;
    }
need_input:
    // everything that could be decoded has been; whole bytes parked in the bit buffer stay there (they are part of a
    // symbol in progress) unless the phase is byte aligned
out:
    if (rc >= 0) {
        fold_crc();
        // Whole bytes still parked in the bit buffer go back to the caller (they were loaded in THIS call: at most 7 bits
        // are ever carried from one call to the next, and bits leave the buffer in the order they came), so that
        // "give the bytes back" in a byte-aligned phase can never reach behind the start of a later input buffer.
        in.p -= bc >> 3;
        bc &= 7;
        S->bitbuf = bb & ((1ull << bc) - 1);
        S->bitcnt = bc;
    }
    *in_used = in.p - in_buf;
    *out_used = op - out_begin;
    return rc;
}

// ---- BGZF: gzip members that state their own compressed size ----------------------------------------------------------
// The blocked gzip of the SAM/BAM specification (section 4.1), which bgzip and Illumina's FASTQ writers produce and which
// this package's own compressed outputs use: every member is at most 64 KiB, starts with FLG = FEXTRA and carries a
// 'B','C' subfield whose value is the member's total size - 1.  Members are independent DEFLATE streams, so once their
// boundaries are known without decoding they inflate on as many threads as there are.
static int bgzf_block_size(const uint8_t *p, int64_t n)       // > 0: size;  0: header incomplete;  -1: not a BGZF member
{
    static const uint8_t magic[4] = {0x1f, 0x8b, 8, 4};          // gzip, deflate, FLG = FEXTRA only
    for (int k = 0; k < 4 && k < n; k++)
        if (p[k] != magic[k]) return -1;
    if (n < 12) return 0;
    const int64_t xlen = p[10] | (p[11] << 8);
    if (n < 12 + xlen) return 0;
    int64_t q = 12;
    const int64_t xend = 12 + xlen;
    while (q + 4 <= xend) {
        const int64_t slen = p[q + 2] | (p[q + 3] << 8);
        if (p[q] == 'B' && p[q + 1] == 'C' && slen == 2 && q + 6 <= xend) {
            const int size = (p[q + 4] | (p[q + 5] << 8)) + 1;
            return size >= (int)(xend + 2 + 8) ? size : -1;     // header + an empty stored/fixed block + trailer at least
        }
        q += 4 + slen;
    }
    return -1;
}

int64_t mio_bgzf_scan(const uint8_t *in, int64_t in_len, int64_t max_blocks, int64_t max_out, int64_t *offs, int32_t *sizes,
                      int64_t *out_offs, int32_t *why)
{
    if (!in || in_len < 0 || max_blocks < 0 || !offs || !sizes || !out_offs || !why) return ifail(MIO_E_INVALID, "mio_bgzf_scan: bad arguments");
    int64_t pos = 0, nb = 0, total = 0;
    *why = 0;
    while (nb < max_blocks) {
        const int size = bgzf_block_size(in + pos, in_len - pos);
        if (size < 0) { *why = 1; break; }
        if (size == 0 || pos + size > in_len) { *why = 0; break; }
        const uint8_t *t = in + pos + size - 4;
        const int64_t isize = t[0] | (t[1] << 8) | (t[2] << 16) | ((int64_t)t[3] << 24);
        if (isize > (1 << 16)) { *why = 1; break; }              // not what the format allows: the serial decoder takes it
        if (nb > 0 && total + isize > max_out) { *why = 2; break; }
        offs[nb] = pos; sizes[nb] = size; out_offs[nb] = total;
        total += isize; pos += size; nb++;
        if (nb == max_blocks) *why = 2;
    }
    out_offs[nb] = total;
    return nb;
}

int32_t mio_bgzf_inflate_mt(const uint8_t *in, const int64_t *offs, const int32_t *sizes, const int64_t *out_offs, int64_t n,
                            uint8_t *out, int32_t threads)
{
    if (!in || !offs || !sizes || !out_offs || n < 0 || (!out && n > 0 && out_offs[n] > 0)) return ifail(MIO_E_INVALID, "mio_bgzf_inflate_mt: bad arguments");
    if (n == 0) return 0;
    int T = threads < 1 ? 1 : threads > 64 ? 64 : threads;
    if (T > n) T = (int)n;
    std::atomic<int64_t> next{0};
    std::atomic<bool> failed{false};
    std::mutex mu;
    int64_t bad_block = -1;
    char bad_msg[200] = "";
    auto work = [&]() {
        mio_inflate *S = new (std::nothrow) mio_inflate();
        uint8_t dummy[8];
        for (;;) {
            const int64_t b = next.fetch_add(4);                  // four blocks (<= 256 KiB of text) per grab
            if (b >= n || failed.load()) break;
            for (int64_t k = b; k < n && k < b + 4; k++) {
                const char *msg = nullptr;
                if (!S) msg = "out of memory";
                else {
                    S->phase = mio_inflate::GZ_HEADER; S->bitbuf = 0; S->bitcnt = 0; S->any_member = false;
                    S->last_block = false; S->stored_left = 0;
                    const int64_t want = out_offs[k + 1] - out_offs[k];
                    int64_t used = 0, made = 0;
                    const int rc = mio_inflate_gzip(S, in + offs[k], sizes[k], 1, want ? out + out_offs[k] : dummy, 0, want, &used, &made);
                    if (rc < 0) msg = g_ierr;
                    else if (rc != 2 || used != sizes[k] || made != want) msg = "block does not match its stated size";
                }
                if (msg) {
                    std::lock_guard<std::mutex> g(mu);
                    if (bad_block < 0 || k < bad_block) { bad_block = k; snprintf(bad_msg, sizeof(bad_msg), "%s", msg); }
                    failed.store(true);
                    break;
                }
            }
        }
        delete S;
    };
    std::vector<std::thread> th;
    try {
        for (int t = 1; t < T; t++) th.emplace_back(work);
    } catch (...) { /* fewer threads: the caller's thread does the rest */ }
    work();
    for (auto &t : th) t.join();
    if (bad_block >= 0) {
        char m[200];
        snprintf(m, sizeof(m), "BGZF block %lld: %.150s", (long long)bad_block, bad_msg);
        return ifail(MIO_E_INVALID, m);
    }
    return 0;
}

}  // extern "C"
