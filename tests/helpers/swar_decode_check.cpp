// Host check of the word-wide FASTQ decode (decode4) against the byte-by-byte statement of the rules (decode4_bytes):
// both are device functions of moira_amd/csrc/mpb_kernels.hip; tests/test_library_abi.py cuts them out of the source
// into swar_funcs.h and compiles this file with g++.  Random dwords biased towards the interesting bytes (letters
// A C G T N n, quality characters around the offset, incl. below it), every offset class, every valid-byte count.
#include <cstdint>
#include <cstdio>
#include <initializer_list>
#define __device__
#define __forceinline__ inline
#include "swar_funcs.h"
int main()
{
    long bad_cases = 0, n = 0;
    uint32_t seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed >> 8; };
    for (int offset : {1, 2, 33, 64, 127, 128, 200, 255}) {
        for (long it = 0; it < 1000000; it++) {
            uint32_t sw = rnd() ^ (rnd() << 12), qw = rnd() ^ (rnd() << 12);
            for (int t = 0; t < 4; t++) {
                if (rnd() % 3) { const char L[] = "ACGTNnacgt"; sw = (sw & ~(0xffu << (8 * t))) | ((uint32_t)L[rnd() % 10] << (8 * t)); }
                if (rnd() % 2) { const uint32_t v = (uint32_t)(offset + (int)(rnd() % 50) - 3) & 0xffu; qw = (qw & ~(0xffu << (8 * t))) | (v << (8 * t)); }
            }
            const int nv = (int)(rnd() % 9) - 2;
            int b1 = 0, b2 = 0;
            const uint32_t a = decode4_bytes(sw, qw, nv, offset, b1);
            const uint32_t b = decode4(sw, qw, nv, offset, (uint32_t)(offset & 0xff) * 0x01010101u, b2);
            n++;
            if (a != b || b1 != b2) {
                if (bad_cases++ < 10) printf("MISMATCH off=%d sw=%08x qw=%08x nv=%d: %08x/%d vs %08x/%d\n", offset, sw, qw, nv, a, b1, b, b2);
            }
        }
    }
    printf("%ld cases, %ld mismatches\n", n, bad_cases);
    return bad_cases != 0;
}
