#!/usr/bin/env python3
"""Experiment builds WITHOUT switches in the shipped kernels: a variant is a list of exact text replacements applied
to a COPY of moira_amd/csrc/mpb_kernels.hip; the copy is written next to the original's includes so that it compiles unchanged.

    python tools/experiments/make_variant.py NAME OUT.hip [SRC.hip]   # writes the patched source (NAME "none": SRC as it is, with
                                                                         # absolute includes); exits non-zero if a patch does not apply
    python tools/experiments/make_variant.py --list

Use with tools/experiments/variants.sh:   name:""@/tmp/var/NAME.hip
Every variant here is a MEASUREMENT device; none of them is bit-exact unless it says so.  Results: profiles/r06_*."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "moira_amd", "csrc", "mpb_kernels.hip")

# k_narrow_rs / k_narrow_rg: the panel stream alone (loads + tile writes, two tile reads per panel, no arithmetic) and the
# arithmetic alone (stale panels after the first two) -- round 5's -DMPB_NAR_NOARITH / -DMPB_NAR_NODMA, now patches
RS_LOOP_HEAD = "        if (t + 1 < total) {                                    // in flight while this panel is computed on\n"
RS_HALVES_OPEN = "#pragma unroll\n        for (int h = 0; h < 2; h++) {\n            if (ALIGNED) {"
RS_AFTER_HALVES = "        if (++cur_pk == NP) { cur_pk = 0; cur_sb += W; sread = 0; }"

VARIANTS = {
    # the narrow passes' table as {p'} alone (8 bytes: half the LDS cycles of a look-up), 1 - p recomputed by the IEEE subtraction
    # the host used (one more FP64 instruction per base): round 5's -DMPB_NAR_LUT64.  Bit-exact (mpb_create checks a == 1 - p').
    "nar_lut64": [
        ("typedef double2 nar_entry_t;\n#define NAR_P(e) ((e).y)\n#define NAR_A(e) ((e).x)\n",
         "typedef double nar_entry_t;\n#define NAR_P(e) (e)\n#define NAR_A(e) (1.0 - (e))\n"),
        ("s_p[tid] = tid == 255 ? make_double2(__builtin_nan(\"\"), __builtin_nan(\"\")) : lut_g[tid];",
         "s_p[tid] = tid == 255 ? __builtin_nan(\"\") : lut_g[tid].y;", 3),
    ],
    # k_narrow_rs (rows of a multiple of 64 bytes): the two halves of a panel as CHAINED runs -- half 0's run looks up the first
    # eight bases of half 1 in its tail, half 1's run starts without waiting for the LDS.  Bit-exact; a unified diff.
    "rs_chained_halves": "patches/rs_chained_halves.diff",
    # k_narrow_rg without the planned cut (k_rag_scan, rg_first_group, the groups' costs): claim k = position 63 - k / nwin of window
    # k % nwin (every window's longest groups first), wave gw takes one claim of every stripe of W, odd stripes with the waves in
    # reverse; list slots and counts per GROUP, compacted in group order.  Bit-exact; a unified diff.  (profiles/r06_narrow_variants.txt
    # section 10: within 1.5 % of the planned cut either way -- a SIMD's four waves share its FP64 pipe, so a wave that ends early
    # gives its cycles to the others and the 9 % spread of the waves' ranges never shows.  One atomic claim per group instead: 1.10 ms,
    # a single address takes about 70 M atomics a second.)
    "rg_striped_claims": "patches/rg_striped_claims.diff",
    # k_narrow_rg: when a group is armed, every lane asks for one dword of line c8 (c8 >= 1) of each of the eight rows its load
    # instructions cover -- the rows' later lines are requested from DRAM together with their first, and wait in the L2 / the
    # Infinity Cache for the panel that needs them.  Results unchanged (the dwords are never looked at).
    "rg_touch_rows": [
        ("        rows[lane] = (uint32_t)rowoff;                          // (the loads of the group before this one have all been issued)\n    };\n",
         "        rows[lane] = (uint32_t)rowoff;                          // (the loads of the group before this one have all been issued)\n"
         "        {\n"
         "            const uint32_t line = (c8 >= 1 && 8 * c8 < ld_maxc) ? (uint32_t)(c8 * 128) : 0u;      // (lanes with nothing to ask for: line 0 again)\n"
         "            _Pragma(\"unroll\") for (int j = 0; j < 8; j++)\n"
         "                touch[j] = *(const __attribute__((address_space(1))) uint32_t *)(wbase + (rows[8 * j + r8] + line));\n"
         "        }\n"
         "    };\n"),
        ("    uint32_t *const rows = s_row[w];\n", "    uint32_t *const rows = s_row[w];\n    uint32_t junk = 0, touch[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // EXPERIMENT: where the touch loads land\n"),
        ("            {\n                int next_pk = pk + 1;\n                const bool next_group = next_pk == np && g + 1 < g1;",
         "            if (pk == 0) { _Pragma(\"unroll\") for (int j = 0; j < 8; j++) junk += touch[j]; }     // (older than the panel the tile write waited for)\n"
         "            {\n                int next_pk = pk + 1;\n                const bool next_group = next_pk == np && g + 1 < g1;"),
        ("    if (lane == 0) wave_count[gw] = nlist;\n}\n\n// The waves' list segments",
         "    if (junk == 0x12345678u) ns[0] = 1;\n    if (lane == 0) wave_count[gw] = nlist;\n}\n\n// The waves' list segments"),
    ],
    # k_narrow_rg: the arithmetic alone -- no row is loaded (the tile keeps what it held: zeros / stale bytes; results are garbage)
    "rg_arith_alone": [
        ("        if (8 * pk + c8 < ld_maxc) {                            // (the group's last panel: only the chunks its longest read has)",
         "        if (8 * pk + c8 < ld_maxc && n < 0) {                   // EXPERIMENT: never"),
    ],
    # k_narrow_rg without the forced four waves per SIMD (the compiler then takes 126 / 132 / 138 registers: three waves at R >= 3)
    "rg_no_min_waves": [
        ("__global__ __launch_bounds__(256, 4) void k_narrow_rg(", "__global__ __launch_bounds__(256) void k_narrow_rg("),
    ],
    "rs_arith_alone": [
        (RS_LOOP_HEAD, "        if (t + 1 < total && t < 1) {                           // EXPERIMENT: no loads after the second panel\n"),
    ],
    "rs_stream_alone": [
        (RS_HALVES_OPEN,
         "        nonzero += *reinterpret_cast<const uint32_t *>(tile + x0) + *reinterpret_cast<const uint32_t *>(tile + (x0 ^ 64));\n"
         "        if (false)\n" + RS_HALVES_OPEN),
        (RS_AFTER_HALVES, "        if (nonzero == 0x12345678u) ns[gw] = 1;                 // EXPERIMENT: keeps the tile reads alive\n" + RS_AFTER_HALVES),
    ],
    # every stride through the LDS-DMA ring form k_narrow (round 5's MPB_NAR_NO_RS=1)
    "force_ring": [
        ("    if (stride % 16 != 0 || stride > (1 << 16)) return 0;", "    return 0;                                                     // EXPERIMENT: never k_narrow_rs\n    if (stride % 16 != 0 || stride > (1 << 16)) return 0;"),
    ],
    # k_narrow_rg: no arithmetic (the gather stream, tile writes, epilogues and result stores stay)
    "rg_stream_alone": [
        ("                const int rem = cur_maxc - cb;                   // chunks of the group's longest read from here on\n                if (rem <= 0) continue;",
         "                const int rem = cur_maxc - cb;                   // chunks of the group's longest read from here on\n                if (rem <= 0 || h >= 0) { nonzero += *reinterpret_cast<const uint32_t *>(tile + (x0 ^ (h << 6))); continue; }   // EXPERIMENT"),
    ],
    # k_narrow_rg: results written at the SORTED position (64 g + lane) instead of the read's own index: coalesced stores, wrong
    # placement -- what the scattered 13-byte result stores cost
    "rg_coalesced_results": [
        ("            const int64_t i = cur_idx;\n            const int li = cur_len;",
         "            const int64_t i = cur_idx < 0 ? (int64_t)cur_idx : (int64_t)g * 64 + lane;\n            const int li = cur_len;"),
    ],
    # k_narrow_rg: no result stores at all (the list of handed-back reads stays)
    "rg_no_results": [
        ("                ee[i] = e;\n                ns[i] = nsv;\n                pass[i] = (uint8_t)((prm.ambig_mode == 2 && nsv > 0) ? 0 : (e <= limit ? 1 : 0));   // moira.py:911\n            }\n            const unsigned long long todo = __ballot(valid && !done);",
         "                if (e == 1.2345e-300) ee[i] = e;\n            }\n            const unsigned long long todo = __ballot(valid && !done);"),
    ],
}


def main():
    if len(sys.argv) == 2 and sys.argv[1] == "--list":
        for k in VARIANTS:
            print(k)
        return 0
    name, out = sys.argv[1], sys.argv[2]
    src = sys.argv[3] if len(sys.argv) > 3 else SRC        # another copy of the kernel file (an A/B baseline kept aside); "none": no patch
    s = open(src).read()
    for part in ([] if name == "none" else name.split("+")):     # A+B: both, in that order
        v = VARIANTS[part]
        if isinstance(v, str):                                 # a unified diff next to this script
            import subprocess
            import tempfile
            with tempfile.TemporaryDirectory() as td:
                a, b = os.path.join(td, "a.hip"), os.path.join(td, "b.hip")
                open(a, "w").write(s)
                r = subprocess.run(["patch", "-s", "-o", b, a, os.path.join(os.path.dirname(os.path.abspath(__file__)), v)],
                                   capture_output=True, text=True)
                if r.returncode != 0:
                    sys.stderr.write("variant %s: %s does not apply:\n%s%s\n" % (part, v, r.stdout, r.stderr))
                    return 1
                s = open(b).read()
            continue
        for item in v:
            old, new, times = item if len(item) == 3 else (item[0], item[1], 1)
            if s.count(old) != times:
                sys.stderr.write("variant %s: a patch applies %d times (must be exactly %d):\n%s\n" % (part, s.count(old), times, old))
                return 1
            s = s.replace(old, new)
    # the copy lives elsewhere: make its relative includes absolute
    s = s.replace('#include "mpb_internal.h"', '#include "%s"' % os.path.join(ROOT, "moira_amd", "csrc", "mpb_internal.h"))
    s = s.replace('#include "../../include/mpb_synth.h"', '#include "%s"' % os.path.join(ROOT, "include", "mpb_synth.h"))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    open(out, "w").write(s)
    return 0


if __name__ == "__main__":
    sys.exit(main())
