// mpb_kernels.hip -- gfx950 (MI355X / CDNA4) kernels of the Poisson-binomial read filter.
//
// Path replaced: moira/bernoullimodule.c:131-263 (prob_j_errors, sum_of_binomials,
// interpolate, test) + the filter half of moira/moira.py:806-831 (process_data) + the
// keep/discard predicate of moira/moira.py:911,925-926,949-950 (write_results).
//
// Pipeline for one HBM-resident batch (all on one stream, no host round trip):
//   k_prepass  streams the quality matrix once (16 B per lane, coalesced), estimates per read
//              mean / variance / third cumulant of the error count, predicts how many DP rows
//              J the read needs (Cornish-Fisher quantile) and assigns a row-budget class;
//              also counts ambiguous bases (Ns) and writes a per-block class histogram.
//   k_scan / k_tables   exclusive scan of the histograms -> stable offsets, tile table.
//   k_scatter  stable counting-sort scatter of read indices by class (deterministic).
//   k_dp       the DP.  One wave = 8 consecutive tiles; a tile = 64/G reads of ONE class, so every
//              lane runs the same trip counts.  Class (G,R): G lanes share a read, each lane keeps R consecutive
//              entries of the running probability vector v[] in VGPRs; per base
//                  v[j] = fl( fl(a*v[j]) + fl(b*v[j-1]) )      (no FMA: bit-exact with the
//              reference, whose inner sum has exactly these two non-zero terms), the row
//              crossing lane boundaries by one DPP wave shift.  {a,b} come from a 4 KB LDS
//              LUT built on the host with libm pow.  Epilogue = sequential CDF, linear
//              interpolation, +Ns / floor / predicate.
//   k_dp (overflow pass)  reads whose CDF did not cross inside their class budget are re-run
//              in the widest class (rare: the predictor is tight).
//   k_count    pass count.
//
// The arithmetic that must match the reference bit for bit is compiled with FP contraction
// off (see build.py: -ffp-contract=off, and the pragma below).

#include "mpb_internal.h"
#include <cstdlib>
#include <cstdio>
#include "../../include/mpb_synth.h"

#pragma clang fp contract(off)

// 4 KB {1-p, p'} LUT in LDS.  Module-scope static LDS: the class bodies (non-inlined functions)
// address it at a link-time constant, so a lookup is one SDWA shift (byte select * 16) + ds_read_b128.
__shared__ double2 mpb_s_lut[256];
// MPB_FLAG_COUNT_CELLS: this workgroup's sum of the ALGORITHMIC DP cells of the reads it reported (k_dp only; one LDS
// atomic per read when the flag is set, one wave-uniform branch per tile when it is not)
__shared__ unsigned long long mpb_s_cells;

namespace {

__constant__ MpbClass c_classes[MPB_NCLS] = MPB_CLASS_TABLE;

// rows -> smallest class whose cap >= rows, built at compile time
struct ClassOfRows {
    uint8_t t[MPB_TILE_MAX_ROWS + 1];          // indexed by rows 0 .. MPB_TILE_MAX_ROWS; more rows = a wide read
    constexpr ClassOfRows() : t{}
    {
        constexpr MpbClass cl[MPB_NCLS] = MPB_CLASS_TABLE;
        for (int rows = 0; rows < MPB_TILE_MAX_ROWS + 1; rows++) {
            int c = 0;
            for (int k = 0; k < MPB_NCLS - 1; k++) c += (rows > cl[k].cap) ? 1 : 0;
            t[rows] = (uint8_t)c;
        }
    }
};
__constant__ ClassOfRows c_class_of_rows{};

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// Loads / stores through pointers that reach a class body inside a struct (DpArgs, parked in LDS) have an
// unknown address space, so the compiler emits FLAT instructions -- which complete out of order with respect
// to LDS traffic, and therefore force `s_waitcnt vmcnt(0) lgkmcnt(0)` before the first use of ANY of them:
// a prefetch issued one super-chunk ahead would be waited for at once.  These helpers state the address
// space (global), which gives global_load / global_store and counted waits (vmcnt(N)).
template <typename T>
__device__ __forceinline__ T gload(const T *p)
{
    return *(const __attribute__((address_space(1))) T *)(p);
}
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 gload16(const uint8_t *p)      // 16 aligned bytes
{
    const u32x4 x = *(const __attribute__((address_space(1))) u32x4 *)(p);
    return make_uint4(x.x, x.y, x.z, x.w);
}
template <typename T>
__device__ __forceinline__ void gstore(T *p, T v)
{
    *(__attribute__((address_space(1))) T *)(p) = v;
}

// Orders this wave's LDS writes before its later LDS reads (LDS operations of one wave execute in
// order; the fence stops the compiler from moving them and drains the counter).  Enough when the
// LDS region is touched by one wave only.
__device__ __forceinline__ void wave_lds_fence()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_s_waitcnt(0xc07f);          // lgkmcnt(0)
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// a length read from a caller-supplied device array never takes a kernel outside its row
__device__ __forceinline__ int clamp_len(int li, int max_len) { return min(max(li, 0), max_len); }

// ------------------------------------------------------------------------------------------
// k_prepass
// ------------------------------------------------------------------------------------------
// A wave takes 64 reads per round, 16 at a time: lane (r, cl) = (lane & 15, lane >> 4) walks the
// 16-byte chunks cl, cl+4, cl+8 ... of read r, so one load instruction covers 64 consecutive bytes
// of each of 16 rows and every partial sum stays in registers until the row is done.  Per byte: one
// ds_read_b64 from a 256-entry float2 LUT {p, p(1-p)}, one packed f32 add (mu, var) and one fma
// (sum of p*p(1-p), from which the third cumulant is var - 2*that).  Ambiguous bytes carry a large
// marker in the second component (128 per 'N', 65536 per 'n'; a lane sees at most 256 bytes, whose
// p(1-p) sum to <= 64, so both counts come off the lane total exactly with two floors -- which is why a
// row longer than 960 bytes is walked in panels of 60 chunk columns, peeled one by one).  Chunks that
// are complete in every lane of the wave take a path without any masking; the ragged tail fills the
// bytes past the read's end with Q254 (p = 4e-26).  The four lanes of a read are then combined in a
// fixed order (deterministic).
typedef float f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ uint32_t mask_dword(uint32_t w, int nvalid_bytes)
{
    // keep the low `nvalid_bytes` bytes (0..4) of w
    if (nvalid_bytes >= 4) return w;
    if (nvalid_bytes <= 0) return 0u;
    return w & ((1u << (8 * nvalid_bytes)) - 1u);
}

__device__ __forceinline__ uint32_t fill_dword(uint32_t w, int nvalid_bytes)
{
    // keep the low `nvalid_bytes` bytes (0..4) of w, set the others to 0xFE
    if (nvalid_bytes >= 4) return w;
    if (nvalid_bytes <= 0) return 0xFEFEFEFEu;
    const uint32_t m = (1u << (8 * nvalid_bytes)) - 1u;
    return (w & m) | (0xFEFEFEFEu & ~m);
}

#define MPB_PRE_NB 5              // column quads (4 x 16 bytes of a row) loaded ahead
#define MPB_MARK_UPPER 128.0f     // second LUT component of 'N'
#define MPB_MARK_LOWER 65536.0f   // ... of 'n' (> 256 * 128)

// 16 bases into the lane's running sums
__device__ __forceinline__ void pre_chunk(const float2 *tab, const uint4 x, f32x2 &a01, float &s3)
{
    const uint32_t ww[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int d = 0; d < 4; d++)
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float2 e = tab[(ww[d] >> (8 * t)) & 0xffu];
            a01 += (f32x2){e.x, e.y};
            s3 = __builtin_fmaf(e.x, e.y, s3);            // p == 0 for marked bytes (a kappa3-free prepass, 2 VALU per byte, was no faster: profiles/r04_prepass_variants.txt)
        }
}

// Raw FASTQ text -> packed qscores (ref: moira/moira.py:1177 `ord(x) - offset`, bernoullimodule.c:104-107 Q0 -> 1,
// :196 N / n).  `bad` counts bytes that decode to Q < 0 or Q > 254 (written as Q1 / Q254).
// Four bases at a time, byte by byte: the plain statement of the rules (and the path for the rare dword that holds
// a quality character below the offset).  `nv` = bases of this dword inside the read (bytes past it become 0).
__device__ __forceinline__ uint32_t decode4_bytes(uint32_t sw, uint32_t qw, int nv, int offset, int &bad)
{
    uint32_t ow = 0;
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const uint32_t b = (sw >> (8 * t)) & 0xffu;
        int qv = (int)((qw >> (8 * t)) & 0xffu) - offset;
        uint32_t o = 0;
        if (t < nv) {
            if (qv < 0) { bad++; qv = 1; }
            if (qv > 254) { bad++; qv = 254; }
            if (qv == 0) qv = 1;
            o = b == 'N' ? 0u : b == 'n' ? 255u : (uint32_t)qv;
        }
        ow |= o << (8 * t);
    }
    return ow;
}

// 0x80 in every byte of x that is zero
__device__ __forceinline__ uint32_t zero_bytes(uint32_t x)
{
    return ~(((x & 0x7f7f7f7fu) + 0x7f7f7f7fu) | x) & 0x80808080u;
}

// The same four bases with word-wide arithmetic (the decode is VALU-bound byte by byte: ~19 instructions per base
// against ~8 here).  Needs offset >= 1 (then Q <= 254 always); a dword with a character below the offset inside the
// read takes the byte path, which counts it.
__device__ __forceinline__ uint32_t decode4(uint32_t sw, uint32_t qw, int nv, int offset, uint32_t off4, int &bad)
{
    const uint32_t H = 0x80808080u;
    const uint32_t d = ((qw | H) - (off4 & ~H)) ^ ((qw ^ ~off4) & H);          // per-byte qw - offset (mod 256), no borrow across bytes
    uint32_t lt = ((~qw & off4) | (~(qw ^ off4) & d)) & H;                      // 0x80 where the byte of qw is below the offset
    uint32_t keep = 0xffffffffu;
    if (nv < 4) keep = nv <= 0 ? 0u : ((1u << (8 * nv)) - 1u);
    lt &= keep;
    if (lt) return decode4_bytes(sw, qw, nv, offset, bad);                      // rare
    uint32_t q = d + (zero_bytes(d) >> 7);                                      // Q0 -> 1
    // 'N' (0x4E) -> 0, 'n' (0x6E) -> 255: the two letters differ in bit 5 only
    const uint32_t amb = zero_bytes((sw | 0x20202020u) ^ 0x6e6e6e6eu);
    const uint32_t amb_full = (amb << 1) - (amb >> 7);                          // 0xff in flagged bytes
    const uint32_t lower_full = ((sw >> 5) & 0x01010101u) * 255u;               // 0xff where bit 5 is set ('n' among the flagged)
    q = (amb_full & lower_full) | (~amb_full & q);
    return q & keep;
}

// 16 bases; `pos0` = position of the first base in the read, bases at or past `li` become 0
__device__ __forceinline__ uint4 decode16(const uint4 sq, const uint4 ql, int pos0, int li, int offset, int &bad)
{
    const int nv = li - pos0;
    if (offset < 1)                                                             // wave-uniform; Q = 255 is possible: byte path
        return make_uint4(decode4_bytes(sq.x, ql.x, nv, offset, bad), decode4_bytes(sq.y, ql.y, nv - 4, offset, bad),
                          decode4_bytes(sq.z, ql.z, nv - 8, offset, bad), decode4_bytes(sq.w, ql.w, nv - 12, offset, bad));
    const uint32_t off4 = (uint32_t)(offset & 0xff) * 0x01010101u;
    if (offset > 255) {                                                         // every character is below the offset
        return make_uint4(decode4_bytes(sq.x, ql.x, nv, offset, bad), decode4_bytes(sq.y, ql.y, nv - 4, offset, bad),
                          decode4_bytes(sq.z, ql.z, nv - 8, offset, bad), decode4_bytes(sq.w, ql.w, nv - 12, offset, bad));
    }
    return make_uint4(decode4(sq.x, ql.x, nv, offset, off4, bad), decode4(sq.y, ql.y, nv - 4, offset, off4, bad),
                      decode4(sq.z, ql.z, nv - 8, offset, off4, bad), decode4(sq.w, ql.w, nv - 12, offset, off4, bad));
}

// where the classified-at-source pass (k_classify_linear<., DECODE>) takes its bytes from, and where it leaves the packed matrix
struct PreDecode {
    const uint8_t *seq;       // n x stride base letters
    const uint8_t *qual;      // n x stride FASTQ quality characters
    uint8_t *out;             // n x stride packed qscores (written here: the matrix the DP then reads)
    int32_t offset;           // --fastq_offset
    int32_t *err;             // device counter of undecodable bytes, may be nullptr
};

// where the classification of a read goes
struct PreOut {
    uint8_t *cls; int32_t *ns; double *ee; uint8_t *pass;
    int32_t *bad_len, *wide_list, *wide_rows, *wide_count;
};

// One read's statistics -> row budget -> class byte, Ns, histogram count (or: settled / wide / bad length).
// mu, var, k3: fp32 sums of p, p(1-p), p(1-p)(1-2p) over the scored bases; ambi = 'N' count | 'n' count << 16;
// li = the (clamped) length, bad = the caller-supplied length did not fit the row.
// i: the read (not its position in the pass, when the pass runs over a LIST of reads: k_prepass<.., LISTED>): class bytes, ns, ee and
// pass are all indexed by read -- the DP looks a read's class byte up by its id (the 'has an upper-case N' bit, moira.py:911).
__device__ __forceinline__ void class_read(int64_t i, float mu, float var, float k3, uint32_t ambi, int li, bool bad, bool ragged,
                                           const MpbDevParams &prm, const PreOut &o, int *s_hist, int nb)
{
    const int nzero = (int)(ambi & 0xffffu), n255 = (int)(ambi >> 16);      // unsigned: a read may hold more than 32767 'n'
    // Cornish-Fisher estimate of the (1-alpha) quantile of the error count; the DP needs
    // rows 0..j* where j* is the first row whose CDF exceeds 1-alpha.
    const float v = fmaxf(var, 1e-12f);
    const float x = mu + prm.z * sqrtf(v) + (k3 / v) * prm.zq;
    int rows = (int)floorf(fminf(x, 1e9f) + 0.5f) + 1;
    if (prm.flags & 4u) rows = rows / 2;                          // MPB_FLAG_TEST_UNDERPREDICT
    const int scored = li - nzero - n255;
    rows = min(rows, scored + 1);
    rows = max(rows, 1);
    o.ns[i] = bad ? 0 : nzero + n255;
    const int c = c_class_of_rows.t[min(rows, MPB_TILE_MAX_ROWS)];
    bool settled = false;
    if (!bad && (prm.flags & 8u)) {                               // MPB_FLAG_DECISION_ONLY
        // Chernoff: P(X <= (1-d)mu) <= exp(-d^2 mu / 2) <= 1-alpha for d = clow/sqrt(mu), so the
        // first CDF row above 1-alpha is > t = mu - clow*sqrt(mu) and ee >= floor(t).  mu is an
        // fp32 sum of approximated p: shave 1e-4 relative and 0.02 absolute before trusting it.
        const float t = mu * (1.0f - 1e-4f) - prm.clow * sqrtf(mu) - 0.02f;
        const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors : (double)li * prm.uncert;
        settled = mu > 1.0f && (double)floorf(t) > limit;
    }
    if (bad) {
        // a length outside 0..max_len is never clamped silently: the read gets no result (NaN, rejected) --
        // also for a caller that never fetches the counts -- and the call that does fetch them fails
        atomicAdd(o.bad_len, 1);
        o.cls[i] = (uint8_t)MPB_CLS_SETTLED;
        o.ee[i] = __builtin_nan("");
        o.pass[i] = 0;
    } else if (settled) {
        o.cls[i] = (uint8_t)(MPB_CLS_SETTLED | (nzero > 0 ? 0x80 : 0));
        o.ee[i] = __builtin_inf();              // "certainly above the threshold"; NaN stays a failure
        o.pass[i] = 0;
    } else if (rows > MPB_TILE_MAX_ROWS) {
        // wide read: more rows than one wave holds -> listed for k_wide (only possible when max_len >= 1024,
        // and then the host has provided the list)
        const int pos = atomicAdd(o.wide_count, 1);
        o.wide_list[pos] = (int32_t)i;
        o.wide_rows[pos] = rows;
        o.cls[i] = (uint8_t)(MPB_CLS_WIDE | (nzero > 0 ? 0x80 : 0));
    } else {
        o.cls[i] = (uint8_t)(c | (nzero > 0 ? 0x80 : 0));
        atomicAdd(&s_hist[c * nb + (ragged ? min(MPB_LEN_BINS - 1, li >> prm.len_shift) : 0)], 1);
    }
}

// RAGGED <=> len != nullptr; the fixed-length instance has one length bin and no len loads.
// LONG <=> rows of more than 960 bytes: walked in panels (below); the short-row instances are one panel by construction.
// (Classified at source -- text in, packed matrix out, classified on the way -- is k_classify_linear below: with three
// streams to move, a linear walk beats this kernel's row-per-lane shape; for the one stream here it is the other way round.)
// LISTED (round 6) <=> the pass runs over the n reads list[0 .. n) of the matrix instead of its first n rows: the reads a narrow pass
// hands back are classified where they lie (no dense copy of their rows); everything a read owns -- class byte, ns, ee, pass -- is
// indexed by the read, so the class-byte array must hold the whole matrix' reads; only the block histograms go by position.
template <bool RAGGED, bool LONG, bool LISTED>
__global__ __launch_bounds__(256) void k_prepass(const uint8_t *__restrict__ q, int64_t n,
                                                 int64_t stride, const int32_t *__restrict__ len_arg,
                                                 MpbDevParams prm, PreOut o, int32_t *__restrict__ blockhist,
                                                 const int32_t *__restrict__ list)
{
    __shared__ float2 s_tab[256];
    __shared__ float4 s_row[4][64];               // per read: {mu, var, k3, ambiguity counts (bits of an int: N | n << 16)}
    __shared__ int s_hist[RAGGED ? MPB_SKEYS : MPB_NCLS];
    const int32_t *__restrict__ len = RAGGED ? len_arg : nullptr;
    constexpr int nb = RAGGED ? MPB_LEN_BINS : 1;   // length bins of the sort key (one fixed length: one bin)
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    {
        const bool amb = tid == 0 || tid == 255;
        float p = __builtin_amdgcn_exp2f(-0.33219281f * (float)tid);      // 10^(-q/10)
        p = amb ? 0.0f : p;
        s_tab[tid] = make_float2(p, tid == 0 ? MPB_MARK_UPPER : tid == 255 ? MPB_MARK_LOWER : p * (1.0f - p));
    }
    for (int k = tid; k < MPB_NCLS * nb; k += 256) s_hist[k] = 0;
    __syncthreads();

    const int r16 = lane & 15, cl = lane >> 4;
    for (int round = 0; round < MPB_PRE_ROUNDS; round++) {
    const int64_t wave_row0 = (int64_t)blockIdx.x * MPB_PRE_READS + round * 256 + w * 64;
    if (wave_row0 >= n) break;                    // wave-uniform; nothing below is a block barrier

    for (int rb = 0; rb < 64; rb += 16) {
        const bool live = wave_row0 + rb + r16 < n;
        const int64_t i = LISTED ? (live ? (int64_t)list[wave_row0 + rb + r16] : 0) : wave_row0 + rb + r16;
        const int li = live ? (len ? clamp_len(len[i], prm.max_len) : prm.fixed_len) : 0;
        int ncol = (li + 15) >> 4;                // chunk columns to walk: the longest read of the 16
        int nfull = live ? (li >> 4) : (1 << 20); // columns complete in every live lane
        if (len || wave_row0 + 64 > n) {
#pragma unroll
            for (int off = 8; off >= 1; off >>= 1) {
                ncol = max(ncol, __shfl_xor(ncol, off));
                nfull = min(nfull, __shfl_xor(nfull, off));
            }
        }
        ncol = __builtin_amdgcn_readfirstlane(ncol);
        nfull = __builtin_amdgcn_readfirstlane(nfull);
        const uint8_t *src = q + (live ? i : (int64_t)0) * stride + cl * 16;
        float mu = 0.f, var = 0.f, k3 = 0.f;
        uint32_t ambi = 0;                         // 'N' count | 'n' count << 16 (unsigned: up to 65535 of either)
        // A panel = 3 * MPB_PRE_NB column quads = at most 240 bytes per lane, so that the ambiguity markers come off
        // the panel's sum exactly.  Rows of up to 960 bytes are one panel (and one set of sums, as before).
        auto panel = [&](const int pb, const int pend) {
        f32x2 a01 = {0.f, 0.f};
        float s3 = 0.f;
        // MPB_PRE_NB column quads at a time: all their loads are issued before the first byte is
        // looked at (a row of 300 bases is one such batch), so a wave keeps 5 KiB in flight
        for (int cb = pb; cb < pend; cb += 4 * MPB_PRE_NB) {
            uint4 xs[MPB_PRE_NB];
#pragma unroll
            for (int b = 0; b < MPB_PRE_NB; b++) {
                const int c0 = cb + 4 * b;
                xs[b] = make_uint4(0, 0, 0, 0);
                if (c0 < pend && li > (c0 + cl) * 16)
                    xs[b] = *reinterpret_cast<const uint4 *>(src + c0 * 16);
            }
#pragma unroll
            for (int b = 0; b < MPB_PRE_NB; b++) {
                const int c0 = cb + 4 * b;
                if (c0 >= pend) break;                     // wave-uniform
                uint4 y = xs[b];
                if (c0 + 4 > nfull) {                      // wave-uniform: ragged tail, fill past the end
                    const int nv = li - (c0 + cl) * 16;
                    y.x = fill_dword(y.x, nv); y.y = fill_dword(y.y, nv - 4);
                    y.z = fill_dword(y.z, nv - 8); y.w = fill_dword(y.w, nv - 12);
                }
                pre_chunk(s_tab, y, a01, s3);
            }
        }
        // second component = sum p(1-p) (<= 64) + 128 per 'N' + 65536 per 'n': peel the markers
        const float n255 = floorf(a01.y * (1.0f / MPB_MARK_LOWER));
        const float rem = a01.y - MPB_MARK_LOWER * n255;
        const float nzero = floorf(rem * (1.0f / MPB_MARK_UPPER));
        const float pvar = rem - MPB_MARK_UPPER * nzero;
        mu += a01.x;
        var += pvar;
        k3 += pvar - 2.0f * s3;                                           // sum p(1-p)(1-2p)
        ambi += (uint32_t)nzero + ((uint32_t)n255 << 16);
        };  // panel
        if (LONG) { for (int pb = 0; pb < ncol; pb += 12 * MPB_PRE_NB) panel(pb, min(ncol, pb + 12 * MPB_PRE_NB)); }
        else panel(0, ncol);
        // the four lanes of a read, combined in a fixed order: (cl0 + cl1) + (cl2 + cl3)
#pragma unroll
        for (int off = 16; off <= 32; off <<= 1) {
            mu += __shfl_xor(mu, off);
            var += __shfl_xor(var, off);
            k3 += __shfl_xor(k3, off);
            ambi += __shfl_xor(ambi, off);
        }
        const float amb = __uint_as_float(ambi);
        if (cl == 0) s_row[w][rb + r16] = make_float4(mu, var, k3, amb);
    }
    wave_lds_fence();          // s_row[w] is private to this wave: no block barrier
    // ---- classing: one lane per read, all 64 lanes busy (was: 16 of 64, four times) ----
    {
        const int64_t k = wave_row0 + lane;
        if (k < n) {
            const int64_t i = LISTED ? (int64_t)list[k] : k;
            const float4 e = s_row[w][lane];
            const int li = len ? clamp_len(len[i], prm.max_len) : prm.fixed_len;
            class_read(i, e.x, e.y, e.z, __float_as_uint(e.w), li, RAGGED && li != len[i], RAGGED, prm, o, s_hist, nb);
        }
    }
    wave_lds_fence();                             // s_row[w] is reused by the next round
    }   // rounds
    __syncthreads();
    for (int k = tid; k < MPB_NCLS * nb; k += 256) blockhist[(int64_t)k * gridDim.x + blockIdx.x] = s_hist[k];   // key-major
}

// ------------------------------------------------------------------------------------------
// k_classify_linear: the same classification, walking the matrix LINEARLY.
// k_prepass gives a lane one row (16 rows x 64 bytes per load instruction) so that a read's sums stay in registers.  When
// the pass also has to move text in and the packed matrix out (classified at source: three streams), that access
// shape, not the arithmetic, is what limits it.  Here thread t of a block takes the 16-byte chunks t, t + 256, ... of a
// tile of whole rows -- consecutive lanes read and write consecutive addresses, exactly as the plain decode kernel does --
// peels its chunk's ambiguity markers at once (16 bytes: always exact), leaves the chunk's partial sums in LDS, and the
// rows of the tile are then summed from LDS in a fixed order (deterministic) by a few lanes each and classified.
// Block b still owns reads [1024 b, 1024 b + 1024): the histograms keep the layout k_scan / k_scatter expect.
// ------------------------------------------------------------------------------------------
#define MPB_LIN_K 5                               // chunks per thread and tile, all loaded before the first is used
#define MPB_LIN_CHUNKS (256 * MPB_LIN_K)

template <bool RAGGED, bool DECODE>
__global__ __launch_bounds__(256) void k_classify_linear(const uint8_t *__restrict__ q, int64_t n, int64_t stride,
                                                         const int32_t *__restrict__ len_arg, MpbDevParams prm, PreOut o,
                                                         int32_t *__restrict__ blockhist, PreDecode dec, uint32_t cpr_magic)
{
    __shared__ float2 s_tab[256];
    __shared__ float4 s_part[MPB_LIN_CHUNKS];     // per chunk: {sum p, sum p(1-p), sum p(1-p)(1-2p), ambiguity counts}
    __shared__ int s_len[MPB_PRE_READS];          // clamped length of each row of the tile
    __shared__ int s_hist[RAGGED ? MPB_SKEYS : MPB_NCLS];
    const int32_t *__restrict__ len = RAGGED ? len_arg : nullptr;
    constexpr int nb = RAGGED ? MPB_LEN_BINS : 1;
    const int tid = threadIdx.x;
    {
        const bool amb = tid == 0 || tid == 255;
        float p = __builtin_amdgcn_exp2f(-0.33219281f * (float)tid);      // 10^(-q/10)
        p = amb ? 0.0f : p;
        s_tab[tid] = make_float2(p, tid == 0 ? MPB_MARK_UPPER : tid == 255 ? MPB_MARK_LOWER : p * (1.0f - p));
    }
    for (int k = tid; k < MPB_NCLS * nb; k += 256) s_hist[k] = 0;
    const int cpr = (int)(stride >> 4);                                   // chunks per row (<= 1024)
    const int tile_rows = max(1, min(MPB_PRE_READS, MPB_LIN_CHUNKS / cpr));
    const int64_t first = (int64_t)blockIdx.x * MPB_PRE_READS;
    const int64_t last = min(n, first + MPB_PRE_READS);
    // lanes that share a row in the reduction: a power of two, at most 64, as many as the block has to spare
    int lpr = 1;
    while (lpr < 64 && lpr * 2 * tile_rows <= 256) lpr *= 2;
    int badq = 0;
    for (int64_t row0 = first; row0 < last; row0 += tile_rows) {
        const int nr = (int)min((int64_t)tile_rows, last - row0);
        const int nchunks = nr * cpr;
        __syncthreads();                                                  // s_len / s_part of the tile before (and the tables above)
        for (int t = tid; t < nr; t += 256) s_len[t] = len ? clamp_len(len[row0 + t], prm.max_len) : prm.fixed_len;
        __syncthreads();
        const int64_t base = row0 * stride;
        // ---- all loads of the tile first ----
        uint4 xs[MPB_LIN_K], ys[MPB_LIN_K];
        int nvs[MPB_LIN_K];
#pragma unroll
        for (int k = 0; k < MPB_LIN_K; k++) {
            const int c = tid + k * 256;
            xs[k] = ys[k] = make_uint4(0, 0, 0, 0);
            nvs[k] = -1;                                                  // -1: no such chunk in this tile
            if (c < nchunks) {
                const int r = cpr_magic ? (int)__umulhi((uint32_t)c, cpr_magic) : c;   // c / cpr (magic 0: cpr == 1), exact for c < 2^16
                const int col = c - r * cpr;
                const int nv = s_len[r] - col * 16;                       // bases of the read in this chunk (may be <= 0)
                nvs[k] = max(nv, 0);
                if (nv > 0) {
                    if (DECODE) {
                        xs[k] = *reinterpret_cast<const uint4 *>(dec.seq + base + (int64_t)c * 16);
                        ys[k] = *reinterpret_cast<const uint4 *>(dec.qual + base + (int64_t)c * 16);
                    } else {
                        xs[k] = *reinterpret_cast<const uint4 *>(q + base + (int64_t)c * 16);
                    }
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- decode (and write the packed chunk), sum, peel, park the partials ----
#pragma unroll
        for (int k = 0; k < MPB_LIN_K; k++) {
            const int c = tid + k * 256;
            const int nv = nvs[k];
            if (nv < 0) continue;
            uint4 y = xs[k];
            if (DECODE) {
                if (nv > 0) y = decode16(xs[k], ys[k], 0, nv, dec.offset, badq);
                *reinterpret_cast<uint4 *>(dec.out + base + (int64_t)c * 16) = y;     // zeros past the read's end
            }
            float4 part = make_float4(0.f, 0.f, 0.f, __int_as_float(0));
            if (nv > 0) {
                if (nv < 16) {
                    y.x = fill_dword(y.x, nv); y.y = fill_dword(y.y, nv - 4);
                    y.z = fill_dword(y.z, nv - 8); y.w = fill_dword(y.w, nv - 12);
                }
                f32x2 a01 = {0.f, 0.f};
                float s3 = 0.f;
                pre_chunk(s_tab, y, a01, s3);
                const float n255 = floorf(a01.y * (1.0f / MPB_MARK_LOWER));
                const float rem = a01.y - MPB_MARK_LOWER * n255;
                const float nzero = floorf(rem * (1.0f / MPB_MARK_UPPER));
                const float pvar = rem - MPB_MARK_UPPER * nzero;
                part = make_float4(a01.x, pvar, pvar - 2.0f * s3, __uint_as_float((uint32_t)nzero + ((uint32_t)n255 << 16)));
            }
            s_part[c] = part;
        }
        __syncthreads();
        // ---- rows of the tile: lpr lanes each, chunk partials in a fixed order ----
        const int sub = tid & (lpr - 1);
        for (int r = tid / lpr; r < nr; r += 256 / lpr) {                 // (256 / lpr) rows per sweep; wave-uniform trip count not needed
            float mu = 0.f, var = 0.f, k3 = 0.f;
            uint32_t ambi = 0;
            for (int j = sub; j < cpr; j += lpr) {
                const float4 e = s_part[r * cpr + j];
                mu += e.x; var += e.y; k3 += e.z; ambi += __float_as_uint(e.w);
            }
            for (int off = 1; off < lpr; off <<= 1) {
                mu += __shfl_xor(mu, off); var += __shfl_xor(var, off);
                k3 += __shfl_xor(k3, off); ambi += __shfl_xor(ambi, off);
            }
            if (sub == 0) {
                const int64_t i = row0 + r;
                const int li = s_len[r];
                class_read(i, mu, var, k3, ambi, li, RAGGED && li != len[i], RAGGED, prm, o, s_hist, nb);
            }
        }
    }
    if (DECODE && badq && dec.err) atomicAdd(dec.err, badq);
    __syncthreads();
    for (int k = tid; k < MPB_NCLS * nb; k += 256) blockhist[(int64_t)k * gridDim.x + blockIdx.x] = s_hist[k];   // key-major
}

// ------------------------------------------------------------------------------------------
// k_scan: block k turns blockhist[k][.] (one sort key) into exclusive prefixes and writes kcount[k]
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_scan(int32_t *__restrict__ blockhist, int nblocks,
                                              MpbTables *__restrict__ tb)
{
    __shared__ int s_part[256];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int seg = (nblocks + 255) / 256;
    const int b0 = tid * seg, b1 = min(nblocks, b0 + seg);
    int sum = 0;
    int32_t *bh = blockhist + (int64_t)c * nblocks;     // this key's row (contiguous)
    for (int b = b0; b < b1; b++) sum += bh[b];
    s_part[tid] = sum;
    __syncthreads();
    // Hillis-Steele inclusive scan over 256 partials
    for (int off = 1; off < 256; off <<= 1) {
        int v = (tid >= off) ? s_part[tid - off] : 0;
        __syncthreads();
        s_part[tid] += v;
        __syncthreads();
    }
    int run = s_part[tid] - sum;   // exclusive prefix of this thread's segment
    for (int b = b0; b < b1; b++) {
        const int h = bh[b];
        bh[b] = run;
        run += h;
    }
    if (tid == 255) tb->kcount[c] = s_part[255];
}

// One block of 512 threads (round 6: the single-thread form walked the 512 keys of a ragged batch through dependent global loads and
// stores, 38 us; this one 5): the keys' counts come into LDS, a thread per class sums its bins and lays out its keys, one thread
// lays out the classes (perm slots padded to 64, tiles widest class first).
__global__ __launch_bounds__(512) void k_tables(MpbTables *__restrict__ tb, int nb, int32_t *__restrict__ ovf_count,
                                                unsigned long long *__restrict__ pass_count)
{
    __shared__ int s_k[MPB_SKEYS], s_cnt[MPB_NCLS], s_base[MPB_NCLS + 1], s_tile[MPB_NCLS + 1];
    const int tid = threadIdx.x;
    for (int k = tid; k < MPB_NCLS * nb; k += 512) s_k[k] = tb->kcount[k];
    __syncthreads();
    if (tid < MPB_NCLS) {
        int cnt = 0;
        for (int b = 0; b < nb; b++) cnt += s_k[tid * nb + b];
        s_cnt[tid] = cnt;
    }
    __syncthreads();
    if (tid == 0) {
        int base = 0;
        for (int c = 0; c < MPB_NCLS; c++) { s_base[c] = base; base += (s_cnt[c] + 63) & ~63; }
        s_base[MPB_NCLS] = base;
        int t = 0;
        for (int c = MPB_NCLS - 1; c >= 0; c--) {        // widest (most expensive) tiles first
            const int rpt = 64 / c_classes[c].G;
            s_tile[c] = t;
            t += (s_cnt[c] + rpt - 1) / rpt;
        }
        s_tile[MPB_NCLS] = t;
    }
    __syncthreads();
    if (tid < MPB_NCLS) {
        const int c = tid;
        tb->perm_base[c] = s_base[c];
        tb->count[c] = s_cnt[c];
        tb->tile_start[c] = s_tile[c];
        int run = s_base[c];
        for (int b = 0; b < nb; b++) {                   // a class's keys are consecutive, shortest reads first
            tb->key_base[c * nb + b] = run;
            run += s_k[c * nb + b];
        }
    }
    if (tid == 0) {
        tb->perm_base[MPB_NCLS] = s_base[MPB_NCLS];
        tb->tile_start[MPB_NCLS] = s_tile[MPB_NCLS];
        tb->total_tiles = s_tile[MPB_NCLS];
        *ovf_count = 0;
        *pass_count = 0ull;
    }
}

// overflow pass: every listed read goes to one class `wc`
__global__ void k_tables_overflow(MpbTables *__restrict__ tb, const int32_t *__restrict__ ovf_count,
                                  int wc, long long *__restrict__ ovf_total)
{
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int cnt = *ovf_count;
    *ovf_total += cnt;
    const int rpt = 64 / c_classes[wc].G;
    // class c owns tiles [tile_start[c], tile_start[c] + tiles(count[c])); only wc is non-empty
    for (int c = 0; c <= MPB_NCLS; c++) {
        if (c < MPB_NCLS) tb->count[c] = (c == wc) ? cnt : 0;
        tb->perm_base[c] = 0;
        tb->tile_start[c] = 0;
    }
    tb->total_tiles = (cnt + rpt - 1) / rpt;
}

// ------------------------------------------------------------------------------------------
// k_scatter: stable scatter of read indices into perm[], grouped by class
// ------------------------------------------------------------------------------------------
// Key = (class, length bin), see MPB_SKEYS.  Position = start of the key in perm[] + reads of the key
// in earlier blocks (the scanned histogram) + rank inside the block, taken from ballots in input
// order: no atomics, deterministic perm.
template <bool RAGGED>
__global__ __launch_bounds__(256) void k_scatter(const uint8_t *__restrict__ cls, int64_t n,
                                                 const int32_t *__restrict__ len_arg, int max_len, int len_shift,
                                                 const int32_t *__restrict__ blockhist,
                                                 const MpbTables *__restrict__ tb,
                                                 const int32_t *__restrict__ ns,
                                                 int32_t *__restrict__ perm, uint16_t *__restrict__ perm_ns,
                                                 const int32_t *__restrict__ list)      // (the pass runs over list[0 .. n): k_prepass<.., LISTED>)
{
    constexpr int nb = RAGGED ? MPB_LEN_BINS : 1;
    constexpr int nkeys = MPB_NCLS * nb;
    __shared__ int s_wcnt[4][nkeys];              // per wave and round: reads of each key
    __shared__ int s_base[nkeys];                 // next free slot of each key for this block
    const int32_t *__restrict__ len = RAGGED ? len_arg : nullptr;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int k = tid; k < nkeys; k += 256) s_base[k] = tb->key_base[k] + blockhist[(int64_t)k * gridDim.x + blockIdx.x];
    // everything this thread needs from global memory for all its rounds, requested up front: the rounds
    // themselves are then LDS / ballot work only (they used to pay two dependent HBM round trips each)
    int kk_r[MPB_PRE_ROUNDS], ns_r[MPB_PRE_ROUNDS], c_r[MPB_PRE_ROUNDS], len_r[MPB_PRE_ROUNDS], id_r[MPB_PRE_ROUNDS];
#pragma unroll
    for (int round = 0; round < MPB_PRE_ROUNDS; round++) {      // independent loads: all in flight together
        const int64_t k = (int64_t)blockIdx.x * MPB_PRE_READS + round * 256 + tid;
        const int64_t i = (list && k < n) ? (int64_t)list[k] : k;
        id_r[round] = (int)i;
        c_r[round] = k < n ? (int)cls[i] : 0xff;
        ns_r[round] = k < n ? ns[i] : 0;
        len_r[round] = (len && k < n) ? len[i] : 0;
    }
#pragma unroll
    for (int round = 0; round < MPB_PRE_ROUNDS; round++) {
        int c = c_r[round] == 0xff ? -1 : (c_r[round] & 0x7f);
        if (c >= MPB_NCLS) c = -1;                // settled by the prepass, or wide (k_wide's list): not part of any tile
        kk_r[round] = c < 0 ? -1 : c * nb + (len ? min(MPB_LEN_BINS - 1, clamp_len(len_r[round], max_len) >> len_shift) : 0);
    }
#pragma unroll
    for (int round = 0; round < MPB_PRE_ROUNDS; round++) {
        for (int k = tid; k < 4 * nkeys; k += 256) (&s_wcnt[0][0])[k] = 0;
        __syncthreads();
        const int kk = kk_r[round];
        int rank = 0;
        unsigned long long remaining = __ballot(kk >= 0);
        while (remaining) {
            const int leader = __ffsll((long long)remaining) - 1;
            const int cc = __shfl(kk, leader);
            const unsigned long long m = __ballot(kk == cc);
            if (kk == cc) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == leader) s_wcnt[w][cc] = __popcll(m);
            remaining &= ~m;
        }
        __syncthreads();
        if (kk >= 0) {
            int off = s_base[kk];
            for (int ww = 0; ww < w; ww++) off += s_wcnt[ww][kk];
            perm[off + rank] = (int32_t)id_r[round];
            perm_ns[off + rank] = (uint16_t)ns_r[round];
        }
        __syncthreads();
        for (int k = tid; k < nkeys; k += 256)
            s_base[k] += s_wcnt[0][k] + s_wcnt[1][k] + s_wcnt[2][k] + s_wcnt[3][k];
        __syncthreads();                          // before the next round clears s_wcnt
    }
}

// ------------------------------------------------------------------------------------------
// k_dp
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ double dpp_prev_row(double x, int keep)
{
    // value of lane-1 (wave_shr:1, lane 0 reads 0 through bound_ctrl), ANDed with `keep`
    // (0 in the first lane of every read, ~0 elsewhere).  Written as ONE 64-bit shift + mask so that
    // the DPP-combine pass folds both halves into a v_and_b32_dpp each (the two-halves spelling left
    // a v_mov_b32_dpp + v_and_b32 pair for the low half).
    const long long k64 = ((long long)keep << 32) | (unsigned int)keep;
    const long long y = __builtin_amdgcn_update_dpp(0ll, __double_as_longlong(x), 0x138, 0xf, 0xf, true) & k64;
    return __longlong_as_double(y);
}

template <bool FMA>
__device__ __forceinline__ double cell(double a, double v, double b, double w)
{
    if (FMA) return __builtin_fma(a, v, b * w);
    const double x = a * v;      // fl(a*v)
    const double y = b * w;      // fl(b*w)
    return x + y;                // fl(x+y): three roundings, as the reference
}

// one base: advance the running vector by (a,b)
template <int R, int G, bool FMA>
__device__ __forceinline__ void dp_step(double (&v)[R], const double2 ab, const int keep)
{
    double cin = 0.0;
    if (G > 1) cin = dpp_prev_row(v[R - 1], keep);
#pragma unroll
    for (int r = R - 1; r >= 1; r--) v[r] = cell<FMA>(ab.x, v[r], ab.y, v[r - 1]);
    if (G > 1) v[0] = cell<FMA>(ab.x, v[0], ab.y, cin);
    else v[0] = ab.x * v[0];
}

template <int R, int G, bool FMA>
__device__ __forceinline__ void dp_dword(double (&v)[R], uint32_t w, int keep)
{
#pragma unroll
    for (int t = 0; t < 4; t++) dp_step<R, G, FMA>(v, mpb_s_lut[(w >> (8 * t)) & 0xffu], keep);
}

// 16 bases.  Narrow bodies are unrolled completely; wide ones loop over the 4 dwords so that the
// LUT entries in flight (registers) and the code size stay bounded.
template <int R, int G, bool FMA>
__device__ __forceinline__ void dp_chunk_compact(double (&v)[R], const uint4 x, int keep)
{
    uint32_t w0 = x.x, w1 = x.y, w2 = x.z, w3 = x.w;
#pragma unroll 1
    for (int d = 0; d < 2; d++) {          // 8 bases per trip
        dp_dword<R, G, FMA>(v, w0, keep);
        dp_dword<R, G, FMA>(v, w1, keep);
        w0 = w2; w1 = w3;
    }
}

// up to 16 bases of a tail chunk: `ndw` (wave-uniform, 1..4) dwords, 8 bases per trip while two dwords remain
template <int R, int G, bool FMA>
__device__ __forceinline__ void dp_chunk_tail(double (&v)[R], const uint4 x, int keep, int ndw)
{
    uint32_t w0 = x.x, w1 = x.y, w2 = x.z, w3 = x.w;
    int d = 0;
#pragma unroll 1
    for (; d + 2 <= ndw; d += 2) {
        dp_dword<R, G, FMA>(v, w0, keep);
        dp_dword<R, G, FMA>(v, w1, keep);
        w0 = w2; w1 = w3;
    }
    if (d < ndw) dp_dword<R, G, FMA>(v, w0, keep);
}

template <int R, int G, bool FMA>
__device__ __forceinline__ void dp_chunk(double (&v)[R], const uint4 x, int keep)
{
    if (R <= 8) {
        dp_dword<R, G, FMA>(v, x.x, keep);
        dp_dword<R, G, FMA>(v, x.y, keep);
        dp_dword<R, G, FMA>(v, x.z, keep);
        dp_dword<R, G, FMA>(v, x.w, keep);
    } else {
        dp_chunk_compact<R, G, FMA>(v, x, keep);
    }
}

// Register budget of the DP kernel: 4 waves per SIMD = at most 128 VGPRs per lane.
#define MPB_DP_WAVES_PER_EU 4

struct DpArgs {
    const uint8_t *q;
    int64_t stride;
    const int32_t *len;
    const int32_t *ns;
    const uint8_t *cls;
    double *ee;
    uint8_t *pass;
    int32_t *ovf_list;
    int32_t *ovf_count;
    unsigned long long *alg_cells;   // MPB_FLAG_COUNT_CELLS: device total of sum_k min(k+1, J) over the reads reported (k_dp)
    const int32_t *perm;        // with perm_ns: the sorted index array the tiles walk (main pass only)
    const uint16_t *perm_ns;    // ns of perm[k]'s read, or nullptr: gather ns[idx]
    MpbDevParams prm;
    int final_pass;
};

// A run of consecutive tiles of one class.  Deliberately NOT inlined: each (R,G) body gets its own
// register allocation, so the kernel's VGPR budget is the widest body's, not the sum of all of them.
// The call saves the callee-saved VGPRs it uses to scratch; taking a run of tiles per call keeps
// that traffic negligible.
template <int R, int G, bool FMA>
__device__ __noinline__ void dp_tiles(const DpArgs *__restrict__ Ap, const int32_t *perm_cls,
                                      int count, int first_tile, int n_tiles)
{
    const DpArgs &A = *Ap;
    constexpr int RPT = 64 / G;
    const int lane = lane_id();
    const int lig = lane & (G - 1);
    const bool leader = lig == 0;
    int keep = leader ? 0 : -1;
    asm volatile("" : "+v"(keep));        // opaque: keeps `& keep` a v_and (foldable into the DPP op), not a select
#pragma unroll 1
    for (int local_tile = first_tile; local_tile < first_tile + n_tiles; local_tile++) {
    const int slot = local_tile * RPT + lane / G;
    const bool valid = slot < count;
    const int idx = gload(perm_cls + (valid ? slot : count - 1));
    const int li = A.len ? clamp_len(gload(A.len + idx), A.prm.max_len) : A.prm.fixed_len;
    const uint8_t *row = A.q + (int64_t)idx * A.stride;

    int limax = li;                       // the longest read of the tile
    int nfull = li >> 4;                  // chunks that are complete in EVERY lane: no masking needed
    if (A.len) {                          // (fixed-length batches: every lane has the same li)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            limax = max(limax, __shfl_xor(limax, off));
            nfull = min(nfull, __shfl_xor(nfull, off));
        }
    }
    limax = __builtin_amdgcn_readfirstlane(limax);  // wave-uniform trip counts
    nfull = __builtin_amdgcn_readfirstlane(nfull);
    const int nch = (limax + 15) >> 4;    // 16-byte chunks to walk

    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = 0.0;
    if (leader) v[0] = 1.0;

    // Each lane pulls its row 64 bytes at a time (4 x dwordx4, issued together, one 64-byte
    // segment of one line) and one super-chunk ahead of the arithmetic, so a cache line is
    // consumed while it is still resident instead of being re-fetched 16 bytes at a time.
    // A load is guarded by a WAVE-UNIFORM test only (the chunk lies inside the row: rows are `stride`
    // bytes whatever the read's length, and bytes past a read's end are masked below), so the
    // prefetch costs one 64-bit pointer bump per 64 bases instead of a compare / exec-mask / zero-fill
    // sequence per 16.
    const int row_chunks = __builtin_amdgcn_readfirstlane((int)(A.stride >> 4));
    const int nsc = (nch + 3) >> 2;                // wave-uniform 64-byte super-chunks
    const int nsc_fast = nfull >> 2;               // ... of which this many hold 64 valid bases in EVERY lane (all inside the row)
    uint4 cur[4], nxt[4];
#pragma unroll
    for (int p = 0; p < 4; p++) cur[p] = nxt[p] = make_uint4(0, 0, 0, 0);
    if (nsc_fast > 0) {
#pragma unroll
        for (int p = 0; p < 4; p++) cur[p] = gload16(row + p * 16);
        // Exactly four loads per trip, whatever the trip, issued BEFORE the trip's arithmetic and first
        // waited for at the top of the next trip: a whole super-chunk of FP64 work (1.5-6 us) covers the
        // memory latency.  The last trip has nothing new to fetch when the row ends here; it re-reads its
        // own (cache-resident) 64 bytes instead of branching, because a path-dependent load count makes
        // the compiler's wait-count bookkeeping fall back to "wait for everything" at once.
        // (Tried and rejected, bit-exact both: re-loading two chunks at a time into the registers just
        // consumed -- no second register set, no copies, but half the prefetch distance: k_dp +2 %.)
        for (int sc = 0; sc < nsc_fast; sc++) {
            const int nxt_sc = (sc * 4 + 8 <= row_chunks) ? sc + 1 : sc;     // scalar select
            const uint8_t *pf = row + (nxt_sc << 6);
#pragma unroll
            for (int p = 0; p < 4; p++) nxt[p] = gload16(pf + p * 16);
            // the machine scheduler otherwise sinks these loads to the end of the trip (their registers are
            // free there) and the next trip opens with vmcnt(0): the whole memory latency exposed
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int p = 0; p < 4; p++) dp_chunk<R, G, FMA>(v, cur[p], keep);
#pragma unroll
            for (int p = 0; p < 4; p++) cur[p] = nxt[p];
        }
    }
    // tail: the super-chunks that are ragged in some lane (a 300-base read: bases 256..299)
    for (int sc = nsc_fast; sc < nsc; sc++) {
        if (!(sc > 0 && sc == nsc_fast && sc * 4 + 4 <= row_chunks)) {      // not already fetched by the loop above
#pragma unroll
            for (int p = 0; p < 4; p++) {
                cur[p] = make_uint4(0, 0, 0, 0);
                if (sc * 4 + p < row_chunks) cur[p] = gload16(row + (sc * 4 + p) * 16);   // wave-uniform guard
            }
        }
        const int pmax = min(4, nch - sc * 4);     // wave-uniform
#pragma unroll 1
        for (int p = 0; p < pmax; p++) {
            uint4 x = cur[0];
            cur[0] = cur[1]; cur[1] = cur[2]; cur[2] = cur[3];   // rotate: keeps every index static
            const int c = sc * 4 + p;
            int ndw = 4;                                 // dwords of this chunk to walk: wave-uniform
            if (c >= nfull) {                            // wave-uniform: only ragged tail chunks are masked
                const int nv = li - c * 16;              // may be <= 0 for reads shorter than the tile's longest
                x.x = mask_dword(x.x, nv); x.y = mask_dword(x.y, nv - 4);
                x.z = mask_dword(x.z, nv - 8); x.w = mask_dword(x.w, nv - 12);
                ndw = min(4, (limax - c * 16 + 3) >> 2); // a 300-base read ends 12 bases into its last chunk: 3 dwords, not 4
            }
            dp_chunk_tail<R, G, FMA>(v, x, keep, ndw);
        }
    }

    // ---- epilogue: sequential CDF (ref: bernoullimodule.c:233-251), first row above thr ----
    const double thr = A.prm.thr;
    double lo = 0.0, hi = 0.0;
    int js = -1;
    bool writer;                                   // the lane that reports this read
    bool never_crossed = false;
    if (G == 1) {
        double acc = 0.0;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double na = acc + v[r];          // r == 0: 0 + v0 is exact
            const bool hit = (js < 0) && (na > thr);
            lo = hit ? acc : lo;
            hi = hit ? na : hi;
            js = hit ? r : js;
            acc = na;
        }
        writer = valid;
        never_crossed = js < 0;
    } else {
        // Phase A: the running sum visits the G lanes of a read in order (same additions, same
        // order as the reference); a lane only notes whether the crossing falls inside its rows.
        // The CDF never decreases, so that is "sum after my rows > thr and nobody before me".
        double acc = 0.0, acc_in_mine = 0.0;
        int found = 0;
        bool mine = false;
#pragma unroll 1
        for (int g = 0; g < G; g++) {
            double acc_s = acc;
            int found_s = found;
            if (g > 0) {
                const int src = (lane & ~(G - 1)) + g - 1;
                acc_s = __shfl(acc, src);
                found_s = __shfl(found, src);
            }
            if (lig == g) {
                double a = acc_s;
#pragma unroll
                for (int r = 0; r < R; r++) a = a + v[r];
                const bool cross = !found_s && (a > thr);
                mine = cross;
                acc_in_mine = acc_s;
                acc = a;
                found = found_s | (cross ? 1 : 0);
            }
        }
        // Phase B: only the crossing lane walks its rows again to pick out the two CDF values.
        if (mine) {
            double a = acc_in_mine;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double na = a + v[r];
                const bool hit = (js < 0) && (na > thr);
                lo = hit ? a : lo;
                hi = hit ? na : hi;
                js = hit ? (lig * R + r) : js;
                a = na;
            }
        }
        never_crossed = (lig == G - 1) && !found;  // the last lane has seen the whole CDF
        writer = valid && (mine || never_crossed);
    }
    if (writer) {
        if (never_crossed && A.final_pass == 0) {
            const int pos = atomicAdd(A.ovf_count, 1);
            A.ovf_list[pos] = idx;
        } else if (never_crossed && A.final_pass == 2) {
            A.pass[idx] = 2;                                   // small-batch path: the host re-runs the batch
        } else {
            double e;
            if (never_crossed) {
                e = __builtin_nan("");                 // CDF never crosses: reference runs off its table
            } else {
                // ref: bernoullimodule.c:170-178  errors1 + ((errors2-errors1)*((1-alpha)-prob1)/(prob2-prob1))
                e = (double)(js - 1) + ((thr - lo) / (hi - lo));
                if (e < 0) e = 0;
            }
            const int nsv = A.perm_ns ? (int)gload(A.perm_ns + (perm_cls - A.perm) + slot) : gload(A.ns + idx);
            if ((A.prm.flags & 32u) && !never_crossed) {                 // MPB_FLAG_COUNT_CELLS (diagnostic, off by default)
                // the table the reference fills for this read: rows 0..js over the L' = len - Ns scored bases, of which
                // row j is non-zero from base j on: sum_k min(k + 1, J), J = js + 1 (SURVEY 8d "algorithmic flops")
                const int J = js + 1, Lp = li - nsv;                 // J <= 1024, Lp <= 16383: 32-bit arithmetic is enough
                const int cells = J <= Lp ? ((J * (J + 1)) >> 1) + (Lp - J) * J : (Lp * (Lp + 1)) >> 1;
                atomicAdd(&mpb_s_cells, (unsigned long long)(unsigned int)cells);
            }
            if (A.prm.ambig_mode == 0) e = e + (double)nsv;              // moira.py:827-828
            const double limit = (A.prm.maxerrors == A.prm.maxerrors) ? A.prm.maxerrors          // moira.py:925-926
                                                                      : (double)li * A.prm.uncert; // moira.py:949-950
            if (FMA && A.final_pass != 1 && !never_crossed) {
                // MPB_FLAG_FAST_FMA keeps the DECISIONS exact: an ee that lands within 1e-9 relative of the
                // threshold (or, with --round, of an integer) is not trusted -- the read goes to the second
                // pass, which always runs the three-rounding arithmetic
                const double tol = 1e-9 * fmax(1.0, fabs(e));
                bool unsure = fabs(e - limit) <= tol;
                if (A.prm.flags & 1u) unsure = unsure || fabs(e - rint(e)) <= tol;
                if (unsure) {
                    if (A.final_pass == 0) { const int pos = atomicAdd(A.ovf_count, 1); A.ovf_list[pos] = idx; }
                    else A.pass[idx] = 2;
                    continue;
                }
            }
            if (A.prm.flags & 1u) e = floor(e);                          // moira.py:830-831
            bool keep_read;
            if (A.prm.ambig_mode == 2 && (gload(A.cls + idx) & 0x80)) keep_read = false;    // moira.py:911
            else keep_read = e <= limit;
            gstore(A.ee + idx, e);
            gstore(A.pass + idx, (uint8_t)(keep_read ? 1 : 0));
        }
    }
    }   // tiles of this run
}

// DpArgs arrive as a kernel argument and are parked in LDS, so that the non-inlined class bodies can
// take a pointer to them (no extra launch to put them into device memory)

// One wave takes `chunk_tiles` consecutive tiles (almost always one class: one call, one save of
// callee-saved VGPRs); chunks are dealt to blocks by the hardware dispatcher, which balances the very
// different tile costs dynamically.  Tiles are ordered widest class first = longest first, so the tail
// is made of the cheapest tiles.  The chunk loop is grid-strided because the host sizes the grid
// from an upper bound that assumes one read per lane; every wave's loop ends when its chunk index
// passes the tile count.
template <bool FMA, bool OVERFLOW_PASS>
__global__ __launch_bounds__(256, MPB_DP_WAVES_PER_EU) void k_dp(DpArgs args,
                                            const double2 *__restrict__ lut_g,
                                            MpbTables *__restrict__ tb,
                                            const int32_t *__restrict__ perm, int chunk_tiles)
{
    __shared__ DpArgs s_args;
    mpb_s_lut[threadIdx.x] = lut_g[threadIdx.x];
    if (threadIdx.x == 0) { s_args = args; mpb_s_cells = 0ull; }
    __syncthreads();
    const DpArgs *A = &s_args;
    const int total = tb->total_tiles;
    const int nchunks = (total + chunk_tiles - 1) / chunk_tiles;
    for (int chunk = blockIdx.x * 4 + (threadIdx.x >> 6); chunk < nchunks; chunk += gridDim.x * 4) {
        const int t0 = chunk * chunk_tiles;
        const int t1 = min(total, t0 + chunk_tiles);
        for (int c = MPB_NCLS - 1; c >= 0; c--) {          // tile order = descending class
            const int cnt = tb->count[c];
            const int rpt = 64 / c_classes[c].G;
            const int s = tb->tile_start[c];
            const int e = s + (cnt + rpt - 1) / rpt;
            const int lo = max(t0, s), hi = min(t1, e);
            if (lo >= hi) continue;
            const int32_t *pc = perm + tb->perm_base[c];
            switch (c) {
#define MPB_CASE(ID, RR, GG) case ID: dp_tiles<RR, GG, FMA>(A, pc, cnt, lo - s, hi - lo); break;
                MPB_CLASSES(MPB_CASE)
#undef MPB_CASE
            default: break;
            }
        }
    }
    if (args.prm.flags & 32u) {                                // MPB_FLAG_COUNT_CELLS: one device atomic per workgroup
        __syncthreads();
        if (threadIdx.x == 0 && mpb_s_cells) atomicAdd(args.alg_cells, mpb_s_cells);
    }
}

// ------------------------------------------------------------------------------------------
// k_wide: reads that need more than MPB_TILE_MAX_ROWS rows of the DP table (long AND bad: more than ~1000
// expected errors).  One workgroup of MPB_WIDE_WAVES waves per read; wave w keeps rows w*1024 .. w*1024+1023 of the
// running vector in registers (lane l: 16 consecutive rows, as the widest tile class).  The recurrence
//     v[j] = fl( fl(a*v[j]) + fl(b*v[j-1]) )
// needs, for the first row of a wave, the last row of the wave before it as it was BEFORE this base -- a stream of
// one double per base.  So the waves run one 64-base block apart (wave w works on block s - w in step s): wave w
// writes the pre-update value of its last row for each base of its block into an LDS stream, wave w + 1 reads it
// one step later; one workgroup barrier per 64 bases, two stream buffers per wave (written in step s, read in
// step s + 1, rewritten in step s + 2).  Only the waves a read's row budget needs do any work.  The epilogue is
// the same sequential CDF: the waves take their rows in order, the running sum handed on through LDS.
// Same arithmetic as the tile classes (three roundings per cell, no FMA), so results are bit-identical with
// the reference's at any length.
// ------------------------------------------------------------------------------------------
#define MPB_WIDE_R 16

// result of one read: ref bernoullimodule.c:170-178 + moira.py:827-831,911,925-926,949-950 (as the tile epilogue)
__device__ __forceinline__ void wide_report(const DpArgs &A, int idx, int li, int js, double lo, double hi, bool never_crossed)
{
    const double thr = A.prm.thr;
    double e;
    if (never_crossed) {
        e = __builtin_nan("");                     // CDF never crosses: the reference runs off its table
    } else {
        e = (double)(js - 1) + ((thr - lo) / (hi - lo));
        if (e < 0) e = 0;
    }
    const int nsv = gload(A.ns + idx);
    if (A.prm.ambig_mode == 0) e = e + (double)nsv;
    const double limit = (A.prm.maxerrors == A.prm.maxerrors) ? A.prm.maxerrors : (double)li * A.prm.uncert;
    if (A.prm.flags & 1u) e = floor(e);
    bool keep_read;
    if (A.prm.ambig_mode == 2 && (gload(A.cls + idx) & 0x80)) keep_read = false;
    else keep_read = e <= limit;
    gstore(A.ee + idx, e);
    gstore(A.pass + idx, (uint8_t)(keep_read ? 1 : 0));
}

// FINAL: every read gets len + 1 rows (the overflow pass); else the prepass' prediction.
// W: waves per workgroup (2, 4, 8, 16).  An instance takes the reads of its list that need more than W/2 and at most
// W waves (the W = 2 instance also those that need one: short reads of a long batch in the overflow pass) and skips the
// others, so that a read of 2,000 rows occupies 128 lanes, not 1,024 -- the four instances are launched back to back.
template <bool FINAL, int W>
__global__ __launch_bounds__(64 * W) void k_wide(DpArgs A, const double2 *__restrict__ lut_g,
                                                 const int32_t *__restrict__ list,
                                                 const int32_t *__restrict__ budget,
                                                 const int32_t *__restrict__ count)
{
    constexpr int R = MPB_WIDE_R;
    __shared__ double s_edge[W][2][64];
    __shared__ double s_acc[W];                     // CDF after the rows of waves 0..w
    __shared__ int s_found[W];                      // ... and whether it has crossed 1 - alpha by then
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    for (int k = tid; k < 256; k += 64 * W) mpb_s_lut[k] = lut_g[k];
    __syncthreads();
    const int nlist = gload(count);
    int keep = lane == 0 ? 0 : -1;
    asm volatile("" : "+v"(keep));
    const double thr = A.prm.thr;
    for (int k = blockIdx.x; k < nlist; k += gridDim.x) {          // block-uniform: every barrier below is reached by all
        const int idx = gload(list + k);
        const int li = A.len ? clamp_len(gload(A.len + idx), A.prm.max_len) : A.prm.fixed_len;
        const int rows = FINAL ? li + 1 : gload(budget + k);
        const int nw = max(1, min(MPB_WIDE_WAVES, (rows + 64 * R - 1) / (64 * R)));
        if (nw > W || (W > 2 && nw <= W / 2)) continue;            // another instance's read (block-uniform)
        const int nblk = (li + 63) >> 6;
        const uint8_t *row = A.q + (int64_t)idx * A.stride;

        double v[R];
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = 0.0;
        if (tid == 0) v[0] = 1.0;

        // lane l holds base t*64 + l of the wave's current block; bases past the read's end are byte 0 = the identity step
        uint32_t curb = 0, nxtb = 0;
        const int nsteps = nblk + nw - 1;
        for (int s = 0; s < nsteps; s++) {
            const int t = s - w;                                   // wave-uniform
            if (w < nw && t >= 0 && t < nblk) {
                if (t == 0) curb = (lane < li) ? (uint32_t)gload(row + lane) : 0u;
                const int nb0 = (t + 1) * 64 + lane;
                nxtb = (nb0 < li) ? (uint32_t)gload(row + nb0) : 0u;          // next block, in flight during this one
                __builtin_amdgcn_sched_barrier(0);
                const double *ein = &s_edge[w > 0 ? w - 1 : 0][(s + 1) & 1][0];   // what wave w-1 wrote in step s-1
                double *eout = &s_edge[w][s & 1][0];
#pragma unroll 1
                for (int kb = 0; kb < 64; kb += 4) {
#pragma unroll
                    for (int tt = 0; tt < 4; tt++) {
                        const uint32_t byte = (uint32_t)__builtin_amdgcn_readlane((int)curb, kb + tt);
                        const double2 ab = mpb_s_lut[byte];
                        const double top = v[R - 1];
                        double cin = dpp_prev_row(top, keep);       // lane 0: 0
                        if (w > 0) {                                // wave-uniform
                            const double e = ein[kb + tt];
                            cin = lane == 0 ? e : cin;
                        }
                        if (lane == 63) eout[kb + tt] = top;
#pragma unroll
                        for (int r = R - 1; r >= 1; r--) v[r] = cell<false>(ab.x, v[r], ab.y, v[r - 1]);
                        v[0] = cell<false>(ab.x, v[0], ab.y, cin);
                    }
                }
                curb = nxtb;
            }
            __syncthreads();
        }

        // ---- epilogue: sequential CDF over the rows of waves 0 .. nw-1 (ref: bernoullimodule.c:233-251) ----
        for (int ww = 0; ww < nw; ww++) {
            if (w == ww) {
                const double acc0 = ww ? s_acc[ww - 1] : 0.0;
                const int found0 = ww ? s_found[ww - 1] : 0;
                double acc = acc0;
                int found = found0;
                if (!found0) {                                      // wave-uniform
                    double acc_in_mine = 0.0;
                    bool mine = false;
#pragma unroll 1
                    for (int g = 0; g < 64; g++) {
                        double acc_s = acc0;
                        int found_s = 0;
                        if (g > 0) {
                            acc_s = __shfl(acc, g - 1);
                            found_s = __shfl(found, g - 1);
                        }
                        if (lane == g) {
                            double a = acc_s;
#pragma unroll
                            for (int r = 0; r < R; r++) a = a + v[r];
                            const bool cross = !found_s && (a > thr);
                            mine = cross;
                            acc_in_mine = acc_s;
                            acc = a;
                            found = found_s | (cross ? 1 : 0);
                        }
                    }
                    if (mine) {
                        double a = acc_in_mine, lo = 0.0, hi = 0.0;
                        int js = -1;
#pragma unroll
                        for (int r = 0; r < R; r++) {
                            const double na = a + v[r];
                            const bool hit = (js < 0) && (na > thr);
                            lo = hit ? a : lo;
                            hi = hit ? na : hi;
                            js = hit ? (w * 64 * R + lane * R + r) : js;
                            a = na;
                        }
                        wide_report(A, idx, li, js, lo, hi, false);
                    }
                }
                if (lane == 63) { s_acc[ww] = acc; s_found[ww] = found; }
            }
            __syncthreads();
        }
        if (tid == 0 && !s_found[nw - 1]) {                        // the CDF did not cross inside the row budget
            if (FINAL) {
                wide_report(A, idx, li, -1, 0.0, 0.0, true);
            } else {
                const int pos = atomicAdd(A.ovf_count, 1);
                A.ovf_list[pos] = idx;
            }
        }
        __syncthreads();                                           // s_acc / s_found / s_edge are reused by the next read
    }
}

// ------------------------------------------------------------------------------------------
// k_small: prediction + DP of ONE read per wave, one launch for the whole (small) batch.
// The batched pipeline costs eight launches and a sort whatever the batch size; a caller that hands
// over one read at a time (bernoulli.calculate_errors_PB from an unchanged moira.py) pays only
// latency.  Same prediction formula, same class bodies, same results; a read whose CDF does not
// cross inside its class is marked pass = 2 and the host sends the batch down the batched path.
// ------------------------------------------------------------------------------------------
// One read, one wave: statistics (and parking the row in device memory when it arrives in pinned host memory), the row
// prediction, the latency body of its class.  Shared by k_small (a launch per micro-batch) and k_serve (resident waves).
// `prm` and `li` are the read's own; s_args (LDS, this wave's) is what the non-inlined class body reads.
// SYS: the row lies in pinned host memory that is rewritten while the kernel runs (k_serve): it is read with system-scope
// loads, which never stop at a cache, instead of behind a cache-invalidating fence (which would also empty the L2 of what
// the class body is about to read: measured 6 us per request).  `get_prm` hands over the read's parameters when they are
// first needed -- after the statistics -- so that a caller who fetches them over the link can have that load in flight
// beside the row's.
// Returns true when the results went out with system-scope stores (the register-resident body below): the caller then needs
// no cache write-back before it tells the host.
template <bool FMA, bool SYS, typename PrmFn>
__device__ __forceinline__ bool small_one_read(const DpArgs &args, const DpArgs *s_args, PrmFn get_prm, const int li,
                                               const int64_t i, const int lane, int32_t *__restrict__ ns_out,
                                               uint8_t *__restrict__ cls_out, int32_t *__restrict__ ident,
                                               uint8_t *__restrict__ stage, const float2 *s_tab)
{
    // lane k takes the 16-byte chunks k, k + 64, ... of the row (at most 16 of them: 256 bytes, so the markers peel exactly)
    f32x2 a01 = {0.f, 0.f};
    float s3 = 0.f;
    uint4 yr = make_uint4(0, 0, 0, 0);                     // reads of <= 1024 bases: lane k's chunk k, bytes past the end zeroed
    for (int c0 = 0; c0 * 16 < li; c0 += 64) {            // wave-uniform trip count
        const int nv = li - (c0 + lane) * 16;
        if (nv > 0) {
            const uint8_t *src = args.q + i * args.stride + (int64_t)(c0 + lane) * 16;
            uint4 y;
            if (SYS) {
                const unsigned long long lo = __hip_atomic_load((const unsigned long long *)src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long hi = __hip_atomic_load((const unsigned long long *)src + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                y = make_uint4((uint32_t)lo, (uint32_t)(lo >> 32), (uint32_t)hi, (uint32_t)(hi >> 32));
            } else {
                y = *reinterpret_cast<const uint4 *>(src);
            }
            if (stage) *reinterpret_cast<uint4 *>(stage + i * args.stride + (int64_t)(c0 + lane) * 16) = y;
            if (SYS) yr = make_uint4(mask_dword(y.x, nv), mask_dword(y.y, nv - 4), mask_dword(y.z, nv - 8), mask_dword(y.w, nv - 12));
            y.x = fill_dword(y.x, nv); y.y = fill_dword(y.y, nv - 4);
            y.z = fill_dword(y.z, nv - 8); y.w = fill_dword(y.w, nv - 12);
            pre_chunk(s_tab, y, a01, s3);
        }
    }
    const float n255 = floorf(a01.y * (1.0f / MPB_MARK_LOWER));
    const float rem = a01.y - MPB_MARK_LOWER * n255;
    const float nzero_f = floorf(rem * (1.0f / MPB_MARK_UPPER));
    float mu = a01.x, var = rem - MPB_MARK_UPPER * nzero_f;
    float k3 = var - 2.0f * s3;
    uint32_t ambi = (uint32_t)nzero_f + ((uint32_t)n255 << 16);
#pragma unroll
    for (int off = 1; off <= 32; off <<= 1) {
        mu += __shfl_xor(mu, off);
        var += __shfl_xor(var, off);
        k3 += __shfl_xor(k3, off);
        ambi += __shfl_xor(ambi, off);
    }
    const int nzero = (int)(ambi & 0xffffu), n_lower = (int)(ambi >> 16);
    const MpbDevParams prm = get_prm();
    const float v = fmaxf(var, 1e-12f);
    const float x = mu + prm.z * sqrtf(v) + (k3 / v) * prm.zq;           // as k_prepass
    int rows = (int)floorf(fminf(x, 1e9f) + 0.5f) + 1;
    if (prm.flags & 4u) rows = rows / 2;                                  // MPB_FLAG_TEST_UNDERPREDICT
    rows = max(min(rows, li - nzero - n_lower + 1), 1);
    if (rows > MPB_TILE_MAX_ROWS) {                                       // a wide read: the host sends the batch down the pipeline
        if (lane == 0) args.pass[i] = 2;
        return false;
    }
    // (the resident server only -- interleaved A/B there: the in-process entry 17.0 against 18.4 us per call, the broker's
    // workers no different; in k_small's launches it was equal for one read and SLOWER for batches of 64-2048 reads,
    // 35 / 74 / 109 us against 29 / 43 / 60: profiles/r05_per_read_server.txt)
    if (SYS && !FMA && li <= 1024 && rows <= 64 && !(prm.flags & ~1u)) {
        // The resident server's common case, entirely in registers: the row is already here (lane k holds chunk k), so the
        // class body's trips to memory -- the parked row, ns / cls / ident read back, the callee-saved registers of a
        // non-inlined body -- are not needed: 2.5 us of a request's 12.9.  One row per lane (the latency bodies' shape), the
        // read's bytes handed round by v_readlane, the same cell, the same sequential CDF, the same expressions at the end.
        int keep = lane == 0 ? 0 : -1;
        asm volatile("" : "+v"(keep));
        double v1 = lane == 0 ? 1.0 : 0.0;
        const int nchunks = (li + 15) >> 4;
#pragma unroll 1
        for (int ck = 0; ck < nchunks; ck++) {
            const uint32_t wq[4] = {(uint32_t)__builtin_amdgcn_readlane((int)yr.x, ck), (uint32_t)__builtin_amdgcn_readlane((int)yr.y, ck),
                                    (uint32_t)__builtin_amdgcn_readlane((int)yr.z, ck), (uint32_t)__builtin_amdgcn_readlane((int)yr.w, ck)};
#pragma unroll
            for (int d = 0; d < 4; d++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double2 ab = mpb_s_lut[(wq[d] >> (8 * t)) & 0xffu];      // bytes past the read's end are zero: the identity step
                    const double cin = dpp_prev_row(v1, keep);
                    v1 = cell<false>(ab.x, v1, ab.y, cin);
                }
            }
        }
        // sequential CDF over the rows (= lanes) in order (ref: bernoullimodule.c:233-251); rows beyond the predicted budget are
        // looked at too (they are there), a read that still does not cross is the host's (pass = 2), as in the class bodies
        const double thr = prm.thr;
        double acc = 0.0, lo = 0.0, hi = 0.0;
        int js = -1;
#pragma unroll 1
        for (int g = 0; g < 64 && js < 0; g++) {
            const long long bits = __double_as_longlong(v1);
            const double vg = __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(bits >> 32), g) << 32) |
                                                   (unsigned int)__builtin_amdgcn_readlane((int)bits, g));
            const double na = acc + vg;                    // g == 0: 0 + v0 is exact
            if (na > thr) { lo = acc; hi = na; js = g; }
            acc = na;
        }
        if (lane == 0) {
            // system-scope stores: straight to the host's memory, whatever a cache would do with them
            const int nsv = nzero + n_lower;
            __hip_atomic_store(ns_out + i, nsv, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (js < 0) {
                __hip_atomic_store(args.pass + i, (uint8_t)2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            } else {
                double e = (double)(js - 1) + ((thr - lo) / (hi - lo));     // ref: bernoullimodule.c:170-178
                if (e < 0) e = 0;
                if (prm.ambig_mode == 0) e = e + (double)nsv;                // moira.py:827-828
                const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors : (double)li * prm.uncert;
                if (prm.flags & 1u) e = floor(e);                            // moira.py:830-831
                __hip_atomic_store((unsigned long long *)(args.ee + i), (unsigned long long)__double_as_longlong(e), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                __hip_atomic_store(args.pass + i, (uint8_t)((prm.ambig_mode == 2 && nzero > 0) ? 0 : (e <= limit ? 1 : 0)),   // moira.py:911
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return true;
    }
    const int c = c_class_of_rows.t[rows];
    // One read per wave: a G = 1 body would keep ONE lane busy.  The latency bodies below spread the read's rows over as
    // many lanes as they fill two (then 4 / 8 / 16) rows each -- cap 2, 4, 8 ... 1024 = the next power of two -- so that a
    // base costs ~10 instead of ~3 * rows instructions of the wave's dependent chain (a 300-base read of 10 rows:
    // ~18 -> ~8 us).  Same cell arithmetic, same sequential CDF: the split of the rows over lanes never shows in a result.
    const int thin = rows <= 2 ? 0 : 31 - __builtin_clz(rows - 1);
    bool settled = false;
    if (prm.flags & 8u) {                                                 // MPB_FLAG_DECISION_ONLY, as k_prepass
        const float t = mu * (1.0f - 1e-4f) - prm.clow * sqrtf(mu) - 0.02f;
        const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors : (double)li * prm.uncert;
        settled = mu > 1.0f && (double)floorf(t) > limit;
    }
    if (lane == 0) {
        ns_out[i] = nzero + n_lower;
        if (args.ns != ns_out) const_cast<int32_t *>(args.ns)[i] = nzero + n_lower;
        cls_out[i] = (uint8_t)((settled ? MPB_CLS_SETTLED : c) | (nzero > 0 ? 0x80 : 0));
        ident[i] = (int32_t)i;
        if (settled) { args.ee[i] = __builtin_inf(); args.pass[i] = 0; }
    }
    if (settled) return false;
    // the class body (this wave, other lanes) reads ns / cls / ident / the parked row back: the stores out of the CU, its L1 emptied
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    const int32_t *pc = ident + i;
    switch (thin) {
#define MPB_CASE(ID, RR, GG) case ID: dp_tiles<RR, GG, FMA>(s_args, pc, 1, 0, 1); break;
        MPB_THIN_CLASSES(MPB_CASE)
#undef MPB_CASE
    default: break;
    }
    return false;
}

// the per-wave tables of the one-read path: {1 - p, p'} in LDS (module scope, for the class bodies) and the fp32 statistics table
__device__ __forceinline__ void small_tables(const double2 *__restrict__ lut_g, float2 *s_tab, const int tid0, const int nthreads)
{
    for (int tid = tid0; tid < 256; tid += nthreads) {
        mpb_s_lut[tid] = lut_g[tid];
        const bool amb = tid == 0 || tid == 255;
        float p = __builtin_amdgcn_exp2f(-0.33219281f * (float)tid);      // 10^(-q/10)
        p = amb ? 0.0f : p;
        s_tab[tid] = make_float2(p, tid == 0 ? MPB_MARK_UPPER : tid == 255 ? MPB_MARK_LOWER : p * (1.0f - p));
    }
}

template <bool FMA>
__global__ __launch_bounds__(256, MPB_DP_WAVES_PER_EU) void k_small(DpArgs args, const double2 *__restrict__ lut_g,
                                                                    int64_t n, int32_t *__restrict__ ns_out,
                                                                    uint8_t *__restrict__ cls_out,
                                                                    int32_t *__restrict__ ident,
                                                                    uint8_t *__restrict__ stage,
                                                                    uint32_t *__restrict__ done, uint32_t token)
{
    // done != nullptr (pinned host memory): when a read's results are out -- and visible system-wide -- its wave writes
    // `token` to done[i], so that the host learns of the END OF ITS READS from memory instead of from the runtime's
    // completion signal (which comes several microseconds after the last wave, and costs a runtime call to ask for).
    // ns_out is where the ambiguity counts are REPORTED; args.ns (device memory, may be the same array) is where the class
    // body reads them back: a report that lives in host memory is not read back over the link.
    // stage != nullptr: args.q is pinned HOST memory (the per-read entry and the broker's micro-batches: one runtime call
    // per launch, no copy in).  The statistics pass reads every chunk of the row anyway -- one trip over the link, all
    // lanes at once -- and leaves it in `stage` (device memory, same shape), which is what the class body then walks: its
    // five dependent 64-byte trips per 300 bases would otherwise each pay the link's latency.
    // Where a one-read call's 12.5 us inside the kernel go (300 bases, in-kernel stamps, profiles/r05_per_read_server.txt):
    // tables + barrier 0.7, row over the link + statistics 2.6, class body 8.5 (2.7 + 19 ns per base), completion 0.6.
    __shared__ float2 s_tab[256];
    __shared__ DpArgs s_args;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    small_tables(lut_g, s_tab, tid, 256);
    if (tid == 0) { s_args = args; if (stage) s_args.q = stage; }
    __syncthreads();
    const int64_t i = (int64_t)blockIdx.x * 4 + w;
    if (i >= n) return;                                   // wave-uniform; no barrier below
    const int li = args.len ? clamp_len(args.len[i], args.prm.max_len) : args.prm.fixed_len;
    (void)small_one_read<FMA, false>(args, &s_args, [&] { return args.prm; }, li, i, lane, ns_out, cls_out, ident, stage, s_tab);
    if (done) {
        __threadfence_system();                           // every lane's stores of this wave: out, and visible to the host
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(done + i, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// ------------------------------------------------------------------------------------------
// k_serve: the one-read path WITHOUT a launch per call (round 5).  A per-read call through k_small costs 27-28 us of which
// the kernel is 12.5 and the launch + dispatch + completion hand-shake the rest; with P workers calling at once the broker
// thread's launches bound the rate.  So the broker keeps this kernel resident while calls arrive: one wave per mailbox entry
// (= per broker slot), every entry in pinned host memory.  A wave polls its entry's 8-byte door word {token, length} over the
// link; a new token = a request: it reads the request's parameters and its row (one trip over the link, as k_small), runs
// small_one_read, writes ee / Ns / pass to the entry, fences system-wide and stores the token to done[e].  The host side
// (mpb_broker.cpp) writes row + parameters, then the door word; it learns of the result from done[e].
// EVERY wave leaves when the host sets *stop, and -- whatever the host does -- when its lifetime (ticks of the 100 MHz
// clock) is over: the grid always drains.  The last wave out stores `generation` to *exited, so that the host re-launches at
// once while calls keep coming.  A request posted to an entry whose wave has just left is served by the next launch (a
// wave starts from done[e]: whatever token differs from it is pending).
// ------------------------------------------------------------------------------------------
// ONE wave per workgroup: a CU's vector memory path hands data back in request order, so a sibling wave that polls the
// link (2 us per look) on the same CU delays every load of the wave that works (measured: 23 instead of 9 us of class body).
// (Launched with 64 threads per workgroup, but DECLARED with k_dp's bounds: the class bodies are functions shared with k_dp,
// and a caller that allowed them 512 registers would halve k_dp's occupancy.)
__global__ __launch_bounds__(256, MPB_DP_WAVES_PER_EU) void k_serve(MpbServeBox box, const double2 *__restrict__ lut_g,
                                                                    uint32_t generation, unsigned long long lifetime_ticks)
{
    __shared__ float2 s_tab[256];
    __shared__ DpArgs s_args[1];
    const int lane = threadIdx.x;
    constexpr int w = 0;
    small_tables(lut_g, s_tab, lane, 64);
    __syncthreads();
    const int e = blockIdx.x;                             // this wave's entry
    if (e < box.n_ent) {
        const long long t_start = wall_clock64();
        // this entry's own pointers (everything below indexes read 0 of them)
        auto at = [e](const void *base, int64_t step) { return (char *)const_cast<void *>(base) + (int64_t)e * step; };
        const unsigned long long *e_door = (const unsigned long long *)at(box.door, box.door_step);
        uint32_t *e_done = (uint32_t *)at(box.done, box.done_step);
        const MpbServePrm *e_prm = (const MpbServePrm *)at(box.prm, box.prm_step);
        int32_t *e_ns = (int32_t *)at(box.ns, box.ns_step);
        uint8_t *e_stage = box.stage + (int64_t)e * box.stride;
        DpArgs A;
        A.q = (const uint8_t *)at(box.q, box.q_step); A.stride = box.stride; A.len = nullptr; A.ns = box.ns_dev + e; A.cls = box.cls + e;
        A.ee = (double *)at(box.ee, box.ee_step); A.pass = (uint8_t *)at(box.pass, box.pass_step);
        A.ovf_list = nullptr; A.ovf_count = nullptr; A.alg_cells = nullptr; A.perm = nullptr; A.perm_ns = nullptr; A.final_pass = 2;
        uint32_t last = __hip_atomic_load(e_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
        int idle_looks = 0;
        for (;;) {
            const unsigned long long door = __hip_atomic_load(e_door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint32_t token = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)door);
            if (token != last) {
                asm volatile("" ::: "memory");               // what the host wrote before the door word is READ from here on, and
                                                             // only with system-scope loads (no cache holds it): no invalidate
                const int li = clamp_len(__builtin_amdgcn_readfirstlane((int)(uint32_t)(door >> 32)), (int)box.stride);
                // the request's parameters: 64 bytes of pinned host memory, 16 per lane of the first four -- requested here,
                // looked at after the row's statistics (one trip over the link for both)
                const unsigned long long *hp = (const unsigned long long *)e_prm + 2 * (lane & 3);
                const unsigned long long p_lo = __hip_atomic_load(hp, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                const unsigned long long p_hi = __hip_atomic_load(hp + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                static_assert(sizeof(MpbServePrm) == 64, "one 64-byte line per request");
                bool bad_request = false;                            // wave-uniform
                auto get_prm = [&] {
                    const uint32_t xs[4] = {(uint32_t)p_lo, (uint32_t)(p_lo >> 32), (uint32_t)p_hi, (uint32_t)(p_hi >> 32)};
                    uint32_t pw[16];
#pragma unroll
                    for (int k = 0; k < 16; k++) pw[k] = (uint32_t)__builtin_amdgcn_readlane((int)xs[k & 3], k >> 2);
                    MpbDevParams prm;
                    __builtin_memcpy(&prm, pw, sizeof(prm));
                    // In direct serving these 64 bytes are the CLIENT's (a shared-memory slot no broker thread has checked): only
                    // alpha's threshold and the prediction constants are taken from them; everything a per-read call never sets is
                    // forced to what mpbi_small_params writes (no opt-in flag, ambiguous bases ignored, no limit), and a threshold
                    // outside (0, 1) makes the request one the host answers itself through its checked path (pass = 2 below).
                    prm.flags = 0;
                    prm.ambig_mode = 1;                             // MPB_AMBIG_IGNORE
                    prm.uncert = 1.0;
                    prm.maxerrors = __builtin_nan("");
                    bad_request = !(prm.thr > 0.0 && prm.thr < 1.0);
                    if (bad_request) prm.thr = 0.5;
                    prm.fixed_len = li;
                    prm.max_len = (int32_t)box.stride;
                    A.prm = prm;
                    if (lane == 0) { s_args[w] = A; s_args[w].q = e_stage; }
                    wave_lds_fence();
                    return prm;
                };
                bool light = small_one_read<false, true>(A, &s_args[w], get_prm, li, 0, lane, e_ns, box.cls + e, box.ident + e, e_stage, s_tab);
                if (bad_request) {                                   // whatever was computed on the stand-in threshold is withdrawn
                    if (lane == 0) __hip_atomic_store(A.pass, (uint8_t)2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    light = false;
                }
                // the results before the word that announces them.  Stores made at system scope go straight out and reach the host
                // in the order they were issued once the wave has seen them acknowledged; anything else (the class bodies'
                // plain stores) is written back from the caches first
                if (light) {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    if (lane == 0) __hip_atomic_store(e_done, token, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                } else {
                    __threadfence_system();
                    __builtin_amdgcn_wave_barrier();
                    if (lane == 0) __hip_atomic_store(e_done, token, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
                }
                last = token;
                idle_looks = 0;
                continue;
            }
            if (__hip_atomic_load(box.stop, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u) break;
            if ((unsigned long long)(wall_clock64() - t_start) > lifetime_ticks) break;
            // an entry that is being called looks again at once (a look is 2.6 us over the link anyway); one that has been quiet for
            // half a millisecond every 3 us, for five every 14 us: 64 waves of a broker whose workers are elsewhere ask the link
            // 5 M times a second instead of 25 M, at the price of 7 us on the first call after a pause
            idle_looks++;
            if (idle_looks < 200) __builtin_amdgcn_s_sleep(16);
            else if (idle_looks < 2000) __builtin_amdgcn_s_sleep(127);
            else { __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); __builtin_amdgcn_s_sleep(127); }
        }
    }
    // the last wave out tells the host
    __builtin_amdgcn_wave_barrier();
    if (lane == 0) {
        const uint32_t gone = atomicAdd(box.gone, 1u) + 1u;
        if (gone == gridDim.x) {
            __threadfence_system();
            __hip_atomic_store(box.exited, generation, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_count(const uint8_t *__restrict__ pass, int64_t n,
                                               unsigned long long *__restrict__ out)
{
    unsigned int s = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        s += pass[i];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) s += __shfl_xor(s, off);
    __shared__ unsigned int s_w[4];
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(out, (unsigned long long)(s_w[0] + s_w[1] + s_w[2] + s_w[3]));
}

// ------------------------------------------------------------------------------------------
// k_lambda (--error_calc poisson): lambda = sum of the error probabilities of a read's non-'N' bases IN BASE
// ORDER, so that the sum has the reference's association (moira/moira.py:1663  Lambda += 10**(qscore / -10.0))
// and is bit-identical to it.  The order pins one accumulator to one read, i.e. one lane per row -- which, read
// straight from HBM, is 64 rows per load instruction and a cache line re-visited eight times.  So a wave moves
// its 64 rows through LDS in panels of 128 columns: coalesced loads shaped like the prepass' (16 rows x 64 B per
// instruction, the next panel in flight while the current one is summed), a padded row pitch (144 B: the 16
// lanes an LDS b128 access groups together land in 16 different bank quads), then every lane walks ITS row.
// The loads go through a range-checked buffer descriptor per 64 rows (round 3): no predicate and no 64-bit address
// arithmetic per load, and the 16 table look-ups of a chunk are all issued before its ordered adds.  Where the time
// goes: the sums alone take 0.41 ms per 10 M x 300 bases, the loads + tile writes alone 0.61-0.65 ms --
// the kernel runs at what its access shape streams, the arithmetic is hidden.
// 'N' (byte 0) looks up 0.0 (x + 0.0 == x: the reference skips the base), lower-case 'n' (byte 255) looks up a NaN
// that poisons the sum and is reported; Ns are counted eight bytes at a time with integer arithmetic.
// ------------------------------------------------------------------------------------------
#define MPB_LAM_W 128                       // 64, 128, 192, 256: measured 128 best
#define MPB_LAM_PITCH (MPB_LAM_W + 16)      // (pitch / 16) odd for every W above
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int count_zero_bytes(uint32_t w)
{
    uint32_t t = (w & 0x7f7f7f7fu) + 0x7f7f7f7fu;       // bit 7 of a byte is set iff its low 7 bits are non-zero
    t = ~(t | w | 0x7f7f7f7fu);                         // 0x80 where the whole byte is zero
    return __popc(t);
}

template <bool RAGGED>
__global__ __launch_bounds__(256) void k_lambda(const uint8_t *__restrict__ q, int64_t n, int64_t stride,
                                                const int32_t *__restrict__ len_arg, int32_t fixed_len,
                                                const double2 *__restrict__ lut_ap,
                                                double *__restrict__ lambda, int32_t *__restrict__ ns,
                                                int32_t *__restrict__ bad)
{
    __shared__ double s_p[256];
    __shared__ __attribute__((aligned(16))) uint8_t s_tile[4][64 * MPB_LAM_PITCH];
    const int32_t *__restrict__ len = RAGGED ? len_arg : nullptr;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    // the DP LUT holds {1-p, p'}; p' == p bit for bit (tests/test_oracle_golden.py::test_lut_pins)
    s_p[tid] = tid == 0 ? 0.0 : (tid == 255 ? __builtin_nan("") : lut_ap[tid].y);
    __syncthreads();
    uint8_t *tile = s_tile[w];
    const int r16 = lane & 15, cl = lane >> 4;
    const int row_bytes = (int)stride;
    for (int64_t row0 = ((int64_t)blockIdx.x * 4 + w) * 64; row0 < n; row0 += (int64_t)gridDim.x * 256) {
        const int64_t i = row0 + lane;
        const bool live = i < n;
        const int li = live ? (len ? clamp_len(len[i], row_bytes) : fixed_len) : 0;
        int lmax = li;
        if (RAGGED || row0 + 64 > n) {
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) lmax = max(lmax, __shfl_xor(lmax, off));
        }
        lmax = __builtin_amdgcn_readfirstlane(lmax);
        const int npanel = (lmax + MPB_LAM_W - 1) / MPB_LAM_W;
        double lam = 0.0;
        int nzero = 0;
        uint4 pre[MPB_LAM_W / 16];
        // The wave's 64 rows as one range-checked buffer: 32-bit offsets (lane part in a VGPR, row-group part in an SGPR),
        // no predicate per load -- a row past the end of the matrix reads zeros, and bytes past a row's end (the next
        // row's, in a panel that overhangs the stride) are never looked at (the chunk loop stops at lmax and masks by li).
        const uint64_t wave_base = (uint64_t)(uintptr_t)q + (uint64_t)row0 * (uint64_t)stride;
        const int64_t rows_here = (n - row0) < 64 ? (n - row0) : 64;
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wave_base >> 32));
        const uint32_t b_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rows_here * stride));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(uintptr_t)(((uint64_t)b_hi << 32) | b_lo), 0, (int)b_n, 0x00020000);
        const int voff0 = r16 * row_bytes + cl * 16;
        auto load_panel = [&](int p) {
#pragma unroll
            for (int rg = 0; rg < 4; rg++)
#pragma unroll
                for (int cq = 0; cq < MPB_LAM_W / 64; cq++) {
                    pre[rg * (MPB_LAM_W / 64) + cq] = make_uint4(0, 0, 0, 0);
                    if (p * MPB_LAM_W + cq * 64 < lmax) {                                   // wave-uniform
                        const u32x4 g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff0 + p * MPB_LAM_W, rg * 16 * row_bytes + cq * 64, 0);
                        pre[rg * (MPB_LAM_W / 64) + cq] = make_uint4(g.x, g.y, g.z, g.w);
                    }
                }
        };
        if (npanel > 0) load_panel(0);
        for (int p = 0; p < npanel; p++) {
#pragma unroll
            for (int rg = 0; rg < 4; rg++)
#pragma unroll
                for (int cq = 0; cq < MPB_LAM_W / 64; cq++)
                    *reinterpret_cast<uint4 *>(tile + (rg * 16 + r16) * MPB_LAM_PITCH + cq * 64 + cl * 16) = pre[rg * (MPB_LAM_W / 64) + cq];
            wave_lds_fence();
            if (p + 1 < npanel) load_panel(p + 1);        // in flight while this panel is summed
            __builtin_amdgcn_sched_barrier(0);
            const int base = p * MPB_LAM_W;
            const int nch = (min(MPB_LAM_W, lmax - base) + 15) >> 4;          // wave-uniform
            const uint8_t *mine = tile + lane * MPB_LAM_PITCH;
            for (int c = 0; c < nch; c++) {
                uint4 x = *reinterpret_cast<const uint4 *>(mine + c * 16);
                const int nv = li - base - c * 16;           // this lane's valid bytes in the chunk (may be <= 0)
                uint32_t cw[4] = {x.x, x.y, x.z, x.w};       // what is counted: bytes past the end must not look like 'N'
                if (nv < 16) {
                    x.x = mask_dword(x.x, nv); x.y = mask_dword(x.y, nv - 4);
                    x.z = mask_dword(x.z, nv - 8); x.w = mask_dword(x.w, nv - 12);
                    cw[0] = fill_dword(cw[0], nv); cw[1] = fill_dword(cw[1], nv - 4);
                    cw[2] = fill_dword(cw[2], nv - 8); cw[3] = fill_dword(cw[3], nv - 12);
                }
                nzero += count_zero_bytes(cw[0]) + count_zero_bytes(cw[1]) + count_zero_bytes(cw[2]) + count_zero_bytes(cw[3]);
                const uint32_t ww[4] = {x.x, x.y, x.z, x.w};
                double pv[16];                                  // all 16 look-ups in flight before the (ordered) adds
#pragma unroll
                for (int d = 0; d < 4; d++)
#pragma unroll
                    for (int t = 0; t < 4; t++) pv[d * 4 + t] = s_p[(ww[d] >> (8 * t)) & 0xffu];
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int k = 0; k < 16; k++) lam = lam + pv[k];                                   // in base order
            }
            wave_lds_fence();                             // the tile is overwritten by the next panel
        }
        if (live) {
            lambda[i] = lam;
            ns[i] = nzero;
            if (lam != lam) atomicAdd(bad, 1);            // only a byte 255 can do that
        }
    }
}

// ------------------------------------------------------------------------------------------
// k_decode_ascii: FASTQ quality bytes + base letters -> packed qscores, 16 bytes per lane
// (ref: moira/moira.py:1177 `ord(x) - offset`, bernoullimodule.c:104-107 Q0->1, :196 N / n)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_decode_ascii(const uint8_t *__restrict__ seq,
                                                      const uint8_t *__restrict__ qual, int64_t n,
                                                      int64_t stride, const int32_t *__restrict__ len,
                                                      int32_t fixed_len, int32_t offset,
                                                      uint8_t *__restrict__ out, int32_t *__restrict__ err)
{
    const int64_t cpr = stride / 16;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * cpr) return;
    const int64_t i = g / cpr;
    const int c = (int)(g - i * cpr);
    const int li = len ? clamp_len(len[i], (int)stride) : fixed_len;
    const uint4 sq = *reinterpret_cast<const uint4 *>(seq + i * stride + (int64_t)c * 16);
    const uint4 ql = *reinterpret_cast<const uint4 *>(qual + i * stride + (int64_t)c * 16);
    int bad = 0;
    *reinterpret_cast<uint4 *>(out + i * stride + (int64_t)c * 16) = decode16(sq, ql, c * 16, li, offset, bad);
    if (bad && err) atomicAdd(err, bad);
}

// the inverse (tests and bench: builds FASTQ-text matrices from a packed one; decode(encode(q)) == q for Q <= 93 at offset 33)
__global__ __launch_bounds__(256) void k_encode_ascii(const uint8_t *__restrict__ q, int64_t n, int64_t stride,
                                                      int32_t offset, uint8_t *__restrict__ seq, uint8_t *__restrict__ qual)
{
    const int64_t cpr = stride / 16;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * cpr) return;
    const uint4 x = *reinterpret_cast<const uint4 *>(q + g * 16);
    const uint32_t xw[4] = {x.x, x.y, x.z, x.w};
    uint32_t sw[4] = {0, 0, 0, 0}, qw[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const uint32_t v = (xw[t >> 2] >> (8 * (t & 3))) & 0xffu;
        const uint32_t base = v == 0 ? 'N' : v == 255 ? 'n' : (uint32_t)"ACGT"[t & 3];
        const uint32_t qc = (v == 0 || v == 255) ? (uint32_t)(offset + 2) : min(255u, v + (uint32_t)offset);
        sw[t >> 2] |= base << (8 * (t & 3));
        qw[t >> 2] |= qc << (8 * (t & 3));
    }
    *reinterpret_cast<uint4 *>(seq + g * 16) = make_uint4(sw[0], sw[1], sw[2], sw[3]);
    *reinterpret_cast<uint4 *>(qual + g * 16) = make_uint4(qw[0], qw[1], qw[2], qw[3]);
}

// ------------------------------------------------------------------------------------------
// synthetic fill: one thread per 16-byte chunk of the matrix
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_synth(uint8_t *__restrict__ q, int64_t n, int64_t stride,
                                               int32_t fixed_len, int32_t min_len, int32_t max_len,
                                               int32_t *__restrict__ len, uint64_t seed,
                                               int64_t first_read, int32_t profile)
{
    const int64_t cpr = stride / 16;
    const int64_t g = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (g >= n * cpr) return;
    const int64_t i = g / cpr;
    const int c = (int)(g - i * cpr);
    const uint64_t h = mpb_synth_read_hash(seed, (uint64_t)(first_read + i));
    const int32_t li = fixed_len > 0 ? fixed_len : mpb_synth_len(h, min_len, max_len);
    uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 16; t++) {
        const int pos = c * 16 + t;
        const uint32_t b = pos < li ? (uint32_t)mpb_synth_byte_profile(h, (uint32_t)pos, (uint32_t)li, profile) : 0u;
        w[t >> 2] |= b << (8 * (t & 3));
    }
    *reinterpret_cast<uint4 *>(q + i * stride + (int64_t)c * 16) = make_uint4(w[0], w[1], w[2], w[3]);
    if (len && c == 0) len[i] = li;
}

// ------------------------------------------------------------------------------------------
// k_narrow<R>: the natural-order pass for batches of GOOD reads (round 5; mpb_internal.h "natural-order narrow pass").
//
// Where it applies the step is bound by HBM, so the matrix is read exactly once, in the order it lies in memory, and nothing
// else of size n is read or written except the results.  One read per lane, R rows of the running vector in registers
// (R = 2..4): rows 0..R-1 of the reference's table depend on no later row, so they are exact whatever the read needs, and a read
// whose CDF crosses 1 - alpha inside them is finished here with the reference's own three-rounding cell, sequential CDF and
// interpolation (ref: moira/bernoullimodule.c:152-166,219-251).  'N' bases are the identity step, as everywhere, and are counted
// four bytes per instruction (v_msad_u8, below).  Every other read -- more rows needed, or a lower-case 'n' (byte 255), whose
// table entry in THIS pass is a NaN that poisons the read's vector and so keeps its CDF from ever crossing: nothing is spent
// on looking for them -- is appended to a list for the ordinary pipeline.
//
// A lane per row read straight from HBM is 64 rows per load instruction and every cache line revisited by eight instructions.
// So a wave moves its 64 rows through LDS in 64-byte panels by LDS-DMA (global_load_lds_dwordx4, no register staging): one
// instruction covers 64 consecutive bytes of each of 16 rows (the prepass' shape) and lands as [4 chunk columns][16 rows] x 16 B,
// so the 16 lanes an LDS b128 read groups together hit 16 different bank quads; a private ring of MPB_NAR_DEPTH panels (4 KiB
// each) per wave, a slot refilled as soon as its panel is in registers, keeps DEPTH panels in flight across row-block
// boundaries -- the grid is persistent, a wave walks row blocks gw, gw + W, ... as one continuous stream of panels.
// Per base: 1 address op + 3R - 2 cell operations + half an instruction of 'N' counting = 5.5 / 8.5 / 11.5 vector instructions
// for R = 2 / 3 / 4 (a form that keeps {p'} alone and recomputes 1 - p costs one more and was 5-8 % slower:
// profiles/r05_narrow_variants.txt, git history; mpb_create still checks a == 1 - b for every score).
// ------------------------------------------------------------------------------------------
#define MPB_NAR_DEPTH 2                     // ring slots per wave: 2 x 4 KiB -> four workgroups (16 waves) per CU; a slot is refilled as
                                            // soon as its panel is in registers, so two panels per wave are in flight during a step
#define MPB_NAR_PANEL 4096                  // 64 rows x 64 bytes

// One LDS-DMA instruction: 16 bytes per lane from `base + voff` (wave-uniform 64-bit base, per-lane 32-bit offset) to LDS at
// lds + 16 * lane.  Written as inline assembly on purpose: the compiler orders every LDS read behind every LDS-DMA that it
// cannot prove independent -- a wave's ring slots are one array indexed at run time, so it drains the whole prefetch stream
// (s_waitcnt vmcnt(0)) before each panel is read.  Here the waits are counted by hand instead (nar_wait): every panel is four
// of these, requests complete in issue order, and anything else the wave has in flight (the results' stores, the list's
// atomic) only makes a counted wait stricter.  M0 = LDS address; one wait state between the M0 write and the DMA.
__device__ __forceinline__ void nar_dma16(const uint8_t *base, uint32_t voff, uint32_t lds)
{
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2"
                 :: "s"(lds), "v"(voff), "s"(base) : "memory");      // (`nt` requests: 5.9 instead of 4.0 GB per launch, slower)
}
template <int N>
__device__ __forceinline__ void nar_wait()                  // until at most N vector-memory operations are outstanding
{
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory");
}
__device__ __forceinline__ uint32_t lds_offset(const void *p)      // a __shared__ object's address inside the workgroup's LDS
{
    return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p;
}

// The table entry: the {1 - p, p'} pair as the context holds it (one ds_read_b128 per base).  (p' alone with 1 - p recomputed by
// the IEEE subtraction the host used -- half the LDS cycles, one more vector instruction per base -- was measured and dropped.)
// Once the pass counts its 'N' bases itself it is bound by vector issue, not by the stream, and the pair is 5 % faster
// (profiles/r05_narrow_variants.txt); LDS is 50 % busy with it.
typedef double2 nar_entry_t;
#define NAR_P(e) ((e).y)
#define NAR_A(e) ((e).x)

template <int R>
__device__ __forceinline__ void nar_step(double (&v)[R], const nar_entry_t e)
{
    const double a = NAR_A(e), p = NAR_P(e);
#pragma unroll
    for (int r = R - 1; r >= 1; r--) v[r] = cell<false>(a, v[r], p, v[r - 1]);
    v[0] = a * v[0];
}

template <int R>
__device__ __forceinline__ void nar_dword(double (&v)[R], const nar_entry_t *tab, uint32_t w)
{
#pragma unroll
    for (int t = 0; t < 4; t++) nar_step<R>(v, tab[(w >> (8 * t)) & 0xffu]);
}


// (Rows whose stride is a multiple of 64 bytes take k_narrow_rs below since the second session of round 5; a form of this
// kernel that fetched the line a row pair shares only once -- tail buffers beside the ring, -DMPB_NAR_TAILS -- is in the
// history: 4.03 -> 3.49 GB read, 6 % slower.)
template <int R, int D>
__global__ __launch_bounds__(256) void k_narrow(const uint8_t *__restrict__ q, int64_t n, int64_t stride, int32_t li,
                                                MpbDevParams prm, const double2 *__restrict__ lut_g,
                                                double *__restrict__ ee, int32_t *__restrict__ ns, uint8_t *__restrict__ pass,
                                                int32_t *__restrict__ seg, int32_t *__restrict__ wave_count)
{
    // seg / wave_count: the reads this pass cannot finish.  A wave appends them to a segment of its own -- wave gw owns the
    // slots of the row blocks it walks, which start at 64 * (blocks owned by the waves before it) -- and leaves its count in
    // wave_count[gw]: no atomic (a returned atomic would drain the prefetch stream: vmcnt counts everything), and a list whose
    // order does not depend on timing.  k_nar_compact then makes the dense list.
    static_assert(D >= 2 && D <= 4, "ring depth");
    __shared__ nar_entry_t s_p[256];
    // one array per ring slot: a read of slot k is then provably independent of a DMA into slot k + 1 (the compiler orders
    // LDS reads behind LDS-DMA by what may alias)
    __shared__ __attribute__((aligned(16))) uint8_t s_ring0[4][MPB_NAR_PANEL];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring1[4][MPB_NAR_PANEL];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring2[D > 2 ? 4 : 1][D > 2 ? MPB_NAR_PANEL : 16];
    __shared__ __attribute__((aligned(16))) uint8_t s_ring3[D > 3 ? 4 : 1][D > 3 ? MPB_NAR_PANEL : 16];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // byte 0 ('N'): the identity step {1, 0} the table holds anyway (counted below); byte 255 ('n'): a NaN -- the read is handed back
    s_p[tid] = tid == 255 ? make_double2(__builtin_nan(""), __builtin_nan("")) : lut_g[tid];
    __syncthreads();                                          // the only block barrier
    const int64_t nblk = (n + 63) >> 6;                       // row blocks of 64 reads
    const int ncq = (li + 63) >> 6;                           // 64-byte panels per row block (li >= 1)
    const int64_t gw = (int64_t)blockIdx.x * 4 + w, W = (int64_t)gridDim.x * 4;
    if (gw >= nblk) {
        if (lane == 0) wave_count[gw] = 0;
        return;
    }
    const int64_t total = ((nblk - gw + W - 1) / W) * ncq;    // panels this wave walks
    int32_t *const my_seg = seg + 64 * (gw * (nblk / W) + min(gw, nblk % W));
    int nlist = 0;                                            // wave-uniform
    const int r16 = lane & 15, cl = lane >> 4;
    const int row_chunks = __builtin_amdgcn_readfirstlane((int)(stride >> 4));
    uint32_t voff[4];
#pragma unroll
    for (int rg = 0; rg < 4; rg++) voff[rg] = (uint32_t)((rg * 16 + r16) * (int)stride + cl * 16);
    uint8_t *const ring[4] = {s_ring0[w], s_ring1[w], s_ring2[D > 2 ? w : 0], s_ring3[D > 3 ? w : 0]};
    const uint32_t ring_lds[4] = {lds_offset(s_ring0[w]), lds_offset(s_ring1[w]), lds_offset(s_ring2[D > 2 ? w : 0]),
                                  lds_offset(s_ring3[D > 3 ? w : 0])};

    // one panel = four DMA instructions, always four (the waits below count them)
    auto issue = [&](const int64_t b, const int c, const uint32_t slot) {
        const uint8_t *base = q + b * 64 * stride + c * 64;                        // wave-uniform
        const bool edge = (b * 64 + 64 > n) || (c * 4 + 4 > row_chunks);          // wave-uniform
        if (!edge) {
#pragma unroll
            for (int rg = 0; rg < 4; rg++) nar_dma16(base, voff[rg], slot + rg * 1024);
        } else {
            // last row block of the batch / last chunk column of a row whose stride is not a multiple of 64: rows and chunks
            // clamped into the matrix (what they deliver is never looked at)
            const int last_row = (int)(n - 1 - b * 64);                            // >= 0: the block holds at least one read
            const int ch = min(cl, row_chunks - 1 - c * 4);                        // >= 0: the panel starts inside the row
#pragma unroll
            for (int rg = 0; rg < 4; rg++)
                nar_dma16(base, (uint32_t)(min(rg * 16 + r16, last_row) * (int)stride + ch * 16), slot + rg * 1024);
        }
    };

    int64_t pf_b = gw, cur_b = gw;            // row block of the next panel to request / being computed
    int pf_c = 0, cur_c = 0;
    int64_t pf = 0;                           // panels requested so far
    auto request = [&](const uint32_t slot) {
        if (pf < total) {
            issue(pf_b, pf_c, slot);
            pf++;
            if (++pf_c == ncq) { pf_c = 0; pf_b += W; }
        }
    };
#pragma unroll
    for (int k = 0; k < D - 1; k++) request(ring_lds[k]);
    // A panel's slot is free as soon as its 64 bytes per lane are in registers -- at the START of its step, not at the end: the
    // request that refills it is made right behind those reads, so D panels are in flight while one is computed on, not D - 1
    // (what a CU can have in flight is what bounds the stream, and LDS capacity is what bounds that: profiles/r05_narrow_variants.txt).
    request(ring_lds[D - 1]);

    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
    const uint32_t tl = (uint32_t)((lane >> 4) * 1024 + (lane & 15) * 16);      // this lane's row inside a panel
    const double thr = prm.thr;
    // 'N' bases of the lane's read: the bytes that are NOT zero are counted, four per instruction -- v_msad_u8 adds |a - b| over the
    // bytes whose reference byte b is non-zero, and a = b ^ 1 differs from b by exactly one -- which costs half an instruction
    // per base where looking for zero bytes would cost one
    uint32_t nonzero = 0;
    const int ndw = (li + 3) >> 2;                            // dwords counted per read (bytes past its end are made non-zero)

    auto step = [&](const int S, const int64_t s) {
        (void)s;
        // the panel of this step has landed when at most the requests made after it are still out
        const int64_t younger = pf - (s + 1);               // 0 .. D-1 panels (wave-uniform)
        {
        if (younger >= 3) nar_wait<12>();
        else if (younger == 2) nar_wait<8>();
        else if (younger == 1) nar_wait<4>();
        else nar_wait<0>();
        }
        const uint8_t *mine = ring[S] + tl;
        const int nbases = min(64, li - cur_c * 64);        // wave-uniform
        if (nbases == 64) {
            // All four chunks of the panel at once; the table two dwords (eight bases) ahead of the arithmetic; 1 - p one dword
            // ahead; and inside a base the operations in an order that keeps dependent FP64 instructions three issue slots apart
            // (a dependent v_mul_f64 / v_add_f64 issues 8-9 cycles after its producer, tools/experiments/fp64_latency.hip: back
            // to back it costs a slot): every product first, then the sums, then the next dword's 1 - p and the table address of
            // the base eight ahead.  The fences pin that order (the machine scheduler otherwise puts each sum right behind its
            // product).  Fully unrolled, so the rotating register roles cost no copies.
            uint32_t wd[16];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(mine + k * 256);
                wd[4 * k] = x.x; wd[4 * k + 1] = x.y; wd[4 * k + 2] = x.z; wd[4 * k + 3] = x.w;
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // the panel is in registers: its slot may be overwritten
            request(ring_lds[S]);
#define NAR_LOOKUP(w, t) s_p[((w) >> (8 * (t))) & 0xffu]
            nar_entry_t P[18][4];                       // P[d]: table entries of dword d (static indices only)
            double A[17][4];                            // A[d]: their 1 - p
#pragma unroll
            for (int t = 0; t < 4; t++) P[0][t] = NAR_LOOKUP(wd[0], t);
#pragma unroll
            for (int t = 0; t < 4; t++) P[1][t] = NAR_LOOKUP(wd[1], t);
#pragma unroll
            for (int t = 0; t < 4; t++) A[0][t] = NAR_A(P[0][t]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < 16; d++) {
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const double a = A[d][t], p = NAR_P(P[d][t]);
                    double x[R], y[R];
#pragma unroll
                    for (int r = R - 1; r >= 1; r--) x[r] = a * v[r];
#pragma unroll
                    for (int r = R - 1; r >= 1; r--) y[r] = p * v[r - 1];
                    v[0] = a * v[0];
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int r = R - 1; r >= 1; r--) v[r] = x[r] + y[r];
                    __builtin_amdgcn_sched_barrier(0);
                    if (d < 15) A[d + 1][t] = NAR_A(P[d + 1][t]);
                    if (d < 14) P[d + 2][t] = NAR_LOOKUP(wd[d + 2], t);
                    if (t == 0) nonzero = __builtin_amdgcn_msad_u8(wd[d] ^ 0x01010101u, wd[d], nonzero);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        } else {
            for (int k = 0; k * 16 < nbases; k++) {
                const uint4 x = *reinterpret_cast<const uint4 *>(mine + k * 256);
                uint32_t w0 = x.x, w1 = x.y, w2 = x.z, w3 = x.w;
                const int nb = min(16, nbases - k * 16);
                int d = 0;
#pragma unroll 1
                for (; d * 4 + 4 <= nb; d++) {
                    nonzero = __builtin_amdgcn_msad_u8(w0 ^ 0x01010101u, w0, nonzero);
                    nar_dword<R>(v, s_p, w0);
                    w0 = w1; w1 = w2; w2 = w3;
                }
                if (d * 4 < nb) {                           // a length that is not a multiple of 4: its last 1..3 bases
                    const uint32_t wf = w0 | (0x01010101u << (8 * (nb - d * 4)));     // bytes past the end: non-zero
                    nonzero = __builtin_amdgcn_msad_u8(wf ^ 0x01010101u, wf, nonzero);
                    for (int t = d * 4; t < nb; t++) {
                        nar_step<R>(v, s_p[w0 & 0xffu]);
                        w0 >>= 8;
                    }
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");          // (a partial panel is read chunk by chunk: free at its end)
            request(ring_lds[S]);
        }
        if (++cur_c == ncq) {
            // ---- a row block is done: sequential CDF, interpolation, predicate (as the tile classes' epilogue) ----
            const int64_t i = cur_b * 64 + lane;
            const bool valid = i < n;
            double acc = 0.0, lo = 0.0, hi = 0.0;
            int js = -1;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double na = acc + v[r];
                const bool hit = (js < 0) && (na > thr);
                lo = hit ? acc : lo;
                hi = hit ? na : hi;
                js = hit ? r : js;
                acc = na;
            }
            const bool done = valid && js >= 0;
            if (done) {
                double e = (double)(js - 1) + ((thr - lo) / (hi - lo));     // ref: bernoullimodule.c:170-178
                if (e < 0) e = 0;
                const int nsv = 4 * ndw - (int)nonzero;                      // 'N' bases (a read with an 'n' never gets here)
                if (prm.ambig_mode == 0) e = e + (double)nsv;                // moira.py:827-828
                const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors            // moira.py:925-926
                                                                      : (double)li * prm.uncert; // moira.py:949-950
                if (prm.flags & 1u) e = floor(e);                            // moira.py:830-831
                ee[i] = e;
                ns[i] = nsv;
                pass[i] = (uint8_t)((prm.ambig_mode == 2 && nsv > 0) ? 0 : (e <= limit ? 1 : 0));   // moira.py:911
            }
            const unsigned long long todo = __ballot(valid && js < 0);
            if (todo) {
                if (valid && js < 0) my_seg[nlist + __popcll(todo & ((1ull << lane) - 1ull))] = (int32_t)i;
                nlist += __popcll(todo);
            }
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
            nonzero = 0;
            cur_c = 0;
            cur_b += W;
        }
    };
    for (int64_t s = 0; s < total; s += D) {
        step(0, s);
        if (D > 1 && s + 1 < total) step(1 % D, s + 1);
        if (D > 2 && s + 2 < total) step(2 % D, s + 2);
        if (D > 3 && s + 3 < total) step(3 % D, s + 3);
    }
    if (lane == 0) wave_count[gw] = nlist;
}

// ------------------------------------------------------------------------------------------
// k_narrow_rs<R>: the narrow pass with the panel stream staged in REGISTERS and every request a whole, aligned 128-byte line
// (round 5, second form; rows whose stride is a multiple of 64 bytes).
//
// What k_narrow's measurements said (profiles/r05_narrow_variants.txt): its LDS-DMA stream stops at 0.78 ms per 10 M x 320 B
// whatever the ring depth, the resident workgroups or the bytes it over-fetches, while k_prepass / k_lambda pull the same
// matrix through plain loads in 0.58-0.64 ms; and a row of 320 bytes starts in the middle of a line every other row, so that
// 64-byte panels fetch a fifth to two fifths of the lines twice.  Both go away when a LANE walks a whole number of lines:
// k = 128 / gcd(stride, 128) consecutive reads (k = 2 for the 300-base / 320-byte layout) are one stream of k * stride bytes
// per lane, cut into 128-byte panels that are lines in memory.  A wave's 64 streams are loaded 8 streams x 128 bytes per
// instruction (eight lanes = one line) into registers, one panel (32 VGPRs) ahead of the arithmetic through a range-checked
// buffer (rows past the end of the matrix read zeros = 'N' = the identity step), written into a private 8 KB tile whose
// 16-byte slots are XOR-swizzled by the stream number (writes and reads both conflict-free, no padding: 16 waves per CU),
// and read back a stream per lane.  The reads of a lane follow each other in its stream: at every end of a read the same
// epilogue as k_narrow's; bytes between a read's end and the next read's start (the row padding) are skipped by whole
// 64-byte halves (wave-uniform) or never looked at.  Same cell arithmetic, same sequential CDF, same list of reads handed
// back (a wave's segment holds 64 k slots per stream block).
// ------------------------------------------------------------------------------------------
#define MPB_NRS_TILE 8192                   // 64 streams x 128 bytes

// ND dwords (4 ND bases) of the lane's stream, already in registers: the table two dwords ahead of the arithmetic, products
// before sums (k_narrow's unrolled body; see there).  One straight line: a branch inside it makes the compiler drain the LDS
// queue where the paths meet (measured: a loop over 16-base chunks with the same look-ahead is 8 % slower than this).
template <int R, int ND>
__device__ __forceinline__ void nar_run(double (&v)[R], uint32_t &nonzero, const nar_entry_t *s_p, const uint32_t (&wd)[16])
{
    nar_entry_t P[ND + 2][4];
    double A[ND + 1][4];
#pragma unroll
    for (int t = 0; t < 4; t++) P[0][t] = NAR_LOOKUP(wd[0], t);
#pragma unroll
    for (int t = 0; t < 4; t++) P[1][t] = NAR_LOOKUP(wd[1], t);
#pragma unroll
    for (int t = 0; t < 4; t++) A[0][t] = NAR_A(P[0][t]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int d = 0; d < ND; d++) {
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const double a = A[d][t], p = NAR_P(P[d][t]);
            double x[R], y[R];
#pragma unroll
            for (int r = R - 1; r >= 1; r--) x[r] = a * v[r];
#pragma unroll
            for (int r = R - 1; r >= 1; r--) y[r] = p * v[r - 1];
            v[0] = a * v[0];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = R - 1; r >= 1; r--) v[r] = x[r] + y[r];
            __builtin_amdgcn_sched_barrier(0);
            if (d < ND - 1) A[d + 1][t] = NAR_A(P[d + 1][t]);
            if (d < ND - 2) P[d + 2][t] = NAR_LOOKUP(wd[d + 2], t);
            if (t == 0) nonzero = __builtin_amdgcn_msad_u8(wd[d] ^ 0x01010101u, wd[d], nonzero);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

template <int R, bool ALIGNED>
__global__ __launch_bounds__(256) void k_narrow_rs(const uint8_t *__restrict__ q, int64_t n, int64_t stride, int32_t li, int32_t k,
                                                   MpbDevParams prm, const double2 *__restrict__ lut_g,
                                                   double *__restrict__ ee, int32_t *__restrict__ ns, uint8_t *__restrict__ pass,
                                                   int32_t *__restrict__ seg, int32_t *__restrict__ wave_count)
{
    __shared__ nar_entry_t s_p[256];
    __shared__ __attribute__((aligned(128))) uint8_t s_tile[4][MPB_NRS_TILE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_p[tid] = tid == 255 ? make_double2(__builtin_nan(""), __builtin_nan("")) : lut_g[tid];
    __syncthreads();                                          // the only block barrier
    const int64_t rows_sb = 64 * (int64_t)k;                  // reads of a stream block: 64 lanes x k reads each
    const int64_t nsb = (n + rows_sb - 1) / rows_sb;
    const int KB = __builtin_amdgcn_readfirstlane((int)(k * stride));      // bytes of a lane's stream: a multiple of 128
    const int NP = KB >> 7;                                   // its panels
    const int istride = __builtin_amdgcn_readfirstlane((int)stride);
    const int64_t gw = (int64_t)blockIdx.x * 4 + w, W = (int64_t)gridDim.x * 4;
    if (gw >= nsb) {
        if (lane == 0) wave_count[gw] = 0;
        return;
    }
    const int64_t total = ((nsb - gw + W - 1) / W) * NP;      // panels this wave walks
    int32_t *const my_seg = seg + rows_sb * (gw * (nsb / W) + min(gw, nsb % W));
    int nlist = 0;                                            // wave-uniform
    uint8_t *const tile = s_tile[w];
    // loading: lane (r8, c8) of instruction j holds the 16-byte slot c8 of panel bytes of stream 8 j + r8
    const int r8 = lane >> 3, c8 = lane & 7;
    const int voff = r8 * KB + c8 * 16;
    // its place in the tile: stream * 128 + ((slot ^ ((stream >> 1) & 7)) * 16); (8 j + r8) >> 1 & 7 = 4 (j & 1) + (r8 >> 1)
    const int wr_even = r8 * 128 + ((c8 ^ (r8 >> 1)) << 4), wr_odd = r8 * 128 + ((c8 ^ (4 + (r8 >> 1))) << 4);
    // reading: lane = stream; slot c at x0 ^ (c << 4)
    const int x0 = lane * 128 + (((lane >> 1) & 7) << 4);

    u32x4 pre[8];
    auto load_panel = [&](const int64_t sb, const int pk) {
        const uint64_t base = (uint64_t)(uintptr_t)q + (uint64_t)(sb * rows_sb) * (uint64_t)stride;
        const int64_t rows_here = (n - sb * rows_sb) < rows_sb ? (n - sb * rows_sb) : rows_sb;
        const uint32_t b_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)base);
        const uint32_t b_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(base >> 32));
        const uint32_t b_n = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(rows_here * stride));
        const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(
            (void *)(uintptr_t)(((uint64_t)b_hi << 32) | b_lo), 0, (int)b_n, 0x00020000);
#pragma unroll
        for (int j = 0; j < 8; j++) pre[j] = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, j * 8 * KB + pk * 128, 0);
    };
    int64_t pf_sb = gw, cur_sb = gw;
    int pf_pk = 0, cur_pk = 0;
    load_panel(pf_sb, pf_pk);
    if (++pf_pk == NP) { pf_pk = 0; pf_sb += W; }

    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
    const double thr = prm.thr;
    uint32_t nonzero = 0;                                      // as k_narrow: the bytes that are NOT zero, four per instruction
    int u = 0, sread = 0;                                      // position in the stream: byte u of its read number sread (wave-uniform)

    // ---- a read is done: sequential CDF, interpolation, predicate (as k_narrow / the tile classes' epilogue) ----
    // `nonzero` has counted the non-zero bytes of the read's chunks; the bytes of its last chunk past its end were made zero
    // before they were looked up (the identity step) and so count as 'N' here: 'N' bases = li - nonzero.
    // (two reads per lane, result arrays aligned for a lane's pair: see finish; R <= 3 only: with four rows the held results do
    // not fit the 128 registers of four waves per SIMD)
    const bool pair_stores = R <= 3 && k == 2 && (((uintptr_t)ee & 15) | ((uintptr_t)ns & 7) | ((uintptr_t)pass & 1)) == 0;
    double held_e = 0.0;
    int held_ns = 0;
    uint8_t held_ps = 0;
    bool held_ok = false;
    auto finish = [&](const int64_t sb, const int sr) {
        const int64_t i = sb * rows_sb + (int64_t)lane * k + sr;
        const bool valid = i < n;
        double acc = 0.0, lo = 0.0, hi = 0.0;
        int js = -1;
#pragma unroll
        for (int r = 0; r < R; r++) {
            const double na = acc + v[r];
            const bool hit = (js < 0) && (na > thr);
            lo = hit ? acc : lo;
            hi = hit ? na : hi;
            js = hit ? r : js;
            acc = na;
        }
        const bool done = valid && js >= 0;
        if (done) {
            double e = (double)(js - 1) + ((thr - lo) / (hi - lo));     // ref: bernoullimodule.c:170-178
            if (e < 0) e = 0;
            const int nsv = li - (int)nonzero;                           // 'N' bases (a read with an 'n' never gets here)
            if (prm.ambig_mode == 0) e = e + (double)nsv;                // moira.py:827-828
            const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors            // moira.py:925-926
                                                                  : (double)li * prm.uncert; // moira.py:949-950
            if (prm.flags & 1u) e = floor(e);                            // moira.py:830-831
            const uint8_t ps = (uint8_t)((prm.ambig_mode == 2 && nsv > 0) ? 0 : (e <= limit ? 1 : 0));   // moira.py:911
            if (pair_stores && sr == 0) {
                // two reads per lane: the first one's results wait in registers for the second's, and go out together -- 16 + 8 + 2
                // contiguous bytes per lane instead of two half-used sectors a panel and a half apart (writes 0.26 -> 0.13 GB)
                held_e = e; held_ns = nsv; held_ps = ps;
            } else if (pair_stores && held_ok) {
                *reinterpret_cast<double2 *>(ee + i - 1) = make_double2(held_e, e);
                *reinterpret_cast<int2 *>(ns + i - 1) = make_int2(held_ns, nsv);
                *reinterpret_cast<uint16_t *>(pass + i - 1) = (uint16_t)(held_ps | ((uint16_t)ps << 8));
            } else {
                ee[i] = e;
                ns[i] = nsv;
                pass[i] = ps;
            }
        }
        if (pair_stores) {
            if (sr == 0) held_ok = done;
            else if (held_ok && !done) { ee[i - 1] = held_e; ns[i - 1] = held_ns; pass[i - 1] = held_ps; }   // the second one is handed back (or past the end)
        }
        const unsigned long long todo = __ballot(valid && js < 0);
        if (todo) {
            if (valid && js < 0) my_seg[nlist + __popcll(todo & ((1ull << lane) - 1ull))] = (int32_t)i;
            nlist += __popcll(todo);
        }
#pragma unroll
        for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
        nonzero = 0;
    };
    for (int64_t t = 0; t < total; t++) {
        // the panel requested one panel ago -> tile (the tile's last reads were issued before: LDS runs a wave's operations in order)
#pragma unroll
        for (int j = 0; j < 8; j++)
            *reinterpret_cast<u32x4 *>(tile + j * 1024 + ((j & 1) ? wr_odd : wr_even)) = pre[j];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (t + 1 < total) {                                    // in flight while this panel is computed on
            load_panel(pf_sb, pf_pk);
            if (++pf_pk == NP) { pf_pk = 0; pf_sb += W; }
        }
        __builtin_amdgcn_sched_barrier(0);
        // ---- the panel's two 64-byte halves, four 16-byte chunks each.  The chunks of a half that belong to ONE read go through
        // one straight-line run (nar_run<R, 4 x chunks>); the bytes of a read's last chunk past its end are made zero first (the
        // identity step), so a 300-base read's last 44 bases take the same code as the others, as 48.  A read may end -- and the
        // next one begin -- anywhere a chunk does (strides that are no multiple of 64): then the half is several runs; row
        // padding is skipped by whole chunks.  All of it wave-uniform.
#pragma unroll
        for (int h = 0; h < 2; h++) {
            if (ALIGNED) {
                // rows of a multiple of 64 bytes: a read starts with a half, so a half holds chunks of ONE read -- one run, every
                // address a constant (the form measured in profiles/r05_narrow_variants.txt; the loop below costs it 2-3 %)
                const int nb = li - u;                          // bases of the current read from this half on
                if (nb > 0) {
                    const int nch = nb >= 64 ? 4 : (nb + 15) >> 4;
                    uint32_t wd[16];
#define NRS_LOADA(NC)                                                                                        \
                    _Pragma("unroll") for (int c = 0; c < NC; c++) {                                         \
                        const u32x4 x = *reinterpret_cast<const u32x4 *>(tile + (x0 ^ ((h * 4 + c) << 4)));  \
                        wd[4 * c] = x.x; wd[4 * c + 1] = x.y; wd[4 * c + 2] = x.z; wd[4 * c + 3] = x.w;       \
                    }                                                                                        \
                    if (nb < 16 * NC) {                                                                      \
                        _Pragma("unroll") for (int d = 0; d < 4; d++)                                        \
                            wd[4 * (NC - 1) + d] = mask_dword(wd[4 * (NC - 1) + d], nb - 16 * (NC - 1) - 4 * d); \
                    }
                    switch (nch) {
                    case 4: { NRS_LOADA(4) nar_run<R, 16>(v, nonzero, s_p, wd); break; }
                    case 3: { NRS_LOADA(3) nar_run<R, 12>(v, nonzero, s_p, wd); break; }
                    case 2: { NRS_LOADA(2) nar_run<R, 8>(v, nonzero, s_p, wd); break; }
                    default: { NRS_LOADA(1) nar_run<R, 4>(v, nonzero, s_p, wd); break; }
                    }
#undef NRS_LOADA
                    if (nb <= 64) finish(cur_sb, sread);
                }
                u += 64;
                if (u == istride) { u = 0; sread++; }
                continue;
            }
            int p = 0;                                          // chunk of this half
            while (p < 4) {
                const int nb = li - u;                          // bases of the current read from here on
                if (nb <= 0) {                                  // its padding: on to the next read, or to the end of the half
                    const int skip = min((istride - u) >> 4, 4 - p);
                    p += skip;
                    u += 16 * skip;
                } else {
                    const int nch = min((nb + 15) >> 4, 4 - p); // chunks of this read in what is left of the half
                    const int c0 = h * 4 + p;
                    uint32_t wd[16];
#define NRS_LOAD(NC)                                                                                         \
                    _Pragma("unroll") for (int c = 0; c < NC; c++) {                                         \
                        const u32x4 x = *reinterpret_cast<const u32x4 *>(tile + (x0 ^ ((c0 + c) << 4)));     \
                        wd[4 * c] = x.x; wd[4 * c + 1] = x.y; wd[4 * c + 2] = x.z; wd[4 * c + 3] = x.w;       \
                    }                                                                                        \
                    if (nb < 16 * NC) {                                                                      \
                        _Pragma("unroll") for (int d = 0; d < 4; d++)                                        \
                            wd[4 * (NC - 1) + d] = mask_dword(wd[4 * (NC - 1) + d], nb - 16 * (NC - 1) - 4 * d); \
                    }
                    switch (nch) {
                    case 4: { NRS_LOAD(4) nar_run<R, 16>(v, nonzero, s_p, wd); break; }
                    case 3: { NRS_LOAD(3) nar_run<R, 12>(v, nonzero, s_p, wd); break; }
                    case 2: { NRS_LOAD(2) nar_run<R, 8>(v, nonzero, s_p, wd); break; }
                    default: { NRS_LOAD(1) nar_run<R, 4>(v, nonzero, s_p, wd); break; }
                    }
#undef NRS_LOAD
                    p += nch;
                    u += 16 * nch;
                    if (16 * nch >= nb) finish(cur_sb, sread);  // the read is done (u may stand in its padding now)
                }
                if (u >= istride) { u = 0; sread++; }
            }
        }
        if (++cur_pk == NP) { cur_pk = 0; cur_sb += W; sread = 0; }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is overwritten by the next panel
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    }
    if (lane == 0) wave_count[gw] = nlist;
}

// ------------------------------------------------------------------------------------------
// k_narrow_rg<R>: the narrow pass for RAGGED batches (round 6: one padded matrix + int32 len[], e.g. the contigs the
// reference's paired mode produces, moira/moira.py:789-801,1376-1558).
//
// A wave's 64 lanes run in lock step, so a wave is as slow as its longest read; and a padded row is mostly padding, so the
// matrix must not be read as it lies (a stride-640 matrix of U{50..600} reads is 1.9 x its bases).  Both go away when a wave
// takes 64 reads of (nearly) ONE length and fetches only the lines their bases lie in:
//   k_rag_sort   windows of MPB_RG_WIN consecutive reads are counting-sorted by ceil(len / 16) (stable, in LDS, one workgroup
//                per window: 8 bytes per read in, 8 out); 64 consecutive entries of the sorted order are a GROUP, whose cost is
//                its longest read's 16-byte chunks.  A window is small enough that a group's rows lie within 4096 rows of
//                each other, large enough that a group straddles a length boundary about as often as not (<= 1 chunk wasted);
//   k_rag_scan   prefix sums of the windows' costs (one small block); the list of groups is cut into one contiguous range per
//                wave of the persistent grid, equal in cost -- each wave finds its own borders (rg_first_group; deterministic);
//   k_narrow_rg  a wave walks its groups.  Memory path as k_narrow_rs: eight lanes fetch one 128-byte piece of one row per
//                instruction (per-lane row offsets from a wave-uniform window base: the rows of a group are a gather), eight
//                instructions = one panel of the 64 rows, staged in registers one panel ahead -- across group borders: the next
//                group's order entries are fetched two groups ahead -- written to the wave's private XOR-swizzled tile and read
//                back a row per lane.  Only chunks below the group's longest read are requested.  Arithmetic: the same
//                straight-line runs per 64-byte half (nar_run); chunks that are complete in every lane take them unmasked, the
//                one or two last chunks are masked per lane to the identity step (zero bytes) by the lane's own length.
//                Epilogue per group, with the lane's own length in the predicate (moira.py:949-950).  Reads the pass cannot finish
//                (more rows, a lower-case 'n', a length outside 0..max_len: left to the sorted pipeline's checks) are listed
//                in the wave's own segment (64 slots per group it owns).
// Rows of a multiple of 128 bytes are fetched in whole lines; other strides work (a line shared by two rows is fetched twice).
// ------------------------------------------------------------------------------------------
#define MPB_RG_WIN 4096                     // reads per sort window (64 groups)
#define MPB_RG_BINS 64                      // sort keys per window: ceil(len / 16) >> key_shift

// cost of a group = its chunks + a panel's fixed work per 8 chunks + the epilogue (in chunk units; it balances, nothing else)
__device__ __forceinline__ int rg_group_cost(int chunks) { return chunks + ((chunks + 7) >> 3) + 2; }

// exclusive prefix over the 256 threads of a block (wave scans by lane shuffles, the four wave totals through LDS)
__device__ __forceinline__ uint32_t rg_block_excl_scan(uint32_t x, uint32_t *s_wave, int tid)
{
    const int lane = tid & 63, w = tid >> 6;
    uint32_t incl = x;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const uint32_t y = __shfl_up(incl, off);
        if (lane >= off) incl += y;
    }
    if (lane == 63) s_wave[w] = incl;
    __syncthreads();
    uint32_t base = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) base += k < w ? s_wave[k] : 0;
    __syncthreads();                                            // (s_wave is used again by the next pass)
    return base + incl - x;
}

// One window: a stable LSD radix sort of its <= 4096 reads by the 6-bit key ceil(len / 16) >> key_shift, two passes of three bits.
// A pass: thread t counts the digits of its 16 consecutive elements into its own column of an 8 x 256 counter matrix (in registers
// first: no chain of dependent LDS updates, no atomics), the matrix is scanned digit-major (a thread's 8 counters are 16 contiguous
// bytes: one b128 read, one write; the 256 partial sums by lane shuffles), and the thread places its elements in order.  28 KB of
// LDS: five workgroups per CU (a block's life is a chain of LDS latencies and barriers, so what counts is how many run at once).
__global__ __launch_bounds__(256) void k_rag_sort(const int32_t *__restrict__ len, int64_t n, int32_t max_len, int key_shift,
                                                  int32_t split, int32_t low_pct,
                                                  int2 *__restrict__ ord, int32_t *__restrict__ gpre,
                                                  unsigned long long *__restrict__ wsum)
{
    __shared__ __attribute__((aligned(16))) uint16_t s_cnt[8 * 256];
    __shared__ __attribute__((aligned(16))) uint16_t s_idx[2][MPB_RG_WIN];     // pass 1: local index | high digit << 12; pass 2: local index
    __shared__ __attribute__((aligned(16))) uint16_t s_len[MPB_RG_WIN];        // lengths in natural order (0xffff: outside 0..max_len)
    __shared__ uint32_t s_wave[4];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t w0 = (int64_t)blockIdx.x * MPB_RG_WIN;
    const int m = (int)min((int64_t)MPB_RG_WIN, n - w0);
    // the window's lengths, loaded a line per 32 lanes (thread-contiguous loads of 16 ints cost a request per LANE), checked,
    // parked in natural order (max_len <= MPB_RG_MAX_STRIDE = 4096: 16 bits hold them)
#pragma unroll
    for (int k = 0; k < 16; k++) {
        const int i = k * 256 + tid;
        const int l = i < m ? len[w0 + i] : 0;
        s_len[i] = (uint16_t)((l < 0 || l > max_len) ? 0xffff : l);        // (a length outside the row: key 0, marked for the pass)
    }
    __syncthreads();
    auto key_of = [key_shift](uint32_t l) { return l == 0xffffu ? 0u : ((l + 15u) >> 4) >> key_shift; };
    // thread t owns elements 16 t .. 16 t + 15 of the current order (pass 1: the natural order)
    uint32_t el[16];                                            // pass 1: the key; pass 2: local index | high digit << 12
    {
        const u32x4 a = *reinterpret_cast<const u32x4 *>(s_len + 16 * tid), b = *reinterpret_cast<const u32x4 *>(s_len + 16 * tid + 8);
        const uint32_t ww[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
        for (int e = 0; e < 8; e++) { el[2 * e] = key_of(ww[e] & 0xffffu); el[2 * e + 1] = key_of(ww[e] >> 16); }
    }
    const int mine = min(16, max(0, m - 16 * tid));             // elements of this thread that exist
#pragma unroll
    for (int pass = 0; pass < 2; pass++) {
        if (pass == 1) {
            __syncthreads();                                    // pass 1's order is complete
            const u32x4 a = *reinterpret_cast<const u32x4 *>(s_idx[0] + 16 * tid), b = *reinterpret_cast<const u32x4 *>(s_idx[0] + 16 * tid + 8);
            const uint32_t ww[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int e = 0; e < 8; e++) { el[2 * e] = ww[e] & 0xffffu; el[2 * e + 1] = ww[e] >> 16; }
        }
        // the thread's eight digit counts, in registers (16-bit fields of two 64-bit words)
        unsigned long long c_lo = 0, c_hi = 0;
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = pass == 0 ? (int)(el[r] & 7u) : (int)(el[r] >> 12);
            const unsigned long long inc = r < mine ? 1ull << (16 * (d & 3)) : 0ull;
            c_lo += (d & 4) ? 0ull : inc;
            c_hi += (d & 4) ? inc : 0ull;
        }
#pragma unroll
        for (int d = 0; d < 8; d++) s_cnt[d * 256 + tid] = (uint16_t)(((d & 4) ? c_hi : c_lo) >> (16 * (d & 3)));   // the thread's own column
        __syncthreads();
        // digit-major exclusive scan: thread t holds counters 8 t .. 8 t + 7 (one digit, eight threads)
        u32x4 c4 = *reinterpret_cast<const u32x4 *>(s_cnt + 8 * tid);
        uint32_t cc[8] = {c4.x & 0xffffu, c4.x >> 16, c4.y & 0xffffu, c4.y >> 16, c4.z & 0xffffu, c4.z >> 16, c4.w & 0xffffu, c4.w >> 16};
        uint32_t sum = 0;
#pragma unroll
        for (int e = 0; e < 8; e++) sum += cc[e];
        uint32_t run = rg_block_excl_scan(sum, s_wave, tid);
#pragma unroll
        for (int e = 0; e < 8; e++) { const uint32_t c = cc[e]; cc[e] = run; run += c; }
        c4.x = cc[0] | (cc[1] << 16); c4.y = cc[2] | (cc[3] << 16); c4.z = cc[4] | (cc[5] << 16); c4.w = cc[6] | (cc[7] << 16);
        *reinterpret_cast<u32x4 *>(s_cnt + 8 * tid) = c4;
        __syncthreads();
        unsigned long long p_lo = 0, p_hi = 0;                  // where the thread's next element of each digit goes
#pragma unroll
        for (int d = 0; d < 8; d++) {
            const unsigned long long v = (unsigned long long)s_cnt[d * 256 + tid] << (16 * (d & 3));
            if (d & 4) p_hi |= v; else p_lo |= v;
        }
#pragma unroll
        for (int r = 0; r < 16; r++) {
            const int d = pass == 0 ? (int)(el[r] & 7u) : (int)(el[r] >> 12);
            const uint32_t pos = (uint32_t)(((d & 4) ? p_hi : p_lo) >> (16 * (d & 3))) & 0xffffu;
            const unsigned long long inc = 1ull << (16 * (d & 3));
            p_lo += (d & 4) ? 0ull : inc;
            p_hi += (d & 4) ? inc : 0ull;
            if (r < mine) s_idx[pass][pos] = (uint16_t)(pass == 0 ? (uint32_t)(16 * tid + r) | ((el[r] >> 3) << 12) : el[r] & 0xfffu);
        }
    }
    __syncthreads();
    for (int p = tid; p < m; p += 256) {
        const int i = s_idx[1][p];
        const uint32_t l = s_len[i];
        ord[w0 + p] = make_int2((int)(w0 + i), l == 0xffffu ? -1 : (int)l);
    }
    // A group's cost comes from the 16-byte chunks of its longest read.  The order is ascending in the key, so that is the key of the
    // group's last entry (its upper edge when keys are coarser than a chunk: the cost only balances the waves' ranges; the pass
    // finds a group's exact longest read itself).  gpre[g] = cost of the window's groups up to and with g, wsum[window] = the total.
    if (w == 0) {
        int c = 0;
        if (lane * 64 < m) {
            const uint32_t k = key_of(s_len[s_idx[1][min(m, lane * 64 + 64) - 1]]);
            const int chunks = (int)(key_shift ? ((k + 1) << key_shift) - 1 : k);
            c = rg_group_cost(chunks) * 8;                      // (eighths, so that the short groups' discount has a resolution)
            if (chunks <= split) c = c * low_pct / 100;         // mixed rows: a group that runs with a row less is that much cheaper
        }
        int pre = c;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int x = __shfl_up(pre, off);
            if (lane >= off) pre += x;
        }
        if (lane * 64 < m) gpre[(w0 >> 6) + lane] = pre;
        if (lane == 63) wsum[blockIdx.x] = (unsigned long long)pre;
    }
}

// exclusive prefix of the windows' costs (one block): wpre[0 .. nwin], wpre[nwin] = the batch's total
__global__ __launch_bounds__(1024) void k_rag_scan(const unsigned long long *__restrict__ wsum, int nwin,
                                                   unsigned long long *__restrict__ wpre)
{
    __shared__ unsigned long long s_sum[1024];
    const int tid = threadIdx.x;
    const int per = (nwin + 1023) / 1024;
    const int a = min(nwin, tid * per), b = min(nwin, a + per);
    unsigned long long sum = 0;
    for (int k = a; k < b; k++) sum += wsum[k];
    s_sum[tid] = sum;
    __syncthreads();
    for (int off = 1; off < 1024; off <<= 1) {
        const unsigned long long x = tid >= off ? s_sum[tid - off] : 0;
        __syncthreads();
        s_sum[tid] += x;
        __syncthreads();
    }
    unsigned long long run = s_sum[tid] - sum;
    for (int k = a; k < b; k++) { wpre[k] = run; run += wsum[k]; }
    if (tid == 1023) wpre[nwin] = s_sum[1023];
}

// The groups are cut into one contiguous range per wave of the persistent grid, equal in cost: wave w of W starts at the first
// group whose inclusive cost prefix exceeds floor(w T / W).  Every wave finds its own two borders (a 64-ary search over the
// windows' prefixes, then the 64 groups of one window: three or four dependent loads), so there is no serial plan pass.
__device__ __forceinline__ int rg_first_group(const unsigned long long *__restrict__ wpre, const int32_t *__restrict__ gpre,
                                              int nwin, int ngroups, int wv, int nwaves, int lane)
{
    if (wv >= nwaves) return ngroups;
    const unsigned long long T = wpre[nwin];
    const unsigned long long target = T * (unsigned long long)wv / (unsigned long long)nwaves;   // (T < 2^35, wv < 2^13)
    int lo = 0, hi = nwin;                                     // wpre[lo] <= target < wpre[hi]
    while (hi - lo > 1) {
        const int step = (hi - lo + 63) >> 6;
        const int idx = lo + lane * step;
        const bool ok = idx < hi && wpre[idx] <= target;
        const int k = __popcll(__ballot(ok)) - 1;              // lanes 0 .. k hold a prefix <= target (lane 0 always does)
        lo += k * step;
        hi = min(hi, lo + step);
    }
    const unsigned long long rem = target - wpre[lo];
    const int ng = min(64, ngroups - lo * 64);
    const bool over = lane < ng && (unsigned long long)gpre[lo * 64 + lane] > rem;
    const unsigned long long m = __ballot(over);               // never empty: the window's total exceeds rem
    return lo * 64 + (m ? __builtin_ctzll(m) : ng - 1);
}

// RLO < R (round 6, "mixed rows"): a group whose longest read has at most `split` chunks runs with RLO rows, the others with R --
// how many rows a read of a given quality needs grows with its length (a clean read needs the third row from about 400 bases on), the
// groups are sorted by length, and a row less is 3 of a base's 7 FP64 operations.  `split` comes from the batch's sample (the shortest
// sampled read that needs R rows, less a margin); a read of a short group that needs R rows after all is handed back like any other.
template <int R, int RLO>
__global__ __launch_bounds__(256, 4) void k_narrow_rg(const uint8_t *__restrict__ q, int64_t n, int64_t stride,
                                                   const int2 *__restrict__ ord, const unsigned long long *__restrict__ wpre,
                                                   const int32_t *__restrict__ gpre, int32_t *__restrict__ gstart, int32_t split,
                                                   MpbDevParams prm, const double2 *__restrict__ lut_g,
                                                   double *__restrict__ ee, int32_t *__restrict__ ns, uint8_t *__restrict__ pass,
                                                   int32_t *__restrict__ seg, int32_t *__restrict__ wave_count)
{
    __shared__ nar_entry_t s_p[256];
    __shared__ __attribute__((aligned(128))) uint8_t s_tile[4][MPB_NRS_TILE];
    __shared__ uint32_t s_row[4][64];                         // byte offsets of the rows of the group being loaded, from its window's base
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    s_p[tid] = tid == 255 ? make_double2(__builtin_nan(""), __builtin_nan("")) : lut_g[tid];
    __syncthreads();                                          // the only block barrier
    const int gw = blockIdx.x * 4 + w;
    const int ngroups = (int)((n + 63) >> 6), nwin = (int)((n + MPB_RG_WIN - 1) / MPB_RG_WIN), nwaves = (int)gridDim.x * 4;
    const int g0 = __builtin_amdgcn_readfirstlane(rg_first_group(wpre, gpre, nwin, ngroups, gw, nwaves, lane));
    const int g1 = __builtin_amdgcn_readfirstlane(rg_first_group(wpre, gpre, nwin, ngroups, gw + 1, nwaves, lane));
    if (lane == 0) gstart[gw] = g0;                           // (where the wave's list segment starts: k_nar_compact)
    if (g0 >= g1) {
        if (lane == 0) wave_count[gw] = 0;
        return;
    }
    int32_t *const my_seg = seg + 64 * (int64_t)g0;
    int nlist = 0;                                            // wave-uniform
    uint8_t *const tile = s_tile[w];
    const int istride = __builtin_amdgcn_readfirstlane((int)stride);
    // loading: lane (r8, c8) of instruction j holds the 16-byte slot c8 of the panel of stream 8 j + r8; tile layout as k_narrow_rs
    const int r8 = lane >> 3, c8 = lane & 7;
    const int wr_even = r8 * 128 + ((c8 ^ (r8 >> 1)) << 4), wr_odd = r8 * 128 + ((c8 ^ (4 + (r8 >> 1))) << 4);
    const int x0 = lane * 128 + (((lane >> 1) & 7) << 4);

    // a group's order entries {read, length}: -1 / -1 past the batch or past the wave's range; length -1: outside 0..max_len
    auto fetch = [&](const int g, int &idx, int &ln) {
        const int64_t p = (int64_t)g * 64 + lane;
        unsigned long long e = ~0ull;                           // {-1, -1}
        if (g < g1 && p < n) e = gload(reinterpret_cast<const unsigned long long *>(ord) + p);
        idx = (int)(uint32_t)e; ln = (int)(uint32_t)(e >> 32);
    };
    int cur_idx, cur_len, nx_idx, nx_len, nn_idx, nn_len;
    fetch(g0, cur_idx, cur_len);
    fetch(g0 + 1, nx_idx, nx_len);
    fetch(g0 + 2, nn_idx, nn_len);

    // arming a group for loading: per-lane row offsets from the window's base, the group's chunks (longest read) and the
    // chunks complete in every lane (shortest)
    uint32_t *const rows = s_row[w];
    const uint8_t *wbase = q;
    int ld_maxc = 0, ld_full = 0;
    auto arm = [&](const int g, const int idx, const int ln) {
        const int64_t wrow = ((int64_t)g * 64) & ~(int64_t)(MPB_RG_WIN - 1);          // first row of the group's window
        wbase = q + wrow * stride;
        const bool good = idx >= 0 && ln >= 0;
        const int rowoff = idx >= 0 ? (int)(idx - wrow) * istride : 0;                // (rows past the batch: the window's first)
        int mx = good ? (ln + 15) >> 4 : 0, mn = good ? ln >> 4 : 0x7fffffff;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { mx = max(mx, __shfl_xor(mx, off)); mn = min(mn, __shfl_xor(mn, off)); }
        ld_maxc = __builtin_amdgcn_readfirstlane(mx);
        ld_full = __builtin_amdgcn_readfirstlane(mn);
        rows[lane] = (uint32_t)rowoff;                          // (the loads of the group before this one have all been issued)
    };
    u32x4 pre[8];
    auto load_panel = [&](const int pk) {
        const uint8_t *pb = wbase + pk * 128;                   // wave-uniform
        uint32_t voff[8];
#pragma unroll
        for (int j = 0; j < 8; j++) voff[j] = rows[8 * j + r8] + (uint32_t)(c8 * 16);
        if (8 * pk + c8 < ld_maxc) {                            // (the group's last panel: only the chunks its longest read has)
#pragma unroll
            for (int j = 0; j < 8; j++) pre[j] = *(const __attribute__((address_space(1))) u32x4 *)(pb + voff[j]);
        }
    };
    arm(g0, cur_idx, cur_len);
    int cur_maxc = ld_maxc, cur_full = ld_full;
    load_panel(0);

    double v[R];
#pragma unroll
    for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
    const double thr = prm.thr;
    uint32_t nonzero = 0;

    for (int g = g0; g < g1; g++) {
        const int np = __builtin_amdgcn_readfirstlane(max(1, (cur_maxc + 7) >> 3));      // panels of this group
        for (int pk = 0; pk < np; pk++) {
#pragma unroll
            for (int j = 0; j < 8; j++)
                *reinterpret_cast<u32x4 *>(tile + j * 1024 + ((j & 1) ? wr_odd : wr_even)) = pre[j];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            {
                int next_pk = pk + 1;
                const bool next_group = next_pk == np && g + 1 < g1;
                if (next_group) { arm(g + 1, nx_idx, nx_len); next_pk = 0; }
                if (next_pk < np || next_group) load_panel(next_pk);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int cb = 8 * pk + 4 * h;                   // first chunk of this half
                const int rem = cur_maxc - cb;                   // chunks of the group's longest read from here on
                if (rem <= 0) continue;
                const int fullc = cur_full - cb;                 // ... that are complete in every lane
                const int nbl = cur_len - 16 * cb;               // this lane's bases from here on (may be <= 0)
                uint32_t wd[16];
#define NRG_LOAD(NC)                                                                                         \
                _Pragma("unroll") for (int c = 0; c < NC; c++) {                                             \
                    const u32x4 x = *reinterpret_cast<const u32x4 *>(tile + (x0 ^ ((h * 4 + c) << 4)));      \
                    wd[4 * c] = x.x; wd[4 * c + 1] = x.y; wd[4 * c + 2] = x.z; wd[4 * c + 3] = x.w;           \
                    if (c >= fullc) {                                                                        \
                        _Pragma("unroll") for (int d = 0; d < 4; d++)                                        \
                            wd[4 * c + d] = mask_dword(wd[4 * c + d], nbl - 16 * c - 4 * d);                 \
                    }                                                                                        \
                }
                if (RLO < R && cur_maxc <= split) {               // a short group: rows 0 .. RLO-1 only (v[RLO ..] stay zero)
                    double (&vl)[RLO] = *reinterpret_cast<double (*)[RLO]>(&v[0]);
                    switch (rem >= 4 ? 4 : rem) {
                    case 4: { NRG_LOAD(4) nar_run<RLO, 16>(vl, nonzero, s_p, wd); break; }
                    case 3: { NRG_LOAD(3) nar_run<RLO, 12>(vl, nonzero, s_p, wd); break; }
                    case 2: { NRG_LOAD(2) nar_run<RLO, 8>(vl, nonzero, s_p, wd); break; }
                    default: { NRG_LOAD(1) nar_run<RLO, 4>(vl, nonzero, s_p, wd); break; }
                    }
                } else {
                    switch (rem >= 4 ? 4 : rem) {
                    case 4: { NRG_LOAD(4) nar_run<R, 16>(v, nonzero, s_p, wd); break; }
                    case 3: { NRG_LOAD(3) nar_run<R, 12>(v, nonzero, s_p, wd); break; }
                    case 2: { NRG_LOAD(2) nar_run<R, 8>(v, nonzero, s_p, wd); break; }
                    default: { NRG_LOAD(1) nar_run<R, 4>(v, nonzero, s_p, wd); break; }
                    }
                }
#undef NRG_LOAD
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the tile is overwritten by the next panel
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        // ---- the group is done: sequential CDF, interpolation, predicate (as k_narrow / the tile classes' epilogue) ----
        {
            const int64_t i = cur_idx;
            const int li = cur_len;
            const bool valid = cur_idx >= 0, good = valid && li >= 0;
            double acc = 0.0, lo = 0.0, hi = 0.0;
            int js = -1;
#pragma unroll
            for (int r = 0; r < R; r++) {
                const double na = acc + v[r];
                const bool hit = (js < 0) && (na > thr);
                lo = hit ? acc : lo;
                hi = hit ? na : hi;
                js = hit ? r : js;
                acc = na;
            }
            const bool done = good && js >= 0;
            if (done) {
                double e = (double)(js - 1) + ((thr - lo) / (hi - lo));     // ref: bernoullimodule.c:170-178
                if (e < 0) e = 0;
                const int nsv = li - (int)nonzero;                           // 'N' bases (a read with an 'n' never gets here)
                if (prm.ambig_mode == 0) e = e + (double)nsv;                // moira.py:827-828
                const double limit = (prm.maxerrors == prm.maxerrors) ? prm.maxerrors            // moira.py:925-926
                                                                      : (double)li * prm.uncert; // moira.py:949-950
                if (prm.flags & 1u) e = floor(e);                            // moira.py:830-831
                ee[i] = e;
                ns[i] = nsv;
                pass[i] = (uint8_t)((prm.ambig_mode == 2 && nsv > 0) ? 0 : (e <= limit ? 1 : 0));   // moira.py:911
            }
            const unsigned long long todo = __ballot(valid && !done);
            if (todo) {
                if (valid && !done) my_seg[nlist + __popcll(todo & ((1ull << lane) - 1ull))] = (int32_t)i;
                nlist += __popcll(todo);
            }
#pragma unroll
            for (int r = 0; r < R; r++) v[r] = r == 0 ? 1.0 : 0.0;
            nonzero = 0;
        }
        cur_idx = nx_idx; cur_len = nx_len;
        nx_idx = nn_idx; nx_len = nn_len;
        cur_maxc = __builtin_amdgcn_readfirstlane(ld_maxc); cur_full = __builtin_amdgcn_readfirstlane(ld_full);
        fetch(g + 3, nn_idx, nn_len);
    }
    if (lane == 0) wave_count[gw] = nlist;
}

// The waves' list segments -> the dense list, in wave order: block g sums the counts of the waves before it (at most 8192 ints,
// from L2), copies its wave's segment behind them, and the last block leaves the total in *count.  One launch; no atomics, so the
// list's order does not depend on timing.  seg_start: where wave g's segment begins -- per_blk x (row blocks owned by the waves
// before it) for the fixed-length forms (gstart == nullptr), 64 x gstart[g] for the ragged pass.
__global__ __launch_bounds__(256) void k_nar_compact(const int32_t *__restrict__ seg, const int32_t *__restrict__ wave_count,
                                                     int64_t nblk, int nwaves, int per_blk, const int32_t *__restrict__ gstart,
                                                     int32_t *__restrict__ list, int32_t *__restrict__ count)
{
    __shared__ int s_part[4];
    const int g = blockIdx.x, tid = threadIdx.x;
    int sum = 0;
    for (int k = tid; k < g; k += 256) sum += wave_count[k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((tid & 63) == 0) s_part[tid >> 6] = sum;
    __syncthreads();
    const int off0 = s_part[0] + s_part[1] + s_part[2] + s_part[3];
    const int cnt = wave_count[g];
    const int32_t *src = gstart ? seg + 64 * (int64_t)gstart[g]
                                : seg + (int64_t)per_blk * ((int64_t)g * (nblk / nwaves) + min((int64_t)g, nblk % nwaves));
    int32_t *dst = list + off0;
    for (int k = tid; k < cnt; k += 256) dst[k] = src[k];
    if (g == nwaves - 1 && tid == 0) *count = off0 + cnt;
}

// ------------------------------------------------------------------------------------------
// k_sample: which pass does this batch take?  `n_sample` reads spread evenly (with a hashed offset inside each stride) over
// the batch, one wave per read.  Whether a read's CDF crosses 1 - alpha within its first 2, 3 or 4 rows is what the choice
// hangs on, and there the prepass' Cornish-Fisher quantile is at its worst (a clean 300-base read has 0.08 expected errors:
// it predicts 3 rows where 2 do).  So the first rows are computed, not predicted: with r = p / (1 - p) the probability of
// exactly k errors is P0 * e_k(r_1 .. r_L), P0 = prod (1 - p_i), e_k the elementary symmetric polynomials, which come from
// the power sums T1 = sum r, T2 = sum r^2, T3 = sum r^3 by Newton's identities -- four fp32 sums per read (log P0, T1, T2, T3).
// A read that needs more than 4 rows is binned by the prepass' prediction (it only feeds the cost estimate of the sorted
// pipeline).  fp32 throughout: the sample steers speed, never a result.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_sample(const uint8_t *__restrict__ q, int64_t n, int64_t stride, int32_t fixed_len,
                                                const int32_t *__restrict__ len, MpbDevParams prm, int n_sample,
                                                int32_t *__restrict__ hist)
{
    __shared__ float2 s_tab[256];
    __shared__ float4 s_pow[256];                 // {log(1 - p), r, r^2, r^3}; ambiguous bases: zeros
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63;
    {
        const bool amb = tid == 0 || tid == 255;
        float p = __builtin_amdgcn_exp2f(-0.33219281f * (float)tid);      // 10^(-q/10)
        p = amb ? 0.0f : p;
        s_tab[tid] = make_float2(p, tid == 0 ? MPB_MARK_UPPER : tid == 255 ? MPB_MARK_LOWER : p * (1.0f - p));
        const float r = p / (1.0f - p);
        s_pow[tid] = make_float4(log1pf(-p), r, r * r, r * r * r);
    }
    __syncthreads();
    const int k = blockIdx.x * 4 + w;
    if (k >= n_sample) return;
    const int64_t step = n / n_sample;                                     // >= 1 (the host sees to it)
    const int64_t i = min(n - 1, (int64_t)k * step + (int64_t)(mpb_mix64((uint64_t)k) % (uint64_t)step));
    // ragged batches: the read's own length (one outside its row counts as a read the pass hands back); the histogram is then
    // weighted by 16-byte chunks, because a long read costs either pass more than a short one
    const int li_raw = len ? len[i] : fixed_len;
    const bool bad_len = li_raw < 0 || li_raw > prm.max_len;
    const int li = bad_len ? 0 : li_raw;
    float mu = 0.f, var = 0.f, k3 = 0.f;
    float lp0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
    int ambi = 0, lower = 0;                                               // ambiguous bases in all / lower-case 'n' among them
    for (int c0 = 0; c0 * 16 < li; c0 += 64 * 15) {                        // panels of 15 chunks per lane: the markers peel exactly
        f32x2 a01 = {0.f, 0.f};
        float s3 = 0.f;
        for (int c = c0 + lane; c * 16 < li && c < c0 + 64 * 15; c += 64) {
            uint4 y = *reinterpret_cast<const uint4 *>(q + i * stride + (int64_t)c * 16);
            const int nv = li - c * 16;
            y.x = fill_dword(y.x, nv); y.y = fill_dword(y.y, nv - 4);
            y.z = fill_dword(y.z, nv - 8); y.w = fill_dword(y.w, nv - 12);
            pre_chunk(s_tab, y, a01, s3);
            const uint32_t ww[4] = {y.x, y.y, y.z, y.w};
#pragma unroll
            for (int d = 0; d < 4; d++)
#pragma unroll
                for (int t = 0; t < 4; t++) {
                    const float4 e = s_pow[(ww[d] >> (8 * t)) & 0xffu];    // the fill byte (Q254) contributes 4e-26: nothing
                    lp0 += e.x; t1 += e.y; t2 += e.z; t3 += e.w;
                }
        }
        const float n255 = floorf(a01.y * (1.0f / MPB_MARK_LOWER));
        const float rem = a01.y - MPB_MARK_LOWER * n255;
        const float nzero = floorf(rem * (1.0f / MPB_MARK_UPPER));
        const float pvar = rem - MPB_MARK_UPPER * nzero;
        mu += a01.x; var += pvar; k3 += pvar - 2.0f * s3;
        ambi += (int)nzero + (int)n255;
        lower += (int)n255;
    }
#pragma unroll
    for (int off = 1; off <= 32; off <<= 1) {
        mu += __shfl_xor(mu, off); var += __shfl_xor(var, off);
        k3 += __shfl_xor(k3, off); ambi += __shfl_xor(ambi, off); lower += __shfl_xor(lower, off);
        lp0 += __shfl_xor(lp0, off); t1 += __shfl_xor(t1, off);
        t2 += __shfl_xor(t2, off); t3 += __shfl_xor(t3, off);
    }
    if (lane == 0) {
        const float vv = fmaxf(var, 1e-12f);
        const float x = mu + prm.z * sqrtf(vv) + (k3 / vv) * prm.zq;      // as class_read
        int rows = (int)floorf(fminf(x, 1e9f) + 0.5f) + 1;
        rows = max(rows, 5);                                               // rows 1..4 are decided below
        // the first four rows of the CDF from the power sums
        const float thr = (float)prm.thr;
        const float p0 = __expf(lp0);
        const float e1 = t1, e2 = 0.5f * (t1 * t1 - t2), e3 = (t1 * t1 * t1 - 3.0f * t1 * t2 + 2.0f * t3) * (1.0f / 6.0f);
        float cdf = p0;
        if (cdf > thr) rows = 1;
        else if ((cdf += p0 * e1) > thr) rows = 2;
        else if ((cdf += p0 * e2) > thr) rows = 3;
        else if ((cdf += p0 * e3) > thr) rows = 4;
        rows = max(min(rows, li - ambi + 1), 1);
        atomicAdd(hist + ((lower > 0 || bad_len) ? 0 : min(rows, MPB_NAR_BUCKETS - 1)),      // an 'n' is what the narrow pass hands back
                  len ? max(1, (li + 15) >> 4) : 1);
        // ragged batches: the shortest sampled reads (in chunks) that need a third / a fourth row (k_narrow_rg's mixed rows)
        if (len && lower == 0 && !bad_len) {
            if (rows >= 3) atomicMin(hist + MPB_NAR_BUCKETS, (li + 15) >> 4);
            if (rows >= 4) atomicMin(hist + MPB_NAR_BUCKETS + 1, (li + 15) >> 4);
        }
    }
}

}  // namespace

// ------------------------------------------------------------------------------------------
// launch wrappers
// ------------------------------------------------------------------------------------------
static inline int pre_blocks(int64_t n) { return (int)((n + MPB_PRE_READS - 1) / MPB_PRE_READS); }

void mpb_launch_prepass(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                        const MpbDevParams &prm, const MpbWorkspace &ws, int32_t *ns_out,
                        double *ee_out, uint8_t *pass_out, hipStream_t s, const int32_t *list)
{
#define MPB_PRE_LAUNCH(RG, LG, LS)                                                                                      \
    hipLaunchKernelGGL((k_prepass<RG, LG, LS>), dim3(pre_blocks(n)), dim3(256), 0, s, q, n, stride, len, prm, o, ws.blockhist, list)
    const PreOut o = {ws.cls, ns_out, ee_out, pass_out, ws.bad_len, ws.wide_list, ws.wide_rows, ws.wide_count};
    const bool long_rows = prm.max_len > 16 * 12 * MPB_PRE_NB;        // more than one panel of 60 chunk columns
    if (list) {
        if (len) { if (long_rows) MPB_PRE_LAUNCH(true, true, true); else MPB_PRE_LAUNCH(true, false, true); }
        else     { if (long_rows) MPB_PRE_LAUNCH(false, true, true); else MPB_PRE_LAUNCH(false, false, true); }
    } else {
        if (len) { if (long_rows) MPB_PRE_LAUNCH(true, true, false); else MPB_PRE_LAUNCH(true, false, false); }
        else     { if (long_rows) MPB_PRE_LAUNCH(false, true, false); else MPB_PRE_LAUNCH(false, false, false); }
    }
#undef MPB_PRE_LAUNCH
}

// classified at source: decode raw FASTQ text into the packed matrix `out` and classify it in the same pass
void mpb_launch_decode_classify(const uint8_t *seq, const uint8_t *qual, int32_t offset, uint8_t *out, int32_t *err,
                                int64_t n, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                                const MpbWorkspace &ws, int32_t *ns_out, double *ee_out, uint8_t *pass_out, hipStream_t s)
{
    const PreOut o = {ws.cls, ns_out, ee_out, pass_out, ws.bad_len, ws.wide_list, ws.wide_rows, ws.wide_count};
    const PreDecode dec = {seq, qual, out, offset, err};
    const uint8_t *q = out;
    const uint32_t cpr = (uint32_t)(stride >> 4);
    const uint32_t magic = cpr == 1 ? 0u : (uint32_t)(((1ull << 32) + cpr - 1) / cpr);    // ceil(2^32 / cpr); 0 stands for cpr == 1
    if (len) hipLaunchKernelGGL((k_classify_linear<true, true>), dim3(pre_blocks(n)), dim3(256), 0, s, q, n, stride, len, prm, o, ws.blockhist, dec, magic);
    else     hipLaunchKernelGGL((k_classify_linear<false, true>), dim3(pre_blocks(n)), dim3(256), 0, s, q, n, stride, len, prm, o, ws.blockhist, dec, magic);
}

void mpb_launch_encode(const uint8_t *q, int64_t n, int64_t stride, int32_t offset, uint8_t *seq, uint8_t *qual, hipStream_t s)
{
    const int64_t chunks = n * (stride / 16);
    hipLaunchKernelGGL(k_encode_ascii, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, q, n, stride, offset, seq, qual);
}

void mpb_launch_scan(int64_t n, const int32_t *len, const MpbWorkspace &ws, hipStream_t s)
{
    const int nb = len ? MPB_LEN_BINS : 1;
    hipLaunchKernelGGL(k_scan, dim3(MPB_NCLS * nb), dim3(256), 0, s, ws.blockhist, pre_blocks(n), ws.tables);
    hipLaunchKernelGGL(k_tables, dim3(1), dim3(512), 0, s, ws.tables, nb, ws.ovf_count, ws.pass_count);
}

void mpb_launch_scatter(int64_t n, const int32_t *len, const int32_t *ns, const MpbDevParams &prm, const MpbWorkspace &ws,
                        hipStream_t s, const int32_t *list)
{
    if (len)
        hipLaunchKernelGGL((k_scatter<true>), dim3(pre_blocks(n)), dim3(256), 0, s, ws.cls, n, len, prm.max_len, prm.len_shift,
                           ws.blockhist, ws.tables, ns, ws.perm, ws.perm_ns, list);
    else
        hipLaunchKernelGGL((k_scatter<false>), dim3(pre_blocks(n)), dim3(256), 0, s, ws.cls, n, len, prm.max_len, prm.len_shift,
                           ws.blockhist, ws.tables, ns, ws.perm, ws.perm_ns, list);
}

// cap of the DP grid (blocks of 4 waves); beyond it the chunk loop strides
#define MPB_DP_GRID (1 << 20)


static DpArgs make_args(const uint8_t *q, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                        const MpbWorkspace &ws, const int32_t *ns, double *ee, uint8_t *pass, int final_pass)
{
    DpArgs A;
    A.q = q; A.stride = stride; A.len = len; A.ns = ns; A.cls = ws.cls; A.ee = ee; A.pass = pass;
    A.ovf_list = ws.ovf_list; A.ovf_count = ws.ovf_count; A.alg_cells = ws.alg_cells; A.prm = prm; A.final_pass = final_pass;
    A.perm = ws.perm; A.perm_ns = final_pass == 0 ? ws.perm_ns : nullptr;   // overflow / small-batch passes walk other lists
    return A;
}

void mpb_launch_dp(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                   const MpbDevParams &prm, const MpbWorkspace &ws, const int32_t *ns,
                   double *ee, uint8_t *pass, hipStream_t s)
{
    // Upper bound for the tile count without a host round trip: every read in a G==1 class
    // (64 reads per tile) plus one partial tile per class.  Classes with G > 1 have more tiles
    // per read; the kernel's tile loop is grid-strided, so they are still covered.
    const int64_t tiles = (n + 63) / 64 + MPB_NCLS;
    // a small batch (the sub-batch a narrow pass hands back, a short file's last chunk) is latency: fewer tiles per wave so that
    // every SIMD gets one -- 8 tiles per class-body call only pay where there are tens of thousands of tiles
    const int chunk_tiles = tiles >= 32768 ? MPB_DP_CHUNK : tiles >= 16384 ? 4 : tiles >= 8192 ? 2 : 1;
    constexpr int grid_cap = MPB_DP_GRID;
    int64_t want = (tiles + 4 * chunk_tiles - 1) / (4 * chunk_tiles);   // blocks if every wave took one chunk
    const int blocks = (int)(want < grid_cap ? (want > 0 ? want : 1) : grid_cap);
    DpArgs A = make_args(q, stride, len, prm, ws, ns, ee, pass, 0);
    if (prm.flags & 2u)
        hipLaunchKernelGGL((k_dp<true, false>), dim3(blocks), dim3(256), 0, s, A, ws.lut, ws.tables, ws.perm, chunk_tiles);
    else
        hipLaunchKernelGGL((k_dp<false, false>), dim3(blocks), dim3(256), 0, s, A, ws.lut, ws.tables, ws.perm, chunk_tiles);
}

// blocks of a wide-kernel instance (one read per block and trip; the list length is only known on the device)
#define MPB_WIDE_GRID 1024

// the instances whose wave count a batch with up to `max_rows` rows per read can need
template <bool FINAL>
static void launch_wide(const DpArgs &A, const double2 *lut, const int32_t *list, const int32_t *budget,
                        const int32_t *count, int max_rows, hipStream_t s)
{
    const int max_waves = (max_rows + MPB_TILE_MAX_ROWS - 1) / MPB_TILE_MAX_ROWS;
    hipLaunchKernelGGL((k_wide<FINAL, 2>), dim3(MPB_WIDE_GRID), dim3(64 * 2), 0, s, A, lut, list, budget, count);
    if (max_waves > 2) hipLaunchKernelGGL((k_wide<FINAL, 4>), dim3(MPB_WIDE_GRID), dim3(64 * 4), 0, s, A, lut, list, budget, count);
    if (max_waves > 4) hipLaunchKernelGGL((k_wide<FINAL, 8>), dim3(MPB_WIDE_GRID), dim3(64 * 8), 0, s, A, lut, list, budget, count);
    if (max_waves > 8) hipLaunchKernelGGL((k_wide<FINAL, 16>), dim3(MPB_WIDE_GRID), dim3(64 * 16), 0, s, A, lut, list, budget, count);
}

void mpb_launch_wide(const uint8_t *q, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                     const MpbWorkspace &ws, const int32_t *ns, double *ee, uint8_t *pass, hipStream_t s)
{
    DpArgs A = make_args(q, stride, len, prm, ws, ns, ee, pass, 0);
    A.perm_ns = nullptr;
    launch_wide<false>(A, ws.lut, ws.wide_list, ws.wide_rows, ws.wide_count, prm.max_len + 1, s);
}

void mpb_launch_overflow(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                         const MpbDevParams &prm, const MpbWorkspace &ws, const int32_t *ns,
                         double *ee, uint8_t *pass, hipStream_t s)
{
    (void)n;
    if (prm.max_len + 1 > MPB_TILE_MAX_ROWS) {
        // no tile class covers len + 1 rows of this batch's longest reads: the overflow list goes to the wide kernel,
        // every read with len + 1 rows (a short read of such a batch then simply keeps one wave busy)
        hipLaunchKernelGGL(k_tables_overflow, dim3(1), dim3(64), 0, s, ws.tables2, ws.ovf_count, MPB_NCLS - 1, ws.ovf_total);
        DpArgs A = make_args(q, stride, len, prm, ws, ns, ee, pass, 1);
        launch_wide<true>(A, ws.lut, ws.ovf_list, nullptr, ws.ovf_count, prm.max_len + 1, s);
        return;
    }
    static const MpbClass classes[MPB_NCLS] = MPB_CLASS_TABLE;
    int wc = MPB_NCLS - 1;
    for (int c = MPB_NCLS - 1; c >= 0; c--)
        if (classes[c].cap >= prm.max_len + 1) wc = c;
    hipLaunchKernelGGL(k_tables_overflow, dim3(1), dim3(64), 0, s, ws.tables2, ws.ovf_count, wc, ws.ovf_total);
    DpArgs A = make_args(q, stride, len, prm, ws, ns, ee, pass, 1);
    const int blocks = 256;
    // always the three-rounding arithmetic: with MPB_FLAG_FAST_FMA this pass also settles the reads whose fma
    // result was too close to the threshold to decide
    hipLaunchKernelGGL((k_dp<false, true>), dim3(blocks), dim3(256), 0, s, A, ws.lut, ws.tables2, ws.ovf_list, MPB_DP_CHUNK);
}

void mpb_launch_lambda(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len, int32_t fixed_len,
                       const double2 *lut_ap, double *lambda, int32_t *ns, int32_t *bad, hipStream_t s)
{
    int64_t blocks = (n + 255) / 256;
    if (blocks > (1 << 20)) blocks = 1 << 20;            // the tile loop is grid-strided
    if (len)
        hipLaunchKernelGGL((k_lambda<true>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, len,
                           fixed_len, lut_ap, lambda, ns, bad);
    else
        hipLaunchKernelGGL((k_lambda<false>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, len,
                           fixed_len, lut_ap, lambda, ns, bad);
}

void mpb_launch_decode(const uint8_t *seq, const uint8_t *qual, int64_t n, int64_t stride, const int32_t *len,
                       int32_t fixed_len, int32_t offset, uint8_t *out, int32_t *err, hipStream_t s)
{
    const int64_t chunks = n * (stride / 16);
    hipLaunchKernelGGL(k_decode_ascii, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, s, seq, qual, n,
                       stride, len, fixed_len, offset, out, err);
}

void mpb_launch_count(const uint8_t *pass, int64_t n, const MpbWorkspace &ws, hipStream_t s)
{
    int blocks = (int)((n + 255) / 256);
    if (blocks > 1024) blocks = 1024;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(k_count, dim3(blocks), dim3(256), 0, s, pass, n, ws.pass_count);
}

void mpb_launch_synth(uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, int32_t min_len,
                      int32_t max_len, int32_t *len, uint64_t seed, int64_t first_read,
                      hipStream_t s, int32_t profile)
{
    const int64_t chunks = n * (stride / 16);
    const int64_t blocks = (chunks + 255) / 256;
    hipLaunchKernelGGL(k_synth, dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, fixed_len,
                       min_len, max_len, len, seed, first_read, profile);
}

// Small batches: one launch, one read per wave (k_small).  Overflow is reported through pass == 2.
void mpb_launch_small(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                      const MpbWorkspace &ws, int32_t *ns, double *ee, uint8_t *pass, hipStream_t s, const MpbSmallHost *host)
{
    // host == nullptr: everything in device memory.  Else q / ns / ee / pass are pinned host memory and `host` names the
    // device scratch the kernel parks the rows and the counts in, and the completion flags (see k_small).
    DpArgs A = make_args(q, stride, len, prm, ws, host && host->ns_dev ? host->ns_dev : ns, ee, pass, 2);
    const int blocks = (int)((n + 3) / 4);
    uint8_t *stage = host ? host->stage : nullptr;
    uint32_t *done = host ? host->done : nullptr;
    const uint32_t token = host ? host->token : 0u;
    if (prm.flags & 2u)
        hipLaunchKernelGGL((k_small<true>), dim3(blocks), dim3(256), 0, s, A, ws.lut, n, ns, ws.cls, ws.perm, stage, done, token);
    else
        hipLaunchKernelGGL((k_small<false>), dim3(blocks), dim3(256), 0, s, A, ws.lut, n, ns, ws.cls, ws.perm, stage, done, token);
}

// the resident one-read server: one wave per mailbox entry; `gone` (device) is zeroed on the same stream first
void mpb_launch_serve(const MpbServeBox &box, const double2 *lut, uint32_t generation, uint32_t lifetime_ms, hipStream_t s)
{
    (void)hipMemsetAsync(box.gone, 0, sizeof(uint32_t), s);
    hipLaunchKernelGGL(k_serve, dim3(box.n_ent), dim3(64), 0, s, box, lut, generation, (unsigned long long)lifetime_ms * 100000ull);
}

// ---- natural-order narrow pass ----------------------------------------------------------------------------------------
int mpb_narrow_lds_bytes()
{
    return 256 * (int)sizeof(nar_entry_t) + MPB_NAR_DEPTH * 4 * MPB_NAR_PANEL + 32;
}

// k_narrow_rs (register-staged, whole lines): reads per lane, 0: not this form.  Rows of a multiple of 64 bytes: always (a read
// starts with a half panel: one run per half).  Other strides: a lane's reads begin anywhere a 16-byte chunk does and a half may
// be several runs -- measured at stride 304 (profiles/r05_narrow_variants.txt): 7 % faster than the ring at R = 2 (and 3.04
// instead of 4.2 GB read), 4-5 % slower at R = 3, 4: so only for two rows.
int mpb_narrow_rs_reads_per_lane(int64_t stride, int rows0)
{
    if (stride % 16 != 0 || stride > (1 << 16)) return 0;
    if (stride % 64 != 0 && rows0 != 2) return 0;
    int k = 1;                                                        // 128 / gcd(stride, 128): 1, 2, 4 or 8
    while ((k * stride) % 128 != 0) k *= 2;
    return k;
}

int mpb_narrow_rs_lds_bytes()
{
    return 256 * (int)sizeof(nar_entry_t) + 4 * MPB_NRS_TILE;
}

void mpb_launch_narrow(int rows0, const uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, const MpbDevParams &prm,
                       const MpbWorkspace &ws, double *ee, int32_t *ns, uint8_t *pass, int32_t *list, int grid_blocks, hipStream_t s)
{
    const int rs_k = mpb_narrow_rs_reads_per_lane(stride, rows0);
    const int per_blk = 64 * (rs_k ? rs_k : 1);              // reads of one row block / stream block
    const int64_t nblk = (n + per_blk - 1) / per_blk;
    int64_t blocks = (nblk + 3) / 4;
    if (blocks > grid_blocks) blocks = grid_blocks;         // persistent: a wave walks row blocks gw, gw + W, ...
    if (blocks > MPB_NAR_MAX_WAVES / 4) blocks = MPB_NAR_MAX_WAVES / 4;
    if (blocks < 1) blocks = 1;
    const int nwaves = (int)blocks * 4;
#define MPB_NAR_LAUNCH(RR) hipLaunchKernelGGL((k_narrow<RR, MPB_NAR_DEPTH>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, fixed_len, prm, ws.lut, ee, ns, pass, ws.nar_seg, ws.nar_wave_count)
#define MPB_NRS_LAUNCH(RR) hipLaunchKernelGGL((k_narrow_rs<RR, true>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, fixed_len, rs_k, prm, ws.lut, ee, ns, pass, ws.nar_seg, ws.nar_wave_count)
    if (rs_k && stride % 64 != 0) {                        // (two rows only: mpb_narrow_rs_reads_per_lane)
        hipLaunchKernelGGL((k_narrow_rs<2, false>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, fixed_len, rs_k, prm, ws.lut, ee, ns, pass, ws.nar_seg, ws.nar_wave_count);
    } else if (rs_k) {
        switch (rows0) {
        case 2: MPB_NRS_LAUNCH(2); break;
        case 3: MPB_NRS_LAUNCH(3); break;
        default: MPB_NRS_LAUNCH(4); break;
        }
    } else {
        switch (rows0) {
        case 2: MPB_NAR_LAUNCH(2); break;
        case 3: MPB_NAR_LAUNCH(3); break;
        default: MPB_NAR_LAUNCH(4); break;
        }
    }
#undef MPB_NRS_LAUNCH
#undef MPB_NAR_LAUNCH
    hipLaunchKernelGGL(k_nar_compact, dim3((unsigned)nwaves), dim3(256), 0, s, ws.nar_seg, ws.nar_wave_count, nblk, nwaves, per_blk, (const int32_t *)nullptr, list, ws.nar_count);
}

// ragged batches (k_narrow_rg): sort windows by length, cut the groups into one range per wave, walk them
int mpb_narrow_rg_key_shift(int64_t stride)
{
    int ks = 0;                                              // ceil(len / 16) <= stride / 16; the key must stay below MPB_RG_BINS
    while (((stride >> 4) >> ks) >= MPB_RG_BINS) ks++;
    return ks;
}

void mpb_launch_narrow_ragged(int rows0, int split_chunks, const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                              const MpbDevParams &prm, const MpbWorkspace &ws, double *ee, int32_t *ns, uint8_t *pass, int32_t *list,
                              int grid_blocks, hipStream_t s)
{
    const int64_t ngroups = (n + 63) / 64, nwin = (n + MPB_RG_WIN - 1) / MPB_RG_WIN;
    int64_t blocks = (ngroups + 3) / 4;
    // persistent, and every wave's range is fixed before the launch: the grid must be resident at once, so it is sized by what
    // the runtime says fits a CU (registers as well as LDS), never by more than the caller's LDS-only figure
    static int per_cu[5] = {0, 0, 0, 0, 0};
    const int ri = rows0 < 2 ? 2 : rows0 > 4 ? 4 : rows0;
    if (!per_cu[ri]) {
        int nb = 0;
        const void *fn = ri == 2 ? (const void *)k_narrow_rg<2, 2> : ri == 3 ? (const void *)k_narrow_rg<3, 2> : (const void *)k_narrow_rg<4, 3>;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, fn, 256, 0) != hipSuccess || nb < 1) { (void)hipGetLastError(); nb = 1; }
        per_cu[ri] = nb;
    }
    {
        int dev = 0, n_cu = 256;
        if (hipGetDevice(&dev) == hipSuccess) (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
        if ((int64_t)n_cu * per_cu[ri] < grid_blocks) grid_blocks = n_cu * per_cu[ri];
    }
    if (blocks > grid_blocks) blocks = grid_blocks;          // a wave walks a contiguous range of groups
    if (blocks > MPB_NAR_MAX_WAVES / 4) blocks = MPB_NAR_MAX_WAVES / 4;
    if (blocks < 1) blocks = 1;
    const int nwaves = (int)blocks * 4;
    // (what a short group costs against a full one, from the pure kernels' times at R - 1 and R: 0.86 at 2 / 3, 0.80 at 3 / 4)
    hipLaunchKernelGGL(k_rag_sort, dim3((unsigned)nwin), dim3(256), 0, s, len, n, prm.max_len, mpb_narrow_rg_key_shift(stride),
                       split_chunks > 0 && rows0 >= 3 ? split_chunks : -1, rows0 >= 4 ? 80 : 86, ws.rg_ord, ws.rg_gpre, ws.rg_wsum);
    hipLaunchKernelGGL(k_rag_scan, dim3(1), dim3(1024), 0, s, ws.rg_wsum, (int)nwin, ws.rg_wpre);
#define MPB_NRG_LAUNCH(RR, RL) hipLaunchKernelGGL((k_narrow_rg<RR, RL>), dim3((unsigned)blocks), dim3(256), 0, s, q, n, stride, ws.rg_ord, ws.rg_wpre, ws.rg_gpre, ws.rg_gstart, split_chunks, prm, ws.lut, ee, ns, pass, ws.nar_seg, ws.nar_wave_count)
    const bool mixed = split_chunks > 0 && rows0 >= 3;       // short groups with a row less (k_narrow_rg<R, R - 1>)
    switch (rows0) {
    case 2: MPB_NRG_LAUNCH(2, 2); break;
    case 3: if (mixed) MPB_NRG_LAUNCH(3, 2); else MPB_NRG_LAUNCH(3, 3); break;
    default: if (mixed) MPB_NRG_LAUNCH(4, 3); else MPB_NRG_LAUNCH(4, 4); break;
    }
#undef MPB_NRG_LAUNCH
    hipLaunchKernelGGL(k_nar_compact, dim3((unsigned)nwaves), dim3(256), 0, s, ws.nar_seg, ws.nar_wave_count, (int64_t)0, nwaves, 64, ws.rg_gstart, list, ws.nar_count);
}

void mpb_launch_sample(const uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, const int32_t *len, const MpbDevParams &prm,
                       const MpbWorkspace &ws, int n_sample, hipStream_t s)
{
    (void)hipMemsetAsync(ws.nar_sample, 0, MPB_NAR_BUCKETS * sizeof(int32_t), s);
    (void)hipMemsetAsync(ws.nar_sample + MPB_NAR_BUCKETS, 0x7f, 2 * sizeof(int32_t), s);       // the two minima: 0x7f7f7f7f = none
    hipLaunchKernelGGL(k_sample, dim3((unsigned)((n_sample + 3) / 4)), dim3(256), 0, s, q, n, stride, fixed_len, len, prm, n_sample, ws.nar_sample);
}

