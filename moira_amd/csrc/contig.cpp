// contig.cpp -- CPU contig construction (libmoira_contig.so, C ABI in include/moira_contig.h).
//
// The build's own implementation of the reference's paired-read front end, which north_star
// keeps on the CPU: reverse complement (moira/moira.py:1207-1235), mothur-style Needleman-Wunsch
// (moira/nw_align.pyx:49-201) and consensus building (moira/moira.py:1376-1558).  Results must be
// identical to the reference's (pinned by its KATs and golden paired outputs), so tie-breaking
// order, the 3' overlap fix-up and the odd "score = sum of path cells" definition are kept.

#include "../../include/moira_contig.h"
#include "../../include/moira_io.h"      // MIO_* record-index columns

#include <algorithm>
#include <memory>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include <immintrin.h>

static thread_local char g_err[256] = "";
static int fail(int code, const char *fmt, ...)
{
    va_list ap; va_start(ap, fmt); vsnprintf(g_err, sizeof(g_err), fmt, ap); va_end(ap);
    return code;
}

extern "C" const char *mct_last_error(void) { return g_err; }

// ref: moira/moira.py:1210-1213
static char complement_of(char b)
{
    switch (b) {
    case 'A': return 'T'; case 'C': return 'G'; case 'T': return 'A'; case 'G': return 'C';
    case 'N': return 'N'; case 'W': return 'W'; case 'S': return 'S'; case 'R': return 'Y';
    case 'Y': return 'R'; case 'M': return 'K'; case 'K': return 'M'; case 'B': return 'V';
    case 'V': return 'B'; case 'D': return 'H'; case 'H': return 'D'; case '-': return '-';
    case '.': return '.';
    default: return 0;
    }
}

extern "C" int mct_reverse_complement(const char *seq, const int32_t *quals, int32_t len,
                                      char *out_seq, int32_t *out_quals)
{
    if (len < 0 || (len > 0 && (!seq || !out_seq))) return fail(MCT_E_INVALID, "bad arguments");
    for (int i = 0; i < len; i++) {
        const char c = complement_of(seq[len - 1 - i]);
        if (!c) return fail(MCT_E_BASE, "\"%c\" is not a recognizable IUPAC-coded base.", seq[len - 1 - i]);
        out_seq[i] = c;
    }
    if (quals && out_quals)
        for (int i = 0; i < len; i++) out_quals[i] = quals[len - 1 - i];
    return MCT_OK;
}

namespace {

enum : uint8_t { PTR_DIAG = 0, PTR_UP = 1, PTR_LEFT = 2 };   // (-1,-1), (-1,0), (0,-1)

struct NwScratch {
    std::vector<int32_t> score;
    std::vector<uint8_t> ptr;
};

// ref: moira/nw_align.pyx:49-201.  Row index i walks seq1 (with the leading pad), column j seq2.
// Scalar, row-major, 32-bit form: the plain restatement, kept for validation and for parameters whose
// scores could leave 16 bits.
int nw_align_scalar(const char *s1, int n1, const char *s2, int n2, int match, int mismatch, int gap,
                    char *aln1, char *aln2, int32_t *aln_len, int32_t *score_out, NwScratch &sc)
{
    const int L1 = n1 + 1, L2 = n2 + 1;
    sc.score.assign((size_t)L1 * L2, 0);
    sc.ptr.assign((size_t)L1 * L2, PTR_UP);
    int32_t *S = sc.score.data();
    uint8_t *P = sc.ptr.data();
    // first column points up, first row points left, both scored 0 (refine_overlap, :64-76)
    for (int j = 1; j < L2; j++) P[j] = PTR_LEFT;
    for (int i = 1; i < L1; i++) {
        const char a = s1[i - 1];
        int32_t *row = S + (size_t)i * L2;
        const int32_t *prev = row - L2;
        uint8_t *prow = P + (size_t)i * L2;
        for (int j = 1; j < L2; j++) {
            const int diag = prev[j - 1] + (a == s2[j - 1] ? match : mismatch);
            const int up = prev[j] + gap;
            const int left = row[j - 1] + gap;
            // tie-break order of :96-113: diag >= up, diag >= left, up >= left
            if (diag >= up) {
                if (diag >= left) { row[j] = diag; prow[j] = PTR_DIAG; }
                else { row[j] = left; prow[j] = PTR_LEFT; }
            } else {
                if (up >= left) { row[j] = up; prow[j] = PTR_UP; }
                else { row[j] = left; prow[j] = PTR_LEFT; }
            }
        }
    }
    // 3' overlap fix-up (:155-201): last maximum (>=) of the last column and of the last row
    int best_col_score = -10000, best_col_idx = 0;
    for (int i = 0; i < L1; i++) {
        const int c = S[(size_t)i * L2 + (L2 - 1)];
        if (c >= best_col_score) { best_col_score = c; best_col_idx = i; }
    }
    int best_row_score = -10000, best_row_idx = 0;
    for (int j = 0; j < L2; j++) {
        const int c = S[(size_t)(L1 - 1) * L2 + j];
        if (c >= best_row_score) { best_row_score = c; best_row_idx = j; }
    }
    if (best_col_idx == L1 - 1 && best_row_idx == L2 - 1) {
        // nothing to do
    } else if (best_col_score > best_row_score) {
        for (int i = L1 - 1; i > best_col_idx; i--) P[(size_t)i * L2 + (L2 - 1)] = PTR_UP;
    } else {
        for (int j = L2 - 1; j > best_row_idx; j--) P[(size_t)(L1 - 1) * L2 + j] = PTR_LEFT;
    }
    // traceback (:129-150), filled backwards then reversed
    int i = L1 - 1, j = L2 - 1, k = 0;
    int32_t score = 0;
    while (i > 0 || j > 0) {
        const uint8_t p = P[(size_t)i * L2 + j];
        score += S[(size_t)i * L2 + j];
        const bool mv_i = (p == PTR_DIAG || p == PTR_UP), mv_j = (p == PTR_DIAG || p == PTR_LEFT);
        aln1[k] = mv_i ? s1[i - 1] : '-';
        aln2[k] = mv_j ? s2[j - 1] : '-';
        k++;
        if (mv_i) i--;
        if (mv_j) j--;
    }
    for (int a = 0, b = k - 1; a < b; a++, b--) {
        char t = aln1[a]; aln1[a] = aln1[b]; aln1[b] = t;
        t = aln2[a]; aln2[a] = aln2[b]; aln2[b] = t;
    }
    aln1[k] = 0; aln2[k] = 0;
    *aln_len = k;
    *score_out = score;
    return MCT_OK;
}

// One anti-diagonal d = i + j of the matrix: its cells depend only on diagonals d-1 and d-2, so the
// loop below has no carried dependence and compiles to 16-lane integer SIMD where the CPU has AVX2
// (the clone is picked at load time).  All arrays are indexed by the row i of the cell.
__attribute__((target_clones("avx2", "default")))
void nw_diagonal(const int16_t *__restrict d2, const int16_t *__restrict d1, int16_t *__restrict cur,
                 uint8_t *__restrict ptr, const char *__restrict a, const char *__restrict b,
                 int ilo, int ihi, int16_t match, int16_t mismatch, int16_t gap)
{
    for (int i = ilo; i <= ihi; i++) {
        const int16_t dg = (int16_t)(d2[i - 1] + (a[i] == b[i] ? match : mismatch));
        const int16_t up = (int16_t)(d1[i - 1] + gap);
        const int16_t lf = (int16_t)(d1[i] + gap);
        const int16_t mx = dg >= up ? dg : up;
        cur[i] = lf > mx ? lf : mx;
        // tie-break order of nw_align.pyx:96-113 (diag >= up, diag >= left, up >= left), branch-free:
        // left only when it beats both, else diag when diag >= up, else up
        ptr[i] = lf > mx ? (uint8_t)PTR_LEFT : (dg >= up ? (uint8_t)PTR_DIAG : (uint8_t)PTR_UP);
    }
}

// On CPUs with AVX-512BW/VL (the GPU boxes' EPYC 9575F has full-width AVX-512) the same diagonal runs 32 cells per
// instruction (nw_fill_avx512 below; picked once per process): the same integer operations in the same order -- wrapping
// 16-bit adds, max, and the tie-break encoded as  left > max(diag, up) ? LEFT : (diag >= up ? DIAG : UP).
// The whole fill, diagonal by diagonal, in one function per instruction set: a 2 x 250-base pair has 500 diagonals of
// 125 cells on average, so a call through a pointer per diagonal (re-broadcasting the three scores each time) cost as
// much as the cells themselves.  base[d] as in nw_align_diag; s2r = seq_2 reversed.
// Scores are kept for three diagonals only (R[0..2], indexed by the row i of a cell, rotated): a cell needs its two
// predecessors' diagonals and nothing else, and the traceback rebuilds the scores along its path from the pointers (see
// nw_align_diag).  What has to survive is the last column and the last row (the 3' fix-up scans them and the path may run
// along them): lastcol[i] = S(i, n2), lastrow[j] = S(n1, j), captured as their diagonals are finished.  Storing 2 bytes of
// score per cell on top of the pointer byte was three quarters of the fill's memory traffic (270 KB per 2 x 300 pair,
// L2-resident): the fill was bound by it.
typedef void (*nw_fill_fn)(int16_t *R, int rstride, uint8_t *P, const int64_t *base, const char *A, const char *s2r, int n1, int n2,
                           int16_t match, int16_t mismatch, int16_t gap, int16_t *lastcol, int16_t *lastrow);

// rotate the three diagonal buffers for diagonal d, seed its border cells, and after the diagonal record what belongs
// to the last column / row
#define NW_DIAG_BEGIN                                                                                              \
        int16_t *const cur = R + (size_t)(d % 3) * rstride;                                                       \
        const int16_t *const d1 = R + (size_t)((d + 2) % 3) * rstride, *const d2 = R + (size_t)((d + 1) % 3) * rstride; \
        if (d <= n2) cur[0] = 0;                                 /* cell (0, d) */                                 \
        if (d <= n1) cur[d] = 0;                                 /* cell (d, 0) */
#define NW_DIAG_END                                                                                                \
        if (d > n2 && d - n2 <= n1) lastcol[d - n2] = cur[d - n2];          /* cell (d - n2, n2), row >= 1 */       \
        if (d > n1 && d - n1 <= n2) lastrow[d - n1] = cur[n1];              /* cell (n1, d - n1), column >= 1 */

void nw_fill_generic(int16_t *R, int rstride, uint8_t *P, const int64_t *base, const char *A, const char *s2r, int n1, int n2,
                     int16_t match, int16_t mismatch, int16_t gap, int16_t *lastcol, int16_t *lastrow)
{
    const int D = n1 + n2;
    for (int d = 2; d <= D; d++) {
        NW_DIAG_BEGIN
        const int ilo = d - n2 > 1 ? d - n2 : 1, ihi = d - 1 < n1 ? d - 1 : n1;
        // b[i] = s2[(d - i) - 1] = s2r[n2 - d + i]
        if (ilo <= ihi) nw_diagonal(d2, d1, cur, P + base[d], A, s2r + (n2 - d), ilo, ihi, match, mismatch, gap);
        NW_DIAG_END
    }
}

__attribute__((target("avx512f,avx512bw,avx512vl")))
void nw_fill_avx512(int16_t *R, int rstride, uint8_t *P, const int64_t *base, const char *A, const char *s2r, int n1, int n2,
                    int16_t match, int16_t mismatch, int16_t gap, int16_t *lastcol, int16_t *lastrow)
{
    const __m512i vmatch = _mm512_set1_epi16(match), vmis = _mm512_set1_epi16(mismatch), vgap = _mm512_set1_epi16(gap);
    const __m512i one = _mm512_set1_epi16(PTR_UP), two = _mm512_set1_epi16(PTR_LEFT), zero = _mm512_setzero_si512();
    const int D = n1 + n2;
    for (int d = 2; d <= D; d++) {
        NW_DIAG_BEGIN
        const int ilo = d - n2 > 1 ? d - n2 : 1, ihi = d - 1 < n1 ? d - 1 : n1;
        uint8_t *ptr = P + base[d];
        const char *b = s2r + (n2 - d);
        // whole vectors without masks (a masked 512-bit load / store is the slower instruction), the rest of the diagonal as
        // one more whole vector over its LAST 32 cells (cells computed twice get the same values: they depend on the two
        // finished diagonals only); only a diagonal shorter than a vector takes the masked form
#define NW_AVX512_CELLS(i, LOAD8, LOAD16, STORE16, STORE8)                                                              \
        {                                                                                                               \
            const __m256i av = LOAD8(A + (i)), bv = LOAD8(b + (i));                                                     \
            const __mmask32 eq = _mm256_cmpeq_epi8_mask(av, bv);                                                        \
            const __m512i dg = _mm512_add_epi16(LOAD16(d2 + (i) - 1), _mm512_mask_blend_epi16(eq, vmis, vmatch));       \
            const __m512i up = _mm512_add_epi16(LOAD16(d1 + (i) - 1), vgap);                                            \
            const __m512i lf = _mm512_add_epi16(LOAD16(d1 + (i)), vgap);                                                \
            const __m512i mx = _mm512_max_epi16(dg, up);                                                                \
            const __mmask32 left = _mm512_cmpgt_epi16_mask(lf, mx);                                                     \
            const __mmask32 diag = _mm512_cmpge_epi16_mask(dg, up);                                                     \
            STORE16(cur + (i), _mm512_max_epi16(lf, mx));                                                               \
            __m512i p = _mm512_mask_blend_epi16(diag, one, zero);                                                       \
            p = _mm512_mask_blend_epi16(left, p, two);                                                                  \
            STORE8(ptr + (i), _mm512_cvtepi16_epi8(p));                                                                 \
        }
#define NW_L8(ptr_) _mm256_loadu_si256((const __m256i *)(ptr_))
#define NW_L16(ptr_) _mm512_loadu_si512((const void *)(ptr_))
#define NW_S16(ptr_, v) _mm512_storeu_si512((void *)(ptr_), v)
#define NW_S8(ptr_, v) _mm256_storeu_si256((__m256i *)(ptr_), v)
#define NW_ML8(ptr_) _mm256_maskz_loadu_epi8(k, ptr_)
#define NW_ML16(ptr_) _mm512_maskz_loadu_epi16(k, ptr_)
#define NW_MS16(ptr_, v) _mm512_mask_storeu_epi16(ptr_, k, v)
#define NW_MS8(ptr_, v) _mm256_mask_storeu_epi8(ptr_, k, v)
        if (ihi - ilo >= 31) {
            int i = ilo;
            for (; i + 31 <= ihi; i += 32) NW_AVX512_CELLS(i, NW_L8, NW_L16, NW_S16, NW_S8)
            if (i <= ihi) { const int i2 = ihi - 31; NW_AVX512_CELLS(i2, NW_L8, NW_L16, NW_S16, NW_S8) }
        } else if (ilo <= ihi) {
            const __mmask32 k = (__mmask32)((1ull << (ihi - ilo + 1)) - 1ull);
            NW_AVX512_CELLS(ilo, NW_ML8, NW_ML16, NW_MS16, NW_MS8)
        }
#undef NW_AVX512_CELLS
#undef NW_L8
#undef NW_L16
#undef NW_S16
#undef NW_S8
#undef NW_ML8
#undef NW_ML16
#undef NW_MS16
#undef NW_MS8
        NW_DIAG_END
    }
}

// AVX2: 16 cells per instruction, the remainder of a diagonal cell by cell
__attribute__((target("avx2")))
void nw_fill_avx2(int16_t *R, int rstride, uint8_t *P, const int64_t *base, const char *A, const char *s2r, int n1, int n2,
                  int16_t match, int16_t mismatch, int16_t gap, int16_t *lastcol, int16_t *lastrow)
{
    const __m256i vmatch = _mm256_set1_epi16(match), vmis = _mm256_set1_epi16(mismatch), vgap = _mm256_set1_epi16(gap);
    const __m256i one = _mm256_set1_epi16(PTR_UP), two = _mm256_set1_epi16(PTR_LEFT);
    const int D = n1 + n2;
    for (int d = 2; d <= D; d++) {
        NW_DIAG_BEGIN
        const int ilo = d - n2 > 1 ? d - n2 : 1, ihi = d - 1 < n1 ? d - 1 : n1;
        uint8_t *ptr = P + base[d];
        const char *b = s2r + (n2 - d);
        // 16 cells: 0xffff where the bases are equal; UP where up > diag (else DIAG = 0), LEFT where left beats both; the
        // 16 pointer words narrowed to bytes through the low half of a pack
#define NW_AVX2_CELLS(i)                                                                                                \
        {                                                                                                               \
            const __m128i av = _mm_loadu_si128((const __m128i *)(A + i)), bv = _mm_loadu_si128((const __m128i *)(b + i));\
            const __m256i eq = _mm256_cvtepi8_epi16(_mm_cmpeq_epi8(av, bv));                                            \
            const __m256i sub = _mm256_blendv_epi8(vmis, vmatch, eq);                                                   \
            const __m256i dg = _mm256_add_epi16(_mm256_loadu_si256((const __m256i *)(d2 + i - 1)), sub);                \
            const __m256i up = _mm256_add_epi16(_mm256_loadu_si256((const __m256i *)(d1 + i - 1)), vgap);               \
            const __m256i lf = _mm256_add_epi16(_mm256_loadu_si256((const __m256i *)(d1 + i)), vgap);                   \
            const __m256i mx = _mm256_max_epi16(dg, up);                                                                \
            const __m256i left = _mm256_cmpgt_epi16(lf, mx);                                                            \
            const __m256i upwins = _mm256_cmpgt_epi16(up, dg);                                                          \
            _mm256_storeu_si256((__m256i *)(cur + i), _mm256_max_epi16(lf, mx));                                        \
            __m256i p = _mm256_and_si256(upwins, one);                                                                  \
            p = _mm256_blendv_epi8(p, two, left);                                                                       \
            const __m256i packed = _mm256_permute4x64_epi64(_mm256_packus_epi16(p, p), 0xd8);                           \
            _mm_storeu_si128((__m128i *)(ptr + i), _mm256_castsi256_si128(packed));                                     \
        }
        int i = ilo;
        for (; i + 15 <= ihi; i += 16) NW_AVX2_CELLS(i)
        // the rest of the diagonal: one more vector over its LAST 16 cells (cells computed twice get the same values: they
        // depend on the two finished diagonals only); a diagonal shorter than a vector cell by cell
        if (i <= ihi && ihi - ilo >= 15) { const int i2 = ihi - 15; NW_AVX2_CELLS(i2) i = ihi + 1; }
#undef NW_AVX2_CELLS
        for (; i <= ihi; i++) {
            const int16_t dg = (int16_t)(d2[i - 1] + (A[i] == b[i] ? match : mismatch));
            const int16_t up = (int16_t)(d1[i - 1] + gap);
            const int16_t lf = (int16_t)(d1[i] + gap);
            const int16_t mx = dg >= up ? dg : up;
            cur[i] = lf > mx ? lf : mx;
            ptr[i] = lf > mx ? (uint8_t)PTR_LEFT : (dg >= up ? (uint8_t)PTR_DIAG : (uint8_t)PTR_UP);
        }
        NW_DIAG_END
    }
}

nw_fill_fn pick_fill()
{
    const char *force = getenv("MOIRA_CONTIG_ISA");          // tests: "generic", "avx2", "avx512" (default: the best the CPU has)
    __builtin_cpu_init();
    const bool has512 = __builtin_cpu_supports("avx512f") && __builtin_cpu_supports("avx512bw") && __builtin_cpu_supports("avx512vl");
    const bool has2 = __builtin_cpu_supports("avx2");
    if (force) {
        if (!strcmp(force, "generic")) return nw_fill_generic;
        if (!strcmp(force, "avx2") && has2) return nw_fill_avx2;
        if (!strcmp(force, "avx512") && has512) return nw_fill_avx512;
    }
    if (getenv("MOIRA_CONTIG_NO_AVX512")) return has2 ? nw_fill_avx2 : nw_fill_generic;
    if (has512) return nw_fill_avx512;
    return has2 ? nw_fill_avx2 : nw_fill_generic;
}

struct DiagScratch {
    std::vector<int16_t> score;     // three diagonals + the last column + the last row
    std::vector<uint8_t> ptr;
    std::vector<int64_t> base;      // base[d]: index of cell (i, d - i) in ptr is base[d] + i
    std::vector<char> s2r;
};

// Same algorithm on anti-diagonal storage with 16-bit scores.
int nw_align_diag(const char *s1, int n1, const char *s2, int n2, int match, int mismatch, int gap,
                  char *aln1, char *aln2, int32_t *aln_len, int32_t *score_out, DiagScratch &sc)
{
    const int D = n1 + n2;
    sc.base.resize((size_t)D + 1);
    int64_t total = 8;                                   // slack so that base[d] + i - 1 never underflows
    for (int d = 0; d <= D; d++) {
        const int lo = d > n2 ? d - n2 : 0, hi = d < n1 ? d : n1;
        sc.base[d] = total - lo;
        total += hi - lo + 1;
    }
    // Only the first row and column of the pointer matrix are initialised (the column points up, the row left, :64-76):
    // every other cell is written by the fill before anything reads it, and clearing the matrix for every pair cost a
    // third of the alignment.
    if (sc.ptr.size() < (size_t)total + 40) sc.ptr.resize((size_t)total + 40);
    const int rstride = n1 + 72;                         // one diagonal, indexed by row 0..n1 (+ a vector of slack)
    const size_t need = (size_t)3 * rstride + (size_t)n1 + (size_t)n2 + 80;
    if (sc.score.size() < need) sc.score.resize(need);
    int16_t *R = sc.score.data();
    int16_t *lastcol = R + (size_t)3 * rstride, *lastrow = lastcol + n1 + 8;
    uint8_t *P = sc.ptr.data();
    for (int i = 0; i <= n1; i++) { P[sc.base[i] + i] = PTR_UP; lastcol[i] = 0; }                  // cells (i, 0); S(i, n2) until computed
    for (int j = 1; j <= n2; j++) P[sc.base[j] + 0] = PTR_LEFT;                                   // cells (0, j)
    for (int j = 0; j <= n2; j++) lastrow[j] = 0;
    R[0] = 0;                                            // diagonal 0: cell (0, 0)
    R[rstride] = 0; R[rstride + 1] = 0;                  // diagonal 1: cells (0, 1) and (1, 0)
    sc.s2r.resize((size_t)n2 + 1);
    for (int t = 0; t < n2; t++) sc.s2r[t] = s2[n2 - 1 - t];
    const char *A = s1 - 1;                              // A[i] = s1[i - 1]
    static const nw_fill_fn fill = pick_fill();
    fill(R, rstride, P, sc.base.data(), A, sc.s2r.data(), n1, n2, (int16_t)match, (int16_t)mismatch, (int16_t)gap, lastcol, lastrow);
    auto at = [&](int i, int j) { return sc.base[i + j] + i; };
    // 3' overlap fix-up (:155-201): last maximum (>=) of the last column and of the last row
    int best_col_score = -10000, best_col_idx = 0;
    for (int i = 0; i <= n1; i++) {
        const int c = lastcol[i];
        if (c >= best_col_score) { best_col_score = c; best_col_idx = i; }
    }
    int best_row_score = -10000, best_row_idx = 0;
    for (int j = 0; j <= n2; j++) {
        const int c = lastrow[j];
        if (c >= best_row_score) { best_row_score = c; best_row_idx = j; }
    }
    if (best_col_idx == n1 && best_row_idx == n2) {
        // nothing to do
    } else if (best_col_score > best_row_score) {
        for (int i = n1; i > best_col_idx; i--) P[at(i, n2)] = PTR_UP;
    } else {
        for (int j = n2; j > best_row_idx; j--) P[at(n1, j)] = PTR_LEFT;
    }
    // traceback (:129-150), filled backwards then reversed.  The score of the alignment is the sum of the matrix cells on
    // the path (:136); a cell's value is known where it lies on the border (0), on the last column or row (kept), and
    // otherwise follows from its successor on the path: the successor's pointer is the fill's own there (the fix-up only
    // rewrites pointers ON the last column / row, and those lead along it), so S(pred) = S(cell) - what that move scored.
    int i = n1, j = n2, k = 0;
    int32_t score = 0;
    int32_t tracked = 0;                                 // S(i, j) when the cell is interior
    while (i > 0 || j > 0) {
        const uint8_t p = P[at(i, j)];
        const int32_t here = (i == 0 || j == 0) ? 0 : (j == n2 ? lastcol[i] : (i == n1 ? lastrow[j] : tracked));
        score += here;
        const bool mv_i = (p == PTR_DIAG || p == PTR_UP), mv_j = (p == PTR_DIAG || p == PTR_LEFT);
        tracked = here - (p == PTR_DIAG ? (s1[i > 0 ? i - 1 : 0] == s2[j > 0 ? j - 1 : 0] ? match : mismatch) : gap);
        aln1[k] = mv_i ? s1[i - 1] : '-';
        aln2[k] = mv_j ? s2[j - 1] : '-';
        k++;
        if (mv_i) i--;
        if (mv_j) j--;
    }
    for (int a = 0, b = k - 1; a < b; a++, b--) {
        char t = aln1[a]; aln1[a] = aln1[b]; aln1[b] = t;
        t = aln2[a]; aln2[a] = aln2[b]; aln2[b] = t;
    }
    aln1[k] = 0; aln2[k] = 0;
    *aln_len = k;
    *score_out = score;
    return MCT_OK;
}

struct NwWork { NwScratch scalar; DiagScratch diag; };

int nw_align_impl(const char *s1, int n1, const char *s2, int n2, int match, int mismatch, int gap,
                  char *aln1, char *aln2, int32_t *aln_len, int32_t *score_out, NwWork &w)
{
    auto mag = [](int v) { return (int64_t)(v < 0 ? -(int64_t)v : v); };
    const int64_t worst = ((int64_t)n1 + n2 + 2) * std::max(mag(match), std::max(mag(mismatch), mag(gap)));
    if (worst < 30000)        // every score, and the -10000 sentinel of the fix-up scan, fits 16 bits
        return nw_align_diag(s1, n1, s2, n2, match, mismatch, gap, aln1, aln2, aln_len, score_out, w.diag);
    return nw_align_scalar(s1, n1, s2, n2, match, mismatch, gap, aln1, aln2, aln_len, score_out, w.scalar);
}

inline double qual2prob(int q) { return pow(10, q / (-10.0)); }                       // moira.py:1392-1393
inline int prob2qual(double p) { return (int)floor(-10 * log10(p)); }                 // moira.py:1395-1396

// ref: moira/moira.py:1376-1558
int make_contig_impl(const char *fa, const int32_t *fq, const char *ra, const int32_t *rq, int n,
                     int insert, int deltaq, int consensus, int qcap, int trim,
                     char *contig, int32_t *cq, int32_t *clen, int32_t *overlap, int32_t *gaps_out,
                     int32_t *mism_out, std::vector<int32_t> &fqa, std::vector<int32_t> &rqa)
{
    if (insert <= 0) return fail(MCT_E_INVALID, "insert must be a positive integer");
    if (deltaq <= 0) return fail(MCT_E_INVALID, "deltaq must be a positive integer");
    if (qcap < 0) return fail(MCT_E_INVALID, "qscore_cap must be a non-negative integer");
    if (consensus < 0 || consensus > 2) return fail(MCT_E_INVALID, "consensus_qscore must be \"best\", \"sum\" or \"posterior\".");
    // qualities laid onto the alignment (-1 marks a gap column), :1421-1440
    fqa.assign(n, -1); rqa.assign(n, -1);
    int u = 0;
    for (int k = 0; k < n; k++) if (fa[k] != '-') fqa[k] = fq[u++];
    u = 0;
    for (int k = 0; k < n; k++) if (ra[k] != '-') rqa[k] = rq[u++];
    int fstart = -1, rstart = -1, fend = -1, rend = -1;
    for (int k = 0; k < n; k++) { if (fa[k] != '-') { if (fstart < 0) fstart = k; fend = k; } }
    for (int k = 0; k < n; k++) { if (ra[k] != '-') { if (rstart < 0) rstart = k; rend = k; } }
    if (fstart < 0 || rstart < 0) return fail(MCT_E_INVALID, "an aligned read has no bases");
    int ostart, oend; bool reversed;
    if (fstart < rstart) { ostart = rstart; oend = fend; reversed = false; }     // :1463-1470
    else { ostart = fstart; oend = rend; reversed = true; }
    int m = 0, gaps = 0, mism = 0;
    auto push = [&](char b, int q) { contig[m] = b; cq[m] = q; m++; };
    for (int k = 0; k < n; k++) {
        if (k < ostart) {
            if (!trim) { if (reversed) push(ra[k], rqa[k]); else push(fa[k], fqa[k]); }
        } else if (k > oend) {
            if (!trim) { if (reversed) push(fa[k], fqa[k]); else push(ra[k], rqa[k]); }
        } else if (fa[k] == '-') {                                                 // :1498-1507
            gaps++;
            if (consensus == MCT_CONSENSUS_POSTERIOR) push('N', 2);
            else if (rqa[k] > insert) push(ra[k], rqa[k]);
        } else if (ra[k] == '-') {                                                 // :1509-1518
            gaps++;
            if (consensus == MCT_CONSENSUS_POSTERIOR) push('N', 2);
            else if (fqa[k] > insert) push(fa[k], fqa[k]);
        } else if (fa[k] == ra[k]) {                                               // :1520-1531
            int q;
            if (consensus == MCT_CONSENSUS_SUM) q = fqa[k] + rqa[k];
            else if (consensus == MCT_CONSENSUS_POSTERIOR) {
                const double p1 = qual2prob(fqa[k]), p2 = qual2prob(rqa[k]);
                q = prob2qual((p1 * p2 / 3) / (1 - p1 - p2 + (4 * p1 * p2 / 3)));
            } else q = fqa[k] >= rqa[k] ? fqa[k] : rqa[k];
            push(fa[k], q);
        } else {                                                                   // :1533-1556
            mism++;
            if (consensus != MCT_CONSENSUS_POSTERIOR) {
                const int d = fqa[k] - rqa[k];
                if ((d < 0 ? -d : d) < deltaq) push('N', 2);
                else if (fqa[k] >= rqa[k]) push(fa[k], fqa[k]);
                else push(ra[k], rqa[k]);
            } else if (fqa[k] == rqa[k]) {
                push('N', 2);
            } else {
                double p1, p2; char b;
                if (fqa[k] > rqa[k]) { p1 = qual2prob(fqa[k]); p2 = qual2prob(rqa[k]); b = fa[k]; }
                else { p2 = qual2prob(fqa[k]); p1 = qual2prob(rqa[k]); b = ra[k]; }
                push(b, prob2qual(p1 * (1 - p2 / 3) / (p1 + p2 - (4 * p1 * p2 / 3))));
            }
        }
    }
    if (qcap) for (int k = 0; k < m; k++) if (!(cq[k] < qcap)) cq[k] = qcap;      // :1555-1556
    *clen = m; *overlap = oend - ostart; *gaps_out = gaps; *mism_out = mism;
    return MCT_OK;
}

}  // namespace

extern "C" int mct_nw_align(const char *seq1, int32_t len1, const char *seq2, int32_t len2,
                            int32_t match, int32_t mismatch, int32_t gap, char *aln1, char *aln2,
                            int32_t *aln_len, int32_t *score)
{
    if (len1 < 0 || len2 < 0 || !aln1 || !aln2 || !aln_len || !score) return fail(MCT_E_INVALID, "bad arguments");
    // scratch kept per thread: a fresh 190 KB per call sat right at glibc's trim threshold for 2 x 230..256-base pairs
    // (the heap was shrunk and regrown on every call: 19 us per alignment instead of 7.5)
    static thread_local NwWork w;
    return nw_align_impl(seq1, len1, seq2, len2, match, mismatch, gap, aln1, aln2, aln_len, score, w);
}

extern "C" int mct_nw_align_scalar(const char *seq1, int32_t len1, const char *seq2, int32_t len2,
                                   int32_t match, int32_t mismatch, int32_t gap, char *aln1, char *aln2,
                                   int32_t *aln_len, int32_t *score)
{
    if (len1 < 0 || len2 < 0 || !aln1 || !aln2 || !aln_len || !score) return fail(MCT_E_INVALID, "bad arguments");
    NwScratch sc;
    return nw_align_scalar(seq1, len1, seq2, len2, match, mismatch, gap, aln1, aln2, aln_len, score, sc);
}

extern "C" int mct_make_contig(const char *fa, const int32_t *fq, const char *ra, const int32_t *rq,
                               int32_t aln_len, int32_t insert, int32_t deltaq, int32_t consensus,
                               int32_t qscore_cap, int32_t trim_overlap, char *contig,
                               int32_t *contig_quals, int32_t *contig_len, int32_t *overlap_length,
                               int32_t *gaps, int32_t *mismatches)
{
    if (aln_len < 0 || !fa || !ra || !contig || !contig_quals) return fail(MCT_E_INVALID, "bad arguments");
    std::vector<int32_t> a, b;
    return make_contig_impl(fa, fq, ra, rq, aln_len, insert, deltaq, consensus, qscore_cap, trim_overlap,
                            contig, contig_quals, contig_len, overlap_length, gaps, mismatches, a, b);
}

extern "C" int mct_contigs_batch(int64_t n, const char *fwd_seq, const int32_t *fwd_qual, const int64_t *fwd_off,
                                 const char *rev_seq, const int32_t *rev_qual, const int64_t *rev_off,
                                 int32_t match, int32_t mismatch, int32_t gap, int32_t insert, int32_t deltaq,
                                 int32_t consensus, int32_t qscore_cap, int32_t trim_overlap, int32_t threads,
                                 int32_t cap, char *contigs, int32_t *contig_quals, int32_t *contig_len,
                                 int32_t *overlap_length, int32_t *gaps, int32_t *mismatches)
{
    if (n < 0 || cap <= 0) return fail(MCT_E_INVALID, "bad arguments");
    if (threads < 1) threads = 1;
    if (threads > n && n > 0) threads = (int32_t)n;
    std::vector<int> rc(threads, MCT_OK);
    std::vector<std::string> msgs(threads);
    auto work = [&](int t) {
        NwWork sc;
        std::vector<char> rseq, a1, a2;
        std::vector<int32_t> rq, qa, qb;
        for (int64_t i = t; i < n; i += threads) {
            const int l1 = (int)(fwd_off[i + 1] - fwd_off[i]), l2 = (int)(rev_off[i + 1] - rev_off[i]);
            rseq.resize(l2 + 1); rq.resize(l2 + 1); a1.resize(l1 + l2 + 2); a2.resize(l1 + l2 + 2);
            int r = mct_reverse_complement(rev_seq + rev_off[i], rev_qual + rev_off[i], l2, rseq.data(), rq.data());
            int32_t alen = 0, score = 0, clen = 0;
            if (!r) r = nw_align_impl(fwd_seq + fwd_off[i], l1, rseq.data(), l2, match, mismatch, gap,
                                      a1.data(), a2.data(), &alen, &score, sc);
            if (!r && alen > cap) r = fail(MCT_E_BUFFER, "contig %lld needs %d > %d slots", (long long)i, alen, cap);
            if (!r) r = make_contig_impl(a1.data(), fwd_qual + fwd_off[i], a2.data(), rq.data(), alen, insert,
                                         deltaq, consensus, qscore_cap, trim_overlap,
                                         contigs + i * (int64_t)cap, contig_quals + i * (int64_t)cap, &clen,
                                         &overlap_length[i], &gaps[i], &mismatches[i], qa, qb);
            if (r) { rc[t] = r; msgs[t] = g_err; return; }
            contig_len[i] = clen;
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (int t = 0; t < threads; t++)
        if (rc[t]) return fail(rc[t], "%s", msgs[t].c_str());
    return MCT_OK;
}

extern "C" int mct_contigs_from_fastq(int64_t n, const char *fbuf, const int64_t *fidx, const char *rbuf,
                                      const int64_t *ridx, int32_t fastq_offset, int32_t match, int32_t mismatch,
                                      int32_t gap, int32_t insert, int32_t deltaq, int32_t consensus,
                                      int32_t qscore_cap, int32_t trim_overlap, int32_t threads, int64_t rec_cap,
                                      char *out_buf, int64_t *out_idx, int32_t *overlap_length, int32_t *gaps,
                                      int32_t *mismatches)
{
    if (n < 0 || rec_cap <= 0 || (n > 0 && (!fbuf || !fidx || !rbuf || !ridx || !out_buf || !out_idx ||
                                            !overlap_length || !gaps || !mismatches)))
        return fail(MCT_E_INVALID, "bad arguments");
    if (threads < 1) threads = 1;
    if (threads > n && n > 0) threads = (int32_t)n;
    std::vector<int> rc(threads, MCT_OK);
    std::vector<std::string> msgs(threads);
    // Scratch of worker t outlives the call: a fresh 300 KB score matrix per thread and call is mostly page
    // faults, which do not scale over a few hundred threads.
    struct Scratch {
        NwWork sc;
        std::vector<char> rseq, a1, a2, contig;
        std::vector<int32_t> fq, rq_raw, rq, qa, qb, cq;
    };
    static std::mutex pool_lock;
    static std::vector<std::unique_ptr<Scratch>> scratch_pool;
    std::unique_lock<std::mutex> busy(pool_lock);              // one batch at a time owns the pool
    while ((int)scratch_pool.size() < threads) scratch_pool.emplace_back(new Scratch());
    auto work = [&](int t) {
        Scratch &S = *scratch_pool[t];
        NwWork &sc = S.sc;
        auto &rseq = S.rseq; auto &a1 = S.a1; auto &a2 = S.a2; auto &contig = S.contig;
        auto &fq = S.fq; auto &rq_raw = S.rq_raw; auto &rq = S.rq; auto &qa = S.qa; auto &qb = S.qb; auto &cq = S.cq;
        // contiguous shares: neighbouring output slots belong to one thread
        const int64_t i0 = n * t / threads, i1 = n * (t + 1) / threads;
        for (int64_t i = i0; i < i1; i++) {
            const int64_t *f = fidx + i * MIO_IDX_COLS, *r = ridx + i * MIO_IDX_COLS;
            const int l1 = (int)f[MIO_SEQ_LEN], l2 = (int)r[MIO_SEQ_LEN];
            fq.resize(l1 + 1); rq_raw.resize(l2 + 1); rq.resize(l2 + 1); rseq.resize(l2 + 1);
            a1.resize(l1 + l2 + 2); a2.resize(l1 + l2 + 2); contig.resize(l1 + l2 + 2); cq.resize(l1 + l2 + 2);
            const unsigned char *fql = (const unsigned char *)fbuf + f[MIO_QUAL_OFF];
            const unsigned char *rql = (const unsigned char *)rbuf + r[MIO_QUAL_OFF];
            int lo = 0;
            for (int k = 0; k < l1; k++) { fq[k] = (int)fql[k] - fastq_offset; lo |= fq[k]; }      // moira.py:1177
            for (int k = 0; k < l2; k++) { rq_raw[k] = (int)rql[k] - fastq_offset; lo |= rq_raw[k]; }  // :1189
            int rr = MCT_OK;
            if (lo < 0) rr = fail(MCT_E_RANGE, "negative quality in pair %lld", (long long)i);
            if (!rr) rr = mct_reverse_complement(rbuf + r[MIO_SEQ_OFF], rq_raw.data(), l2, rseq.data(), rq.data());
            int32_t alen = 0, score = 0, clen = 0;
            if (!rr) rr = nw_align_impl(fbuf + f[MIO_SEQ_OFF], l1, rseq.data(), l2, match, mismatch, gap,
                                        a1.data(), a2.data(), &alen, &score, sc);
            if (!rr) rr = make_contig_impl(a1.data(), fq.data(), a2.data(), rq.data(), alen, insert, deltaq, consensus,
                                           qscore_cap, trim_overlap, contig.data(), cq.data(), &clen,
                                           &overlap_length[i], &gaps[i], &mismatches[i], qa, qb);
            const int64_t hl = f[MIO_HDR_LEN];
            if (!rr && hl + 2 * (int64_t)clen > rec_cap)
                rr = fail(MCT_E_BUFFER, "contig %lld needs %lld > %lld bytes", (long long)i, (long long)(hl + 2 * (int64_t)clen), (long long)rec_cap);
            if (!rr) {
                char *slot = out_buf + i * rec_cap;
                memcpy(slot, fbuf + f[MIO_HDR_OFF], (size_t)hl);
                memcpy(slot + hl, contig.data(), (size_t)clen);
                unsigned char *qo = (unsigned char *)slot + hl + clen;
                int hi = 0;
                for (int k = 0; k < clen; k++) {
                    const int b = cq[k] + fastq_offset;
                    hi |= (255 - b) | cq[k];                 // sign bit set <=> b > 255 or q < 0
                    qo[k] = (unsigned char)b;
                }
                if (hi < 0) rr = fail(MCT_E_RANGE, "a quality of contig %lld does not fit one byte", (long long)i);
                int64_t *o = out_idx + i * MIO_IDX_COLS;
                const int64_t base = i * rec_cap;
                o[MIO_HDR_OFF] = base; o[MIO_HDR_LEN] = hl;
                o[MIO_SEQ_OFF] = base + hl; o[MIO_SEQ_LEN] = clen;
                o[MIO_QUAL_OFF] = base + hl + clen; o[MIO_QUAL_LEN] = clen;
            }
            if (rr) { rc[t] = rr; msgs[t] = g_err; return; }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < threads; t++) pool.emplace_back(work, t);
    work(0);
    for (auto &th : pool) th.join();
    for (int t = 0; t < threads; t++)
        if (rc[t]) return fail(rc[t], "%s", msgs[t].c_str());
    return MCT_OK;
}
