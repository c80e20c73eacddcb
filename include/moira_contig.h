/*
 * moira_contig.h -- C ABI of libmoira_contig.so: CPU contig construction for paired reads.
 *
 * BASELINE.json's north_star keeps Needleman-Wunsch contig construction on the CPU; this is the
 * build's own C++ version of it, feeding contigs straight into the packed
 * quality matrix the HIP filter consumes (include/moira_pb.h).  No GPU code here.
 *
 * Reference interfaces replaced (paths relative to the reference checkout):
 *   reverse_complement(sequence, quals)                  moira/moira.py:1207-1235
 *   nw_align(seq1, seq2, match, mismatch, gap)           moira/nw_align.pyx:49-201
 *                                                        (pure-Python twin moira/moira.py:1238-1373)
 *   make_contig(fwd_aln, fwd_q, rev_aln, rev_q, insert, deltaq, consensus_qscore, qscore_cap,
 *               trim_overlap)                            moira/moira.py:1376-1558
 * All functions return 0 or a negative MCT_E_* code; mct_last_error() gives the message.
 */
#ifndef MOIRA_CONTIG_H
#define MOIRA_CONTIG_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define MCT_OK          0
#define MCT_E_INVALID  -1
#define MCT_E_BASE     -2   /* not an IUPAC base (ValueError in the reference, moira.py:1229-1230) */
#define MCT_E_NOMEM    -4
#define MCT_E_BUFFER   -6   /* output buffer too small */
#define MCT_E_RANGE    -7   /* a quality that does not fit one byte at the given FASTQ offset */

#define MCT_CONSENSUS_BEST      0
#define MCT_CONSENSUS_SUM       1
#define MCT_CONSENSUS_POSTERIOR 2

const char *mct_last_error(void);

/* Reverse complement `len` bases into out (and reverse quals into out_quals when both non-NULL). */
int mct_reverse_complement(const char *seq, const int32_t *quals, int32_t len,
                           char *out_seq, int32_t *out_quals);

/* Global alignment, mothur flavour (first row/column zero, 3' overlap fix-up).  aln1/aln2 must
 * hold len1 + len2 + 1 bytes; they are NUL terminated.  *score = sum of the score-matrix cells on
 * the traceback path (the reference's definition, nw_align.pyx:136). */
int mct_nw_align(const char *seq1, int32_t len1, const char *seq2, int32_t len2,
                 int32_t match, int32_t mismatch, int32_t gap,
                 char *aln1, char *aln2, int32_t *aln_len, int32_t *score);

/* The same alignment by the plain row-by-row loop with 32-bit scores.  mct_nw_align walks the matrix
 * by anti-diagonals with 16-bit scores (SIMD) and switches to this form by itself when
 * (len1 + len2) x the largest |parameter| could leave 16 bits; exported so that the two can be
 * compared. */
int mct_nw_align_scalar(const char *seq1, int32_t len1, const char *seq2, int32_t len2,
                        int32_t match, int32_t mismatch, int32_t gap,
                        char *aln1, char *aln2, int32_t *aln_len, int32_t *score);

/* Consensus of two aligned reads.  contig/contig_quals must hold aln_len entries. */
int mct_make_contig(const char *fwd_aln, const int32_t *fwd_quals, const char *rev_aln,
                    const int32_t *rev_quals, int32_t aln_len, int32_t insert, int32_t deltaq,
                    int32_t consensus, int32_t qscore_cap, int32_t trim_overlap,
                    char *contig, int32_t *contig_quals, int32_t *contig_len,
                    int32_t *overlap_length, int32_t *gaps, int32_t *mismatches);

/* The paired half of process_data (moira/moira.py:789-801) for a batch, multithreaded:
 * reverse-complement the reverse read, align, build the contig.  Inputs are concatenated strings
 * with offsets (read i = [off[i], off[i+1])).  Outputs: contig strings/quals in caller buffers of
 * capacity cap_per_contig each (row i at i*cap_per_contig), lengths and the three report numbers. */
int mct_contigs_batch(int64_t n, const char *fwd_seq, const int32_t *fwd_qual, const int64_t *fwd_off,
                      const char *rev_seq, const int32_t *rev_qual, const int64_t *rev_off,
                      int32_t match, int32_t mismatch, int32_t gap, int32_t insert, int32_t deltaq,
                      int32_t consensus, int32_t qscore_cap, int32_t trim_overlap, int32_t threads,
                      int32_t cap_per_contig, char *contigs, int32_t *contig_quals,
                      int32_t *contig_len, int32_t *overlap_length, int32_t *gaps, int32_t *mismatches);

/* The same for a chunk of paired FASTQ records that are still text: fbuf/fidx and rbuf/ridx are a file
 * buffer and its record index as mio_fastq_index (include/moira_io.h) produces them.  Qualities are
 * byte - fastq_offset (ref: moira/moira.py:1177,1189), unclamped, as process_data hands them to
 * make_contig.  Output is again a buffer + record index, so that every mio_* function treats a contig
 * like a read: slot i of out_buf (rec_cap bytes) holds the forward header token, the contig and its
 * qualities as bytes (q + fastq_offset); out_idx[i] = the MIO_* columns into out_buf.
 * Returns MCT_E_RANGE when a quality is negative or q + fastq_offset > 255, MCT_E_BUFFER when a
 * record does not fit rec_cap. */
int mct_contigs_from_fastq(int64_t n, const char *fbuf, const int64_t *fidx, const char *rbuf, const int64_t *ridx,
                           int32_t fastq_offset, int32_t match, int32_t mismatch, int32_t gap, int32_t insert,
                           int32_t deltaq, int32_t consensus, int32_t qscore_cap, int32_t trim_overlap,
                           int32_t threads, int64_t rec_cap, char *out_buf, int64_t *out_idx,
                           int32_t *overlap_length, int32_t *gaps, int32_t *mismatches);

#ifdef __cplusplus
}
#endif
#endif
