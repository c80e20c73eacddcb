/*
 * moira_io.h -- C ABI of libmoira_io.so: the text side of moira's pipeline, in C (host only, no HIP).
 *
 * moira.py reads FASTQ one Python string per line and one Python int per base
 * (ref: moira/moira.py:1152-1204 parse_fastq, :1177 `[ord(x) - offset for x in ...]`) and formats
 * every output record with Python string operations (ref: moira/moira.py:842-970 write_results).
 * At 2 x 10^9 reads/s in the filter that text handling is the whole run time, so the CLI keeps a
 * chunk of the input file as ONE byte buffer and lets these functions
 *   1. index the records in it (offset/length of header token, sequence line, quality line),
 *   2. pack selected records straight into the uint8 quality matrix the filter takes
 *      (same rules as mpb_pack_read_ascii in moira_pb.h),
 *   3. format selected records as fasta / qual / fastq text into a caller-owned buffer.
 * Nothing here allocates or keeps state; every buffer is the caller's.  Return values < 0 are errors;
 * mio_last_error() holds the message (thread-local).
 */
#ifndef MOIRA_IO_H
#define MOIRA_IO_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MIO_OK 0
#define MIO_E_INVALID (-1)      /* bad argument */
#define MIO_E_UNSUPPORTED (-2)  /* content the byte-level parser does not reproduce: fall back to the line parser */
#define MIO_E_RANGE (-3)        /* a quality outside 0..254 */
#define MIO_E_SPACE (-4)        /* output buffer too small; the call reports the size it needs */

/* per-record problems found by mio_fastq_index, in the order the reference checks them
 * (ref: moira/moira.py:1178-1183) */
#define MIO_REC_OK 0
#define MIO_REC_EMPTY_SEQ 1
#define MIO_REC_EMPTY_QUAL 2
#define MIO_REC_LENGTH_MISMATCH 3

/* columns of one row of the record index (int64 each) */
#define MIO_HDR_OFF 0   /* header token: after strip(), up to the first space/tab, leading '@'s dropped */
#define MIO_HDR_LEN 1   /*   (':' -> '_' is applied when the header is emitted, ref: moira/moira.py:1175) */
#define MIO_SEQ_OFF 2   /* sequence line, stripped */
#define MIO_SEQ_LEN 3
#define MIO_QUAL_OFF 4  /* quality line, stripped */
#define MIO_QUAL_LEN 5
#define MIO_IDX_COLS 6

const char *mio_version(void);
const char *mio_last_error(void);

/*
 * Index the complete 4-line records in buf[0, len).  Lines end with "\n" or "\r\n"; `final` != 0
 * means the buffer ends the file, so an unterminated last line counts.  Every 4 lines are one record
 * whatever they contain, as in the reference's parser (ref: moira/moira.py:1166-1174), and trailing
 * lines that do not make a record are ignored.
 *   idx        [max_records][MIO_IDX_COLS]
 *   consumed   bytes of buf covered by the returned records (the caller keeps the rest)
 *   bad_kind   MIO_REC_* of the first record that fails the reference's checks (indexing stops
 *              there: the returned count is that record's position), else MIO_REC_OK
 * Returns the number of records indexed, or MIO_E_UNSUPPORTED when the chunk holds a lone "\r"
 * (universal-newline line break) or a byte >= 0x80 (the text-mode parser decodes and strips by
 * Unicode rules there).
 */
int64_t mio_fastq_index(const char *buf, int64_t len, int32_t final, int64_t max_records,
                        int64_t *idx, int64_t *consumed, int32_t *bad_kind);

/* The same index built by `threads` threads (newline counts per byte slice -> line numbers -> record starts ->
 * one sequential indexer per range): identical rows, consumed, bad_kind and return value. */
int64_t mio_fastq_index_mt(const char *buf, int64_t len, int32_t final, int64_t max_records,
                           int64_t *idx, int64_t *consumed, int32_t *bad_kind, int32_t threads);

/* Read [offset, offset + len) of a regular file into dst with `threads` concurrent preads.  Returns the bytes read
 * (short only at the end of the file) or a negative MIO_E_*. */
int64_t mio_pread_mt(int32_t fd, int64_t offset, char *dst, int64_t len, int32_t threads);

/* ---- gzip input (ref: moira/moira.py:1058-1090 reads it through Python's gzip module) ----------------------------------
 * A DEFLATE stream cannot be cut across threads, so the stream decoder itself is what bounds a run on compressed input:
 * this one (csrc/inflate.cpp, written from RFC 1951 / 1952) is resumable at any byte of input and output.
 *   in[0, in_len)   the next compressed bytes; in[in_len, in_len + 8) must be READABLE (any content: it is never used as data)
 *   final           != 0: the input ends with this call
 *   out             out[0, hist) holds the last `hist` (<= 32768 is enough) bytes of the output produced so far (what
 *                   matches may reach back into); new output is written to out[hist, out_cap)
 * NOTE for callers: with final == 0 the decoder does not start a piece it cannot see whole -- a gzip header incl. its
 * optional fields (parsed from one contiguous view; FHCRC is skipped, not verified), a block header (it asks for 320
 * bytes), a member trailer (8 bytes) -- so a return of 0 with *in_used == 0 and *out_used == 0 is LEGAL until final != 0:
 * offer the unconsumed tail again with more data behind it (at least 320 bytes beyond the tail, or everything that is left
 * with final = 1).  Only a call made with final != 0 that makes no progress means a truncated file.
 * Returns 0: every complete symbol of the input has been decoded -- call again with the unconsumed tail
 *            in[*in_used, in_len) followed by more data;  1: the output is full (call again with a new output buffer and
 *            the unconsumed input);  2: the gzip file (all members; zero padding ignored) ended cleanly;  < 0: corrupt or
 *            truncated input, mio_inflate_error() says what (CRC-32 and length of every member are checked). */
typedef struct mio_inflate mio_inflate;
mio_inflate *mio_inflate_create(void);
void mio_inflate_destroy(mio_inflate *s);
const char *mio_inflate_error(void);
int32_t mio_inflate_gzip(mio_inflate *s, const uint8_t *in, int64_t in_len, int32_t final, uint8_t *out, int64_t hist,
                         int64_t out_cap, int64_t *in_used, int64_t *out_used);
uint32_t mio_crc32(uint32_t crc, const uint8_t *p, int64_t n);      /* CRC-32 of RFC 1952 (start with 0) */

/* BGZF (SAM/BAM specification section 4.1; what bgzip and Illumina's FASTQ writers produce, and what this package's own
 * .gz outputs are): gzip members of at most 64 KiB that state their compressed size in a 'B','C' extra subfield, so
 * their boundaries are known without decoding and the members inflate on several threads.
 * mio_bgzf_scan walks in[0, in_len) from a member boundary and fills, for up to max_blocks COMPLETE BGZF members whose
 * stated output sizes sum to at most max_out (the first one is always taken), offs[k] (start), sizes[k] (compressed
 * size) and out_offs[k] (start of its output; out_offs[n] = total, so out_offs needs max_blocks + 1 entries).  Returns
 * n; *why says what stopped it: 0 the next member is incomplete (more input needed), 1 the next member is not BGZF
 * (hand the rest to mio_inflate_gzip), 2 a limit.
 * mio_bgzf_inflate_mt decodes those members into out[out_offs[k], out_offs[k+1]) on `threads` threads with the CRC-32
 * and length checks of mio_inflate_gzip; a member that decodes to anything but its stated size is an error.
 * in[in_len, in_len + 8) must be readable, as for mio_inflate_gzip. */
int64_t mio_bgzf_scan(const uint8_t *in, int64_t in_len, int64_t max_blocks, int64_t max_out, int64_t *offs, int32_t *sizes,
                      int64_t *out_offs, int32_t *why);
int32_t mio_bgzf_inflate_mt(const uint8_t *in, const int64_t *offs, const int32_t *sizes, const int64_t *out_offs, int64_t n,
                            uint8_t *out, int32_t threads);

/*
 * Pack records sel[0..nsel) (rows of idx; sel == NULL: records 0..nsel-1) into an nsel x row_stride
 * uint8 matrix: Q = byte - fastq_offset, Q0 -> 1, 'N' -> 0, 'n' -> 255 (or an ordinary base when
 * lower_n_is_base != 0, the Python twin's rule the Poisson path follows, ref: moira/moira.py:1660),
 * at most max_len bases (--truncate; <= 0: no limit), rows zero-padded.
 *   lens_out   int32[nsel]  bases packed
 *   flags_out  uint8[nsel]  bit 0: the packed part holds an upper-case 'N'   (may be NULL)
 * Returns MIO_OK, MIO_E_RANGE (quality < 0 or > 254; *bad_record = position in sel) or MIO_E_INVALID
 * (a read does not fit row_stride).
 */
int32_t mio_pack(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel,
                 int32_t fastq_offset, int32_t max_len, int32_t lower_n_is_base, int64_t row_stride,
                 uint8_t *out, int32_t *lens_out, uint8_t *flags_out, int64_t *bad_record);

/*
 * fasta + qual input (ref: moira/moira.py:1093-1150 parse_fasta_and_qual: one header line and ONE data
 * line per record in each file).  Records complete in both buffers are copied into `out` as
 * header token | sequence | one byte per quality (the integer itself, i.e. FASTQ offset 0), with their
 * index rows in out_idx -- the same shape mio_fastq_index gives, so that everything downstream is
 * shared.  Anything the line parser treats specially (names that differ, empty lines, unequal
 * lengths, tokens that are not plain decimal integers 0..254 separated by single blanks, lone CR,
 * non-ASCII) returns MIO_E_UNSUPPORTED: the caller re-runs the file with the line parser, which raises
 * the reference's exception.  Returns the number of records; *_consumed / *out_used report progress.
 */
int64_t mio_fasta_qual_index(const char *fbuf, int64_t flen, const char *qbuf, int64_t qlen, int32_t final,
                             int64_t max_records, char *out, int64_t out_cap, int64_t *out_idx,
                             int64_t *f_consumed, int64_t *q_consumed, int64_t *out_used);

/* Paired input: position of the first pair whose header tokens differ (the reference's
 * forward_header != reverse_header, ref: moira/moira.py:1197-1198), or -1. */
int64_t mio_first_header_mismatch(const char *fbuf, const int64_t *fidx, const char *rbuf, const int64_t *ridx, int64_t n);

/*
 * hash(str) of 64-bit CPython 2.7 for the (truncated) sequence of records 0..n-1.  The collapse step
 * keeps the reference's output order, which for groups of equal abundance is the slot order of a
 * Python-2 dict keyed by the sequence (ref: moira/moira.py:459-475, :492; moira_amd/py2dict.py).
 */
int32_t mio_py2_hash(const char *buf, const int64_t *idx, int64_t n, int32_t max_len, uint64_t *out);

/* what mio_format writes per record */
#define MIO_FMT_FASTA 0   /* ">hdr[\tlabel]\nSEQ\n"                 ref: moira/moira.py:904,930,939,954,963 */
#define MIO_FMT_QUAL 1    /* ">hdr[\tlabel]\nq q q ...\n"           ref: moira/moira.py:905,931,940,955,964 */
#define MIO_FMT_FASTQ 2   /* "@hdr[\tlabel]\nSEQ\n+\nQUALSTRING\n"  ref: moira/moira.py:902,928,937,952,961 */

/*
 * Format records sel[0..nsel) into out[0, cap).
 *   relabel     NULL: the record's own header (':' -> '_'); else header = relabel + decimal(relabel_index[k])
 *               (ref: moira/moira.py:854-855)
 *   relabel_index  int64[nsel], required with relabel
 *   ee          double[nsel] or NULL: when non-NULL ";ee=%.2f;size=1;" is appended to the header
 *               (USEARCH pipeline, ref: moira/moira.py:858-863)
 *   labels / label_id   label_id int32[nsel] or NULL; label_id[k] >= 0 appends "\t" + labels[label_id[k]]
 *   out_offset  FASTQ offset of the quality string MIO_FMT_FASTQ writes (= fastq_offset for FASTQ input;
 *               records built by mio_fasta_qual_index carry fastq_offset 0)
 *   max_len     bases written (--truncate; <= 0: all)
 *   clamp_q0    != 0: qualities are shown after the Q0 -> 1 clamp, as the reference's writer sees them
 *               after quality control (ref: moira/moira.py:814); 0: as they are (--only_contig never
 *               clamps, ref: moira/moira.py:809-810)
 * Returns the bytes written, or MIO_E_SPACE with *needed set when cap is too small.
 */
int64_t mio_format(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel, int32_t kind,
                   int32_t fastq_offset, int32_t out_offset, int32_t clamp_q0, int32_t max_len, const char *relabel,
                   const int64_t *relabel_index,
                   const double *ee, const char *const *labels, const int32_t *label_id,
                   char *out, int64_t cap, int64_t *needed);

/* contigs.report lines of records sel[0..nsel) that are written one by one (no collapse, n_seqs = 1):
 * header as in mio_format, aux int32[.][3] indexed by RECORD (not by position in sel)
 * (ref: moira/moira.py:868). */
int64_t mio_format_report(const char *buf, const int64_t *idx, const int64_t *sel, int64_t nsel,
                          const char *relabel, const int64_t *relabel_index, const double *ee,
                          const int32_t *aux, char *out, int64_t cap, int64_t *needed);

/* ---- collapse of identical sequences (ref: moira/moira.py:459-475, :490-493) -----------------
 * The reference keeps one dict entry per distinct (truncated) sequence: the representative is the
 * member with the strictly smallest expected errors (first seen wins ties), names_info lists the
 * members' headers (a new representative is inserted in front, everything else appended), and the
 * groups are written by decreasing abundance, equal abundances in Python-2 dict order.
 * A mio_collapse object is that dict, owning copies of what it needs (sequences, the
 * representatives' qualities, every header), so the input chunks can be released. */
typedef struct mio_collapse mio_collapse;
mio_collapse *mio_collapse_create(void);
void mio_collapse_destroy(mio_collapse *c);
int64_t mio_collapse_count(const mio_collapse *c);          /* distinct sequences so far */
/* Threads mio_collapse_add may use (default 1; at most 64).  The groups live in 64 shards picked by the sequence's
 * hash, each filled by one thread with its reads in file order; the Python-2 dict ORDER -- the one sequential thing --
 * is rebuilt at export time from the order in which the distinct sequences first appeared.  Results do not depend on
 * the thread count. */
int32_t mio_collapse_set_threads(mio_collapse *c, int32_t threads);

/* Add records 0..n-1 of a chunk in file order.  ee double[n] (after +Ns / floor), flags uint8[n] from
 * mio_pack (may be NULL), max_len as in mio_pack, aux int32[n][3] = overlap length, gaps, mismatches of
 * each contig (NULL for single reads: zeros); a group carries its representative's. */
int32_t mio_collapse_add(mio_collapse *c, const char *buf, const int64_t *idx, int64_t n, int32_t max_len,
                         const double *ee, const uint8_t *flags, const int32_t *aux);

/* Fix the output order (moira/moira.py:492) and copy out, in that order, per group: the
 * representative's ee, the sequence length, the abundance, the flags and aux[.][3] (any pointer may be
 * NULL). */
int32_t mio_collapse_export(mio_collapse *c, double *ee, int64_t *len, int64_t *size, uint8_t *flags, int32_t *aux);

#define MIO_FMT_REPORT 4  /* "hdr\tn_seqs\toverlap\tgaps\tmismatches\n"   ref: moira/moira.py:866,868 */
#define MIO_FMT_NAMES 3   /* "hdr\tname,name,...\n"   ref: moira/moira.py:880,919,933,943,957,967 */

/* mio_format for groups sel[0..nsel) (positions in the exported order).  Header = the
 * representative's, or relabel + (position + 1) (ref: moira/moira.py:493, :854-855); with usearch != 0
 * ";ee=%.2f;size=%d;" is appended (size = abundance, ref: moira/moira.py:858-863).  MIO_FMT_NAMES
 * writes the mothur names line of each group; lstrip_gt uint8[nsel] (may be NULL) drops leading '>'
 * from that line's header where the reference does (ref: moira/moira.py:880,894,907,943). */
int64_t mio_collapse_format(const mio_collapse *c, const int64_t *sel, int64_t nsel, int32_t kind,
                            int32_t fastq_offset, int32_t out_offset, int32_t clamp_q0, const char *relabel, int32_t usearch,
                            const char *const *labels, const int32_t *label_id, const uint8_t *lstrip_gt,
                            char *out, int64_t cap, int64_t *needed);

#ifdef __cplusplus
}
#endif
#endif
