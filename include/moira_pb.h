/*
 * moira_pb.h -- C ABI of libmoira_pb.so: the MI355X (gfx950) Poisson-binomial
 * read-quality filter.
 *
 * This is the drop-in boundary for ONE path of fpusan/moira: what
 * moira/bernoullimodule.c (the CPython-2 extension `bernoulli`) and
 * moira.py's per-read dispatch compute today.  Every entry point below names
 * the reference interface it replaces as `ref: file:line` (paths are relative
 * to the reference checkout).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes, no C++/torch types.
 *   - Every function returns an int status: MPB_OK (0) or a negative MPB_E_*.
 *     mpb_last_error() returns a thread-local human-readable message for the
 *     last failing call on the calling thread.  Nothing throws across the ABI.
 *   - The caller owns every buffer it passes.  The context owns its device
 *     workspace, LUTs, stream and events and frees them in mpb_destroy().
 *   - One context per device; calls on one context must not overlap (one host
 *     thread per device, as moira's Pool workers were one process per task).
 *   - There is NO CPU implementation in this library.  If no HIP device is
 *     usable, mpb_create() fails with MPB_E_NODEVICE; nothing falls back.
 *
 * Quality-matrix encoding ("packed qscores"), produced by mpb_pack_read():
 *   one read = one row of `row_stride` bytes (row_stride % 16 == 0),
 *   byte k < len :  1..254  Phred score Q (Q==0 was clamped to 1,
 *                            ref: moira/bernoullimodule.c:104-107, moira/moira.py:814)
 *                   0        the base is 'N' (ambiguous; skipped by the DP,
 *                            ref: moira/bernoullimodule.c:196-199)
 *                   255      the base is 'n' (counted as ambiguous by the C
 *                            reference, ref: moira/bernoullimodule.c:196, but
 *                            not by `--ambigs disallow`, ref: moira/moira.py:911)
 *   byte k >= len:  ignored (the kernels mask it; 0 is conventional).
 */
#ifndef MOIRA_PB_H
#define MOIRA_PB_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: exactly the entry points declared in this header are exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define MPB_OK            0
#define MPB_E_INVALID    -1   /* bad argument (ValueError on the Python side)   */
#define MPB_E_NODEVICE   -2   /* no usable HIP device / bad device id           */
#define MPB_E_HIP        -3   /* a HIP runtime call failed (message has detail) */
#define MPB_E_NOMEM      -4   /* device or host allocation failed               */
#define MPB_E_RANGE      -5   /* a qscore cannot be encoded (Q < 0; Q > 254 in a batch entry) */

/* --ambigs modes, ref: moira/moira.py:658-659, :827-828, :911 */
#define MPB_AMBIG_TREAT_AS_ERRORS 0
#define MPB_AMBIG_IGNORE          1
#define MPB_AMBIG_DISALLOW        2

/* flags for mpb_filter_params.flags */
#define MPB_FLAG_ROUND      1u   /* --round: floor(ee) before the compare, ref: moira/moira.py:830-831 */
#define MPB_FLAG_FAST_FMA   2u   /* contract a*v+b*w into one fma: 2 FP64 ops per cell instead of 3.  ee is then close to,
                                    not bit-identical with, the reference's: the interpolation divides by the mass of
                                    the crossing row, which is of order alpha, so the error grows like 1/alpha (worst
                                    relative difference over 3000 random reads: 6e-14 at alpha 0.005, 1.4e-10 at 1e-6,
                                    1e-7 at 1e-9).  The flag is therefore ACCEPTED ONLY FOR alpha >= 1e-5 (MPB_E_INVALID
                                    below), where ee stays within north_star's 1e-9 relative tolerance and the guard
                                    band below keeps every pass/fail flag EQUAL to the exact computation's: a read
                                    whose ee lands within 1e-9 relative of the threshold (or of an integer with
                                    MPB_FLAG_ROUND) is recomputed with the three-rounding arithmetic.  Off by default. */
#define MPB_FLAG_DECISION_ONLY 8u /* opt-in, NOT the reference's contract: a read whose expected errors are
                                    PROVABLY above the threshold (multiplicative Chernoff lower-tail bound on the
                                    Poisson-binomial quantile, from the prepass' mean) is reported pass = 0,
                                    ee = +infinity without running its DP (a NaN still means what it means without
                                    the flag: the CDF never crossed -- the reference's ReturnedNaNError).  Every other read is computed exactly as
                                    usual, and every pass/fail flag equals the full computation's.  For
                                    pipelines that never look at the ee of a discarded read. */
#define MPB_FLAG_TEST_UNDERPREDICT 4u /* test hook: halve every predicted row budget so that the
                                         overflow (second) pass is exercised; results are unchanged */
#define MPB_FLAG_COUNT_CELLS 32u /* diagnostic (bench.py's fp64_valu.frac_algorithmic): mpb_filter_device also sums, over the reads
                                    its tile kernel reports, the DP cells the ALGORITHM needs -- sum_k min(k + 1, J) over a read's
                                    scored bases, J = rows up to and including the one whose CDF crosses 1 - alpha --
                                    as opposed to the cells its row-budget class pays for.  Fetched with
                                    mpb_last_algorithmic_cells().  Reads settled elsewhere (k_wide, MPB_FLAG_DECISION_ONLY) are
                                    not counted.  Results are unchanged. */
#define MPB_FLAG_BATCHED_ONLY 16u /* mpb_filter_host sends batches of <= 4096 reads through one launch with one
                                     read per wave (latency: what a per-read caller sees) instead of the sorted,
                                     tiled pipeline; this flag forces the pipeline.  Results are identical. */

#define MPB_FLAG_NO_NARROW 64u   /* never take the natural-order narrow pass (below): the call runs the sorted, tiled pipeline whatever
                                    the batch looks like.  For callers of the introspection entries that describe that pipeline
                                    (mpb_last_class_histogram, mpb_last_read_budgets, MPB_FLAG_COUNT_CELLS).  Results are identical. */
#define MPB_FLAG_NARROW_ROWS(r) (((uint32_t)(r) & 15u) << 8)   /* test / measurement hook: take the narrow pass with r rows (2..4)
                                    whatever the sample says and however small the batch; reads it cannot finish still go through the
                                    sorted pipeline, so results are identical. */

#define MPB_FLAG_NARROW_SPLIT(c) (((uint32_t)(c) & 255u) << 12)   /* test / measurement hook, with MPB_FLAG_NARROW_ROWS(r >= 3) on a
                                    ragged batch: groups of the pass whose longest read has at most c 16-byte chunks run with r - 1
                                    rows ("mixed rows"; the library picks c from its sample by itself).  Results are identical. */

/* kernel ids for mpb_kernel_time() */
#define MPB_K_PREPASS   0   /* lambda/sigma/Ns estimate + row classing        */
#define MPB_K_SCAN      1   /* class histogram scan / tile table              */
#define MPB_K_SCATTER   2   /* stable scatter of read indices by class        */
#define MPB_K_DP        3   /* the Poisson-binomial DP + epilogue (dominant)  */
#define MPB_K_OVERFLOW  4   /* re-run of reads whose predicted row count was too small */
#define MPB_K_LAMBDA    5   /* Poisson approximation: per-read sum of error probabilities */
#define MPB_K_WIDE      6   /* reads that need more than 1024 DP rows: one workgroup per read */
#define MPB_K_NARROW    7   /* natural-order narrow pass: one read per lane, 2..4 DP rows, the matrix read once */
#define MPB_K_FALLBACK  8   /* (rounds 5: the gather of the unfinished reads and the scatter of their results; unused since round 6: they run where they lie) */
#define MPB_K_SAMPLE    9   /* ... the batch sample that picks the pass */
#define MPB_K_COUNT     10

typedef struct mpb_ctx mpb_ctx;

/*
 * Filter parameters = the filter-relevant moira flags.
 *   alpha      --alpha     (0,1)   ref: moira/moira.py:668,736; moira/bernoullimodule.c:79-83
 *   uncert     --uncert    (0,1]   ref: moira/moira.py:663-665,949-950
 *   maxerrors  --maxerrors >0 or NaN when unset (then uncert mode)
 *                                  ref: moira/moira.py:666,925-926
 *   ambig_mode --ambigs            ref: moira/moira.py:658,827-828,911
 *   flags      MPB_FLAG_*
 */
typedef struct mpb_filter_params {
    double   alpha;
    double   uncert;
    double   maxerrors;
    int32_t  ambig_mode;
    uint32_t flags;
} mpb_filter_params;

/* Totals of one filter call (what moira prints at moira/moira.py:508-519). */
typedef struct mpb_filter_counts {
    int64_t n_reads;
    int64_t n_pass;
    int64_t n_fail;
    int64_t n_overflow;   /* reads that needed the second (wide) pass; diagnostic */
} mpb_filter_counts;

/* ---- library ------------------------------------------------------------ */
const char *mpb_version(void);
const char *mpb_last_error(void);
/* number of visible HIP devices (0 when there is none; never fails) */
int mpb_device_count(void);

/* ---- context ------------------------------------------------------------ */
/* Creates the per-device context: stream, events, the 256-entry {1-p, p'} LUT
 * built on the host with libm pow exactly as ref: moira/bernoullimodule.c:202
 * and :140-145 do, uploaded once. */
int mpb_create(int device_id, mpb_ctx **out);
int mpb_destroy(mpb_ctx *ctx);
/* The table itself, for pinning it in tests ("the LUT values themselves are part of the fixtures"):
 * a_out[q] = pow(1 - p, 1), b_out[q] = ((1-1+1)/(1.0*1)) * (p/(1-p)) * pow(1 - p, 1) with p = pow(10, q / -10.0),
 * ref: moira/bernoullimodule.c:202,140-145; entries 0 ('N') and 255 ('n') are the identity step {1, 0}.
 * mpb_host_lut computes it on the host exactly as mpb_create does (needs no device); mpb_device_lut reads back
 * what the context uploaded.  256 doubles each. */
int mpb_host_lut(double *a_out, double *b_out);
int mpb_device_lut(mpb_ctx *ctx, double *a_out, double *b_out);
/* The HIP stream the context launches on (as void*), for callers that want to
 * order their own work (e.g. torch) against it. */
int mpb_stream(mpb_ctx *ctx, void **stream_out);
int mpb_synchronize(mpb_ctx *ctx);

/* ---- device memory plumbing (so a ctypes-only host needs no torch) -------- */
int mpb_malloc(mpb_ctx *ctx, int64_t bytes, void **dptr_out);
int mpb_free(mpb_ctx *ctx, void *dptr);
int mpb_memcpy_h2d(mpb_ctx *ctx, void *dst_dev, const void *src_host, int64_t bytes);
int mpb_memcpy_d2h(mpb_ctx *ctx, void *dst_host, const void *src_dev, int64_t bytes);
int mpb_memset(mpb_ctx *ctx, void *dst_dev, int value, int64_t bytes);
/* Pinned (page-locked) host memory.  A batch handed to mpb_filter_host from such a buffer is DMA-ed where
 * it lies; one in ordinary pageable memory is first copied into the context's pinned staging blocks (by a few
 * threads, overlapped with the GPU work of the chunks before).  A parser that packs its reads straight into a
 * pinned matrix (ref: moira/moira.py:1177 builds one Python list per read instead) saves that copy.
 * mpb_host_free waits for the context's streams first. */
int mpb_host_alloc(mpb_ctx *ctx, int64_t bytes, void **hptr_out);
int mpb_host_free(mpb_ctx *ctx, void *hptr);

/* ---- packing (host, integer only) ---------------------------------------- */
/* Encode one read into a row of the quality matrix.
 * Replaces the list->int[] marshalling + Q0 clamp of
 * ref: moira/bernoullimodule.c:92-108 and the N test of :196.
 * `seq` may be NULL (no ambiguous bases).  Returns MPB_E_RANGE if a score is
 * negative (the Python reference raises ValueError, ref: moira/moira.py:1603-1604)
 * or above 254. Bytes [len, row_bytes) are zeroed. */
int mpb_pack_read(const char *seq, const int32_t *quals, int32_t len,
                  uint8_t *row_out, int32_t row_bytes);
/* Same from a raw FASTQ quality string (ASCII, `offset` = --fastq_offset,
 * ref: moira/moira.py:1177). */
int mpb_pack_read_ascii(const char *seq, const char *qual_ascii, int32_t len,
                        int32_t offset, uint8_t *row_out, int32_t row_bytes);

/* Batch form for a parser: n reads concatenated (read i = bytes [off[i], off[i+1]) of seq_cat and
 * qual_cat), FASTQ ASCII qualities, packed into n rows of row_stride bytes; at most `max_len`
 * bases of each read are kept (--truncate, ref: moira/moira.py:806-807; <= 0: no limit).
 * lens_out[i] = bases packed.  One C loop instead of one Python list per read
 * (ref: moira/moira.py:1177 builds `[ord(x) - offset for x in ...]` per read). */
int mpb_pack_batch_ascii(const char *seq_cat, const char *qual_cat, const int64_t *off, int64_t n,
                         int32_t fastq_offset, int32_t max_len, int64_t row_stride,
                         uint8_t *out, int32_t *lens_out);

/* On-device decode: raw ASCII quality bytes and base letters already in HBM
 * (two n x row_stride matrices) -> the packed quality matrix, elementwise, same rules as
 * mpb_pack_read_ascii.  d_err (device int32, may be NULL) counts bytes that decode to Q < 0 or
 * Q > 254 (those are written as Q1 / Q254). */
int mpb_decode_ascii_device(mpb_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual_ascii, int64_t n,
                            int64_t row_stride, const int32_t *d_len, int32_t fixed_len,
                            int32_t fastq_offset, uint8_t *d_q_out, int32_t *d_err);

/* The inverse, for tests and benchmarks that need FASTQ-text matrices resident in HBM: packed byte 0 -> ('N', offset+2),
 * 255 -> ('n', offset+2), Q -> (one of ACGT, chr(Q + offset)).  mpb_decode_ascii_device of its output gives the packed
 * matrix back for every Q <= 255 - offset. */
int mpb_encode_ascii_device(mpb_ctx *ctx, const uint8_t *d_q, int64_t n, int64_t row_stride, int32_t fastq_offset,
                            uint8_t *d_seq_out, uint8_t *d_qual_out);

/* ---- the hot path --------------------------------------------------------- */
/*
 * Filter a batch that is RESIDENT IN HBM.
 *   d_q         device, n rows of row_stride bytes (row_stride % 16 == 0, 16-B aligned; a 64-B aligned
 *               matrix with row_stride % 64 == 0 is the fast layout: every 64-byte lane group then reads one
 *               sector -- a matrix offset by 16 bytes costs the prepass 7 %; and the narrow pass of batches of good reads
 *               then walks whole 128-byte lines at every row count, at other strides only with two rows; for RAGGED batches
 *               (d_len != NULL) make row_stride a multiple of 128: the narrow pass of ragged batches fetches, per read, only the
 *               lines its bases lie in, and a row that starts inside a line shares that line with its neighbour)
 *   d_len       device int32[n], or NULL when every read has `fixed_len` bases
 *   outputs     device: d_ee double[n], d_ns int32[n], d_pass uint8[n]
 *   counts      host, may be NULL (when non-NULL the call synchronises)
 * Per read i this computes what
 *   bernoulli.calculate_errors_PB(seq, quals, alpha)   ref: moira/bernoullimodule.c:66-114,182-263
 *   + process_data's "ee += Ns" / floor                ref: moira/moira.py:827-831
 *   + write_results' keep/discard predicate            ref: moira/moira.py:911,925-926,949-950
 * compute, with the Python semantics where the C reference has undefined
 * behaviour (first CDF row already above 1-alpha -> ee = 0,
 * ref: moira/moira.py:1611,1629 vs moira/bernoullimodule.c:254).
 * d_ee[i] is the value process_data returns (after +Ns / floor).
 * Asynchronous on the context's stream unless `counts` is given -- with one exception since round 5: a batch that is eligible for
 * the narrow pass (mpb_path_info below: fixed-length or ragged rows of up to 4096 bytes, default table, none of the opt-in flags,
 * >= 262144 reads or MPB_FLAG_NARROW_ROWS) synchronises on the stream inside the call -- once per 64 calls of the same batch shape
 * for the sample that picks the pass, and after every narrow pass for the 4-byte count of the reads it hands to the sorted
 * pipeline (the sub-batch is sized from it).  Callers that queue steps behind each other and must not stall pass
 * MPB_FLAG_NO_NARROW.
 *
 * Read length: up to 65535 bases (row_stride <= 65536; rounds 1-3: 16383).  A read whose DP needs at most 1024 rows --
 * every read of up to 1023 bases, and any longer read with fewer than about a thousand expected errors -- runs in one wave
 * (the running probability vector in registers, 16 rows per lane); one that needs more is run by a workgroup of up to
 * 16 waves, the row that crosses a wave boundary travelling through an LDS stream.  Same arithmetic either way.
 * What a read may NEED is 16384 DP rows (16 waves x 1024: about 16,000 expected errors).  A read of up to 16383 bases can
 * never need more; a longer one that does -- tens of thousands of bases of which a third are wrong -- gets ee = NaN, pass = 0.
 * (The reference's C extension keeps its table on the stack and overruns it near 1000 bases; its Python twin,
 * ref: moira/moira.py:1561-1634, has no limit -- `--error_calc poisson_binomial_py`.)
 *
 * A length in d_len outside 0..min(row_stride, 65535) is NEVER clamped: that read gets ee = NaN, pass = 0, Ns = 0 (so
 * a caller that passes counts == NULL and never synchronises on an error cannot consume a result computed on a
 * different length), a device-side counter is raised, and the next call on this context that fetches `counts` from
 * mpb_filter_device fails with MPB_E_INVALID and clears it.
 */
int mpb_filter_device(mpb_ctx *ctx,
                      const uint8_t *d_q, int64_t n, int64_t row_stride,
                      const int32_t *d_len, int32_t fixed_len,
                      const mpb_filter_params *params,
                      double *d_ee, int32_t *d_ns, uint8_t *d_pass,
                      mpb_filter_counts *counts);

/*
 * Classified at source.  For a batch that is PRODUCED on the device -- raw FASTQ text (ref:
 * moira/moira.py:1177) already in HBM -- the pass that decodes it also classifies it: mpb_decode_classify_device
 * decodes d_seq / d_qual_ascii into the packed matrix d_q_out exactly as mpb_decode_ascii_device does AND, from the
 * bytes it holds in registers anyway, sums the row-budget prediction, counts the ambiguous bases (d_ns) and fills the
 * class histograms.  mpb_filter_device_classified, called next on the same context with the same batch, parameters and
 * result arrays, starts at the histogram scan: the packed matrix is read once (by the DP), not twice, which removes
 * the classification pass of mpb_filter_device (16 % of a 300-bp step).  Results are identical to
 * mpb_decode_ascii_device + mpb_filter_device.  Any other filter call on the context in between invalidates the
 * classification (mpb_filter_device_classified then fails with MPB_E_INVALID; nothing stale is ever consumed).
 * Both are asynchronous on the context's stream (unless `counts` is given).
 * Rows of up to 16384 bytes (the fused pass parks whole rows in LDS); longer text: mpb_decode_ascii_device + mpb_filter_device.
 */
int mpb_decode_classify_device(mpb_ctx *ctx, const uint8_t *d_seq, const uint8_t *d_qual_ascii, int64_t n,
                               int64_t row_stride, const int32_t *d_len, int32_t fixed_len, int32_t fastq_offset,
                               const mpb_filter_params *params, uint8_t *d_q_out,
                               double *d_ee, int32_t *d_ns, uint8_t *d_pass, int32_t *d_err);
int mpb_filter_device_classified(mpb_ctx *ctx,
                                 const uint8_t *d_q, int64_t n, int64_t row_stride,
                                 const int32_t *d_len, int32_t fixed_len,
                                 const mpb_filter_params *params,
                                 double *d_ee, int32_t *d_ns, uint8_t *d_pass,
                                 mpb_filter_counts *counts);

/*
 * Same for a batch in HOST memory; synchronous (results are in the caller's arrays on return).
 * Replaces the per-read Pool.apply_async dispatch + .get() barrier of ref: moira/moira.py:431-454
 * with a chunked, double-buffered pipeline: the batch is cut into chunks of <= 128 MiB of qualities
 * (at least four per batch where it is large enough) and the host-to-device copy of chunk k+1, the kernels
 * of chunk k and the device-to-host copy of chunk k-1 run concurrently on three streams through FOUR
 * pinned/device slots (about 0.55 GiB of device memory, and as much pinned host memory when the input is pageable).
 * Inputs in pinned memory (mpb_host_alloc) are copied by DMA from where they lie.
 * Batches of <= 4096 reads take one launch with one read per wave, the read's DP rows spread over the wave's lanes (what a
 * per-read caller needs is latency; MPB_FLAG_BATCHED_ONLY forces the batched pipeline); up to 1 MiB of input is read by that
 * kernel straight from the library's pinned block (one runtime call, no copies), and up to 256 reads report their completion
 * through a word per read in the same block instead of the runtime's completion signal: 27 us for one read.  Results are
 * identical either way.
 * Lengths are validated, never clamped: a len[i] < 0, > row_stride or > 65535 fails the call with
 * MPB_E_INVALID before anything is computed (mpb_filter_device, whose lengths live on the device, gives such a
 * read ee = NaN, pass = 0 and reports the condition from a device-side counter when `counts` is requested).
 */
int mpb_filter_host(mpb_ctx *ctx,
                    const uint8_t *q, int64_t n, int64_t row_stride,
                    const int32_t *len, int32_t fixed_len,
                    const mpb_filter_params *params,
                    double *ee, int32_t *ns, uint8_t *pass,
                    mpb_filter_counts *counts);

/*
 * The same batch call over SEVERAL contexts (normally one per GPU of the node) from one host process: the batch is
 * cut into n_ctx contiguous, balanced shards in read order, one host thread per context runs mpb_filter_host
 * (poisson == 0) or mpb_filter_poisson_host (poisson != 0) on its shard, and every shard's results land in the
 * caller's arrays at the shard's offset -- gathered in read order, no exchange step, no collective.
 * Replaces `Pool(args.processors)` of ref: moira/moira.py:398-399 for a host-fed caller: n_ctx PCIe links from one
 * process.  A context must not be listed twice.  counts (may be NULL) receives the totals over all shards.
 * On failure the status and message of the first failing shard are returned (other shards may have completed).
 */
/* The split mpb_filter_host_multi uses (and moira_amd/shard.py, for one process per GPU): rank r of `world` owns the
 * contiguous range [*lo, *hi) of n reads; the first n % world ranks get one read more.  Host-only. */
int mpb_shard_bounds(int64_t n, int32_t world, int32_t rank, int64_t *lo, int64_t *hi);
int mpb_filter_host_multi(mpb_ctx *const *ctxs, int32_t n_ctx,
                          const uint8_t *q, int64_t n, int64_t row_stride,
                          const int32_t *len, int32_t fixed_len,
                          const mpb_filter_params *params,
                          double *ee, int32_t *ns, uint8_t *pass,
                          mpb_filter_counts *counts, int32_t poisson);

/*
 * One read, the exact signature-level twin of
 *   bernoulli.calculate_errors_PB(contig, contig_quals, alpha) -> (ee, Ns)
 * ref: moira/bernoullimodule.c:66-114.  Validates alpha in (0,1) (:79-83),
 * clamps Q0->1 (:104-107), counts 'N' and 'n' (:196).  `ee` is the raw
 * percentile (no +Ns, no floor).  Runs the HIP path on a batch of one.
 * Any non-negative int is a valid score, as in the reference (:92-108): a read with scores above 254 -- which the byte
 * matrix of the batch entries cannot hold -- gets its own code table for the call ({1-p, p'} of every such score, by the
 * same expressions, under a byte code the read does not use).  MPB_E_RANGE only for a negative score, or for a read with
 * more than 254 distinct scores of which some exceed 254.
 * A stream of such calls pays no kernel launch each (round 5): while they come, the context keeps the one-read kernel
 * resident -- the call writes the packed read into a mailbox in pinned host memory and waits for the kernel's answer there.
 * That kernel leaves by itself 100 ms after its launch at the latest (so a context that is no longer called holds nothing
 * on the GPU), and before any call on this context frees memory.  Environment MPB_SERVE=0: a launch per call, as before.
 */
int mpb_calculate_errors_PB(mpb_ctx *ctx, const char *contig,
                            const int32_t *contig_quals, int32_t len,
                            double alpha, double *ee, int32_t *ns);

/* ---- batches with scores above 254 (round 4) ---------------------------------------------------------------------- */
/*
 * ref: moira/bernoullimodule.c:92-108 takes any int score; the byte matrix has 254 score codes.  mpb_pack_batch_coded packs a
 * batch (concatenated sequences -- NULL: no ambiguous bases -- and INTEGER scores, off[n + 1] offsets; max_len > 0 truncates,
 * moira/moira.py:806-807) and gives every distinct score above 254 a byte code the batch does not use, from 254 down;
 * code_scores[256] receives what each code stands for (code_scores[c] == c for the untouched ones; entries 0 and 255 are the
 * N / n markers).  MPB_E_RANGE when the batch has more such scores than free codes (split it, or use the per-read entry) or a
 * negative score.  mpb_filter_host_coded is mpb_filter_host on such a matrix: the call runs on a private copy of the
 * {1 - p, p'} table in which code c carries the values of score code_scores[c] (same libm expressions); NULL = the plain
 * entry.  A code should stand for a score >= its own value: the row predictor reads codes as scores, so a code that
 * understates its error probability only costs re-runs, never a result.  Poisson-binomial methods only.
 */
int mpb_pack_batch_coded(const char *seq_cat, const int32_t *qual_cat, const int64_t *off, int64_t n, int32_t max_len,
                         int64_t row_stride, uint8_t *q_out, int32_t *len_out, int32_t *code_scores);
int mpb_filter_host_coded(mpb_ctx *ctx, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len,
                          int32_t fixed_len, const mpb_filter_params *params, const int32_t *code_scores,
                          double *ee, int32_t *ns, uint8_t *pass, mpb_filter_counts *counts);

/* ---- per-read calls from many worker processes: the broker -------------------------------------- */
/*
 * Reference shape: moira/moira.py:398-399,431-454 -- `Pool(args.processors)` worker processes, each calling
 * bernoulli.calculate_errors_PB(contig, contig_quals, alpha) once per read (moira/moira.py:817) and blocking for the
 * answer.  One GPU context per worker makes those one-read launches time-share the card.  Instead ONE process -- the
 * broker -- owns the GPU; the workers never touch it: mpb_broker_call() packs the read (host only, the packer of
 * mpb_calculate_errors_PB) into the caller's slot of a shared-memory segment and waits, the broker gathers whatever is
 * pending into one launch of the one-read-per-wave kernel (several such micro-batches in flight) and hands the results
 * back through the slots -- or (the default since round 5) keeps ONE kernel resident while calls arrive, a wave per slot that
 * polls the slot itself (the segment is registered with the runtime: the worker's row, parameters and door word are read
 * where they are written, the answer lands in the slot; MPB_BROKER_DIRECT=0: a mailbox of the broker's own copies): no
 * launch per call, and no pass through the broker thread; that kernel leaves by itself after 100 ms and is launched again
 * while calls keep coming (environment MPB_BROKER_SERVER=0: the launches).
 * Results are those of mpb_calculate_errors_PB bit for bit, scores above 254 included.
 *
 *   mpb_broker_serve   runs the broker loop on `ctx` in the calling thread until mpb_broker_shutdown(name), or until no
 *                      live process has been attached and nothing has been asked for idle_exit_ms milliseconds
 *                      (0: never).  Creates the segment "/moira_pb_<name>" (letters, digits, '_', '-', '.'), n_slots
 *                      (1..256) slots = the most worker processes it serves at once; replaces a segment a dead broker
 *                      left behind; MPB_E_INVALID when a live broker of that name exists.  Blocking.
 *   mpb_broker_attach  no ctx, no HIP call: maps the segment (waiting up to wait_ms for a broker that is still starting)
 *                      and claims a slot for this process.  An attachment does not survive fork(): the child attaches
 *                      itself (mpb_broker_call refuses an attachment made by another pid).
 *   mpb_broker_call    the per-read entry, arguments and errors of mpb_calculate_errors_PB.  MPB_E_HIP when the broker
 *                      stops or dies while the read is pending (the caller may start a new one and attach again).
 *   mpb_broker_stats   reads served / launches (micro-batches, or generations of the resident kernel) / reads run alone (row budget missed, or scores above
 *                      254), the broker's pid (0: not serving) and the number of attached processes; any may be NULL.
 */
typedef struct mpb_broker_client mpb_broker_client;
int mpb_broker_serve(mpb_ctx *ctx, const char *name, int32_t n_slots, int32_t idle_exit_ms);
int mpb_broker_attach(const char *name, int32_t wait_ms, mpb_broker_client **client_out);
int mpb_broker_call(mpb_broker_client *client, const char *contig, const int32_t *contig_quals, int32_t len,
                    double alpha, double *ee, int32_t *ns);
int mpb_broker_detach(mpb_broker_client *client);
int mpb_broker_shutdown(const char *name);
int mpb_broker_stats(const char *name, int64_t *served, int64_t *batches, int64_t *solo, int32_t *pid, int32_t *attached);

/* NUMA placement (round 5).  Each shard thread of mpb_filter_host_multi restricts itself, before its pipeline starts, to the
 * CPUs of the NUMA node its GPU hangs off -- /sys/bus/pci/devices/<bus id>/numa_node, /sys/devices/system/node/node<k>/cpulist,
 * intersected with what the process may use -- so that the pinned staging blocks it allocates and first touches, and the threads
 * that copy a pageable input into them, are node-local (8 GPUs fed from one socket do not scale).  No NUMA information, or
 * MOIRA_PB_NO_NUMA in the environment: threads stay where they are.  mpb_numa_cpulist_for_pci is that lookup, host-only
 * (sysfs_root "" or NULL = the real tree; a directory holding a copy of the two files for tests): *node_out = the node or -1,
 * cpulist_out = its CPU list as sysfs prints it ("0-47,96-143"). */
int mpb_numa_cpulist_for_pci(const char *sysfs_root, const char *pci_bus_id, int32_t *node_out, char *cpulist_out, int32_t cpulist_len);

/* ---- --error_calc poisson -------------------------------------- */
/*
 * Poisson approximation, ref: moira/moira.py:1637-1679 (calculate_errors_poisson).
 * Device part (a pure streaming reduction, the one variant that is HBM-bound):
 *   d_lambda[i] = sum over non-'N' bases, IN BASE ORDER, of pow(10, q / -10.0) (host-built LUT),
 *   bit-identical to the reference's sequential `Lambda += 10**(qscore / -10.0)`; d_ns[i] = #'N'.
 * NOTE the Python reference treats only upper-case 'N' as ambiguous here (moira.py:1660); a
 * byte 255 ('n') is therefore rejected by this entry point -- pack such a base as a normal one.
 */
int mpb_poisson_lambda_device(mpb_ctx *ctx, const uint8_t *d_q, int64_t n, int64_t row_stride,
                              const int32_t *d_len, int32_t fixed_len,
                              double *d_lambda, int32_t *d_ns);
/*
 * Host tail, exactly the reference's scalar loop with the same libm calls (exp, pow) and the
 * correctly rounded factorials Python's int->float conversion yields: CDF until > 1-alpha,
 * interpolate, then +Ns / floor / predicate as for the Poisson-binomial path.  lambda/ns are host
 * arrays (e.g. copied back from mpb_poisson_lambda_device).  ee is NaN where the reference would
 * raise OverflowError (more than 170 terms or pow overflow).
 */
int mpb_poisson_finish_host(const double *lambda, const int32_t *ns, const int32_t *len,
                            int32_t fixed_len, int64_t n, const mpb_filter_params *params,
                            double *ee, uint8_t *pass);
/* Both steps for a batch in host memory, through the same chunked, overlapped pipeline as mpb_filter_host (the host tail
 * of a chunk runs while the GPU works on the next chunks).  Lengths may be as long as the row (no 1023-base limit here). */
int mpb_filter_poisson_host(mpb_ctx *ctx, const uint8_t *q, int64_t n, int64_t row_stride,
                            const int32_t *len, int32_t fixed_len, const mpb_filter_params *params,
                            double *ee, int32_t *ns, uint8_t *pass, mpb_filter_counts *counts);

/*
 * One read, the signature-level twin of moira.py's
 *   calculate_errors_poisson(sequence, quals, alpha) -> (expected_errors, Ns)
 * ref: moira/moira.py:1637-1679.  The function's own rules, not the batch encoding's: any non-negative int is a score
 * (one outside 1..254 gets a private table entry for the call, as in mpb_calculate_errors_PB), Q0 is p = 1 (the clamp to
 * Q1 happens in process_data, moira.py:814, before the function is called), only 'N' is skipped and counted.  `ee` is
 * the raw percentile, NaN where the Python function raises OverflowError.  alpha must be in (0, 1), as the script's own
 * argument check demands (the bare function also takes 1).
 */
int mpb_calculate_errors_poisson(mpb_ctx *ctx, const char *sequence,
                                 const int32_t *quals, int32_t len,
                                 double alpha, double *ee, int32_t *ns);

/* ---- synthetic workload (BASELINE.json configs; integer-only generator) ---- */
/* Fill a device quality matrix with the counter-based synthetic model of
 * include/mpb_synth.h (identical integers on host and device).
 * fixed_len > 0: every read has fixed_len bases, d_len may be NULL.
 * fixed_len == 0: ragged, lengths drawn in [min_len, max_len] and written to d_len. */
int mpb_synth_fill_device(mpb_ctx *ctx, uint8_t *d_q, int64_t n, int64_t row_stride,
                          int32_t fixed_len, int32_t min_len, int32_t max_len,
                          int32_t *d_len, uint64_t seed, int64_t first_read);
/* The same for a named quality profile of include/mpb_synth.h: 0 = the model above (BASELINE's configs), 1 = a clean run
 * (Q33..Q40, 0.003 % ambiguous bases: the HBM-bound regime, bench.py extras.high_quality_300). */
int mpb_synth_fill_device_profile(mpb_ctx *ctx, uint8_t *d_q, int64_t n, int64_t row_stride,
                                  int32_t fixed_len, int32_t min_len, int32_t max_len,
                                  int32_t *d_len, uint64_t seed, int64_t first_read, int32_t profile);

/* ---- measurement ----------------------------------------------------------- */
/* When enabled, every launch of kernel `MPB_K_*` is bracketed by HIP events on
 * the context's stream; mpb_kernel_time() returns accumulated milliseconds and
 * the number of launches since the last reset (it synchronises the stream). */
int mpb_timing_enable(mpb_ctx *ctx, int on);
int mpb_timing_reset(mpb_ctx *ctx);
int mpb_kernel_time(mpb_ctx *ctx, int kernel_id, double *total_ms, int64_t *launches);
/* Histogram of the DP row-budget classes of the last mpb_filter_device call (or mpb_filter_host call
 * that went through the batched pipeline; the one-read-per-wave path keeps no table):
 * caps[i] = rows (J_cap) of class i, counts[i] = reads in it.  Returns the
 * number of classes written (<= max_classes).  Synchronises. */
int mpb_last_class_histogram(mpb_ctx *ctx, int32_t *caps, int64_t *counts, int32_t max_classes);
/* Row budget (J_cap) the prepass assigned to each of the first n reads of the last mpb_filter_device call
 * (0 for a read MPB_FLAG_DECISION_ONLY settled without a DP); host array of n int32.  Diagnostic: what
 * tools/class_efficiency.py uses to build single-class batches.  Synchronises. */
int mpb_last_read_budgets(mpb_ctx *ctx, int32_t *caps_out, int64_t n);
/* (Both fail with MPB_E_INVALID when the last filter call took the narrow pass: the workspace then describes the sub-batch it handed
 * back, or an earlier batch.  Ask for them after a call with MPB_FLAG_NO_NARROW.) */
/* Algorithmic DP cells of the last mpb_filter_device call made with MPB_FLAG_COUNT_CELLS (0 without it).  Synchronises. */
int mpb_last_algorithmic_cells(mpb_ctx *ctx, int64_t *cells);

/*
 * Which pass the last mpb_filter_device call took (round 5).
 * A batch of GOOD reads -- nearly every read's CDF crosses 1 - alpha within its first 2..4 rows of the table
 * (ref: moira/bernoullimodule.c:152-166,219-251: rows 0..j depend on no later row) --
 * is bound by HBM, not by FP64 issue, and the sorted pipeline would read the matrix twice.  Such a batch (>= 262144 reads,
 * default table, none of the opt-in flags) takes the NARROW PASS instead: the matrix is read once, one read per lane with
 * narrow_rows rows in registers; reads it cannot finish (more rows needed, or a lower-case 'n') are listed and run through the
 * sorted pipeline WHERE THEY LIE (round 6: its classification and sort walk the list, its DP addresses rows and results by read
 * anyway; round 5 gathered them into a dense sub-batch of their own); when they are more than half of the batch the whole batch
 * runs through the sorted pipeline instead and n_fallback reports n.  Fixed-length batches are walked in natural order.  RAGGED batches (round 6: d_len != NULL, rows of up
 * to 4096 bytes -- e.g. the contigs of the reference's paired mode, moira/moira.py:789-801) are first sorted by length inside
 * windows of 4096 consecutive reads (8 bytes per read of workspace), so that the 64 reads a wave walks together end together,
 * and only the 128-byte lines a read's bases lie in are fetched; in the sample's histogram each read then weighs its 16-byte
 * chunks, and a length outside its row is a read the pass hands back (the sorted pipeline reports it).  The choice is
 * made from a sample of at most 0.1 % of the reads (the prepass' row prediction on 256..4096 reads spread over the batch), is
 * reused while the batches of a context keep their shape and parameters (re-sampled when a pass had to hand back more reads
 * than the sample promised, and every 64 calls), and steers speed only: every read's result is the reference's bit for bit
 * whichever pass computed it.
 *   narrow_rows  0: the sorted pipeline; 2..4: the narrow pass with that many rows
 *   sampled      1 when this call drew a sample (sample_hist is then its histogram: [0] reads with a lower-case 'n', [r] reads
 *                that need r rows (r = 1..14), [15] more), 0 when it reused the previous decision
 *   n_fallback   reads the narrow pass handed to the sorted pipeline
 *   narrow_split ragged batches, narrow_rows >= 3 ("mixed rows"): how many rows a read of a given quality needs grows with its length,
 *                and the pass walks its reads sorted by length -- groups whose longest read has at most this many 16-byte chunks
 *                (safely fewer than the shortest sampled read that needed narrow_rows rows) ran with narrow_rows - 1 rows; 0: none
 */
typedef struct mpb_path_info {
    int32_t narrow_rows;
    int32_t sampled;
    int64_t n_fallback;
    int32_t sample_hist[16];
    int32_t narrow_split;   /* round 6, ragged batches with narrow_rows >= 3: groups of the pass whose longest read has at most this
                               many 16-byte chunks ran with narrow_rows - 1 rows (0: none) */
    int32_t reserved_;
} mpb_path_info;
int mpb_last_path(mpb_ctx *ctx, mpb_path_info *out);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* MOIRA_PB_H */
