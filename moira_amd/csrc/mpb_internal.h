// mpb_internal.h -- shared between the C-ABI layer (mpb_api.cpp) and the gfx950 kernels
// (mpb_kernels.hip).  Not installed; the public surface is include/moira_pb.h.
#ifndef MPB_INTERNAL_H
#define MPB_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- DP row-budget classes -------------------------------------------------------------
// A read predicted to need J rows of the DP table goes to the smallest class with
// cap >= J.  cap = G * R: G lanes cooperate on one read, each keeps R consecutive rows of the
// running probability vector in registers.  G == 1 is one read per lane.  R <= 16 everywhere so
// the whole kernel fits a small VGPR budget (many waves per SIMD hide the LUT-read latency).
// Tile classes (below) cover up to MPB_TILE_MAX_ROWS rows of the DP table: one wave, 64 lanes x 16 registers.
// A read predicted to need more rows is a WIDE read: one workgroup of up to MPB_WIDE_WAVES waves holds its running
// vector (wave w keeps rows w*1024 .. w*1024+1023), the waves run one 64-base block apart and hand the row that
// crosses a wave boundary on through an LDS stream (k_wide).  16 waves x 1024 rows = 16384 rows: what a read may NEED (about
// 16,000 expected errors); a read that needs more has no result (NaN, pass = 0).  Rounds 1-3 tied the longest read to that
// number (len + 1 <= rows, so that the fallback always covers); round 4: the kernels never cared about the length itself, a
// long read of moderate quality needs few rows, so the longest read is what the 16-bit counters of ambiguous bases hold.
#define MPB_TILE_MAX_ROWS 1024
#define MPB_WIDE_WAVES 16
#define MPB_MAX_ROWS (MPB_TILE_MAX_ROWS * MPB_WIDE_WAVES)
#define MPB_MAX_LEN 65535                 // longest read (perm_ns and the prepass' marker counts are 16 bits wide)
#define MPB_MAX_STRIDE 65536              // widest quality-matrix row (bytes)
#define MPB_SMALL_MAX_STRIDE 16384        // widest row of the one-read-per-wave kernel (a lane peels at most 256 bytes of markers)

struct MpbClass { int cap, G, R; };

// X(id, R, G): the single list the class table AND the kernel's dispatch switch are generated from
#define MPB_NCLS 32
#define MPB_CLASSES(X)                                                                       \
    X(0, 2, 1) X(1, 3, 1) X(2, 4, 1) X(3, 5, 1) X(4, 6, 1) X(5, 7, 1) X(6, 8, 1) X(7, 9, 1)  \
    X(8, 10, 1) X(9, 12, 1) X(10, 14, 1) X(11, 16, 1)                                        \
    X(12, 10, 2) X(13, 12, 2) X(14, 14, 2) X(15, 16, 2)                                      \
    X(16, 10, 4) X(17, 12, 4) X(18, 14, 4) X(19, 16, 4)                                      \
    X(20, 9, 8) X(21, 10, 8) X(22, 11, 8) X(23, 12, 8) X(24, 14, 8) X(25, 16, 8)             \
    X(26, 10, 16) X(27, 12, 16) X(28, 16, 16)                                                \
    X(29, 12, 32) X(30, 16, 32) X(31, 16, 64)

// Latency bodies of the one-read-per-wave kernel (k_small): X(id, R, G) with cap = R * G = 2^(id + 1); a read of `rows` rows takes
// id = ceil(log2(rows)) - 1.  ONE row per lane up to 64 rows (7 instructions of the wave's dependent chain per base), then 2, 4,
// 8, 16 rows per lane.  (16, 64) is a tile class as well; the others exist only here.
#define MPB_THIN_CLASSES(X)                                                                  \
    X(0, 1, 2) X(1, 1, 4) X(2, 1, 8) X(3, 1, 16) X(4, 1, 32) X(5, 1, 64) X(6, 2, 64)         \
    X(7, 4, 64) X(8, 8, 64) X(9, 16, 64)

#define MPB_CLASS_ENTRY(ID, RR, GG) {(RR) * (GG), GG, RR},
#define MPB_CLASS_TABLE { MPB_CLASSES(MPB_CLASS_ENTRY) }

// reads handled by one prepass / scatter block: MPB_PRE_ROUNDS rounds of 256 (one thread per read in
// the ranking step).  More reads per block = shorter histograms to scan.
#define MPB_PRE_ROUNDS 4
#define MPB_PRE_READS (256 * MPB_PRE_ROUNDS)
// class byte of a read the prepass already settled (MPB_FLAG_DECISION_ONLY): never scattered, never run
#define MPB_CLS_SETTLED 0x7f
// class byte of a wide read (more than MPB_TILE_MAX_ROWS rows predicted): listed for k_wide, never scattered
#define MPB_CLS_WIDE 0x7e

// Sort key of a read = (class, length bin of 2^len_shift bases): perm[] is grouped by class and, inside a class,
// by length, because a DP tile runs as long as its longest read.  Batches with one fixed length use a
// single bin.  16 bins: 64 bases wide for rows of up to 1024 bases, wider (MpbDevParams.len_shift) for longer rows.
#define MPB_LEN_SHIFT 6               // narrowest bin width: 2^6 bases
#define MPB_LEN_BINS (1024 >> MPB_LEN_SHIFT)
#define MPB_SKEYS (MPB_NCLS * MPB_LEN_BINS)

// Device-side tables produced by the scan kernel, consumed by scatter and DP kernels.
struct MpbTables {
    int32_t count[MPB_NCLS];          // reads per class
    int32_t perm_base[MPB_NCLS + 1];  // first slot of the class in perm[]
    int32_t tile_start[MPB_NCLS + 1]; // first tile of the class (tiles ordered widest class first)
    int32_t total_tiles;
    int32_t kcount[MPB_SKEYS];        // reads per sort key (class * bins + bin)
    int32_t key_base[MPB_SKEYS];      // first slot of the key in perm[]
};

// consecutive tiles one wave processes: almost always of one class, so a wave enters a class body
// (a real call that saves callee-saved VGPRs to scratch) once per chunk
#define MPB_DP_CHUNK 8

struct MpbDevParams {
    double thr;            // 1 - alpha, computed on the host in double (ref: bernoullimodule.c:244)
    double uncert;
    double maxerrors;      // NaN when unset
    float  z;              // Phi^-1(1 - alpha), prediction only
    float  zq;             // (z*z - 1) / 6,     prediction only
    float  clow;           // sqrt(2 ln(1/(1-alpha))): Chernoff lower-tail coefficient (MPB_FLAG_DECISION_ONLY)
    int32_t ambig_mode;
    uint32_t flags;
    int32_t fixed_len;     // used when d_len == nullptr
    int32_t max_len;       // upper bound of every length in the batch (<= row_stride)
    int32_t len_shift;     // length-bin width of the sort key: 2^len_shift bases, (max_len - 1) >> len_shift < MPB_LEN_BINS
};

struct MpbWorkspace {
    uint8_t  *cls;         // [n]   class id | 0x80 if the read has an upper-case N
    int32_t  *perm;        // [n + MPB_NCLS*64] read indices grouped by class
    uint16_t *perm_ns;     // [n + MPB_NCLS*64] ambiguity count of perm[k]'s read, so that the DP epilogue reads it
                           // next to the index instead of gathering ns[idx] (one line request per read)
    int32_t  *blockhist;   // [MPB_SKEYS][nblocks_pre], key-major
    MpbTables *tables;     // main pass
    MpbTables *tables2;    // overflow pass
    int32_t  *ovf_list;    // [n]
    int32_t  *ovf_count;   // [1]
    int32_t  *bad_len;     // [1] lengths in d_len outside 0..max_len seen by the prepass (sticky; the host reports and clears it)
    long long *ovf_total;  // [1] overflow re-runs summed over the chunks of one host-pipeline call
    int32_t  *wide_list;   // [n] reads the prepass classed as wide (only allocated for batches whose rows can hold > 1023 bases)
    int32_t  *wide_rows;   // [n] ... and the rows predicted for each
    int32_t  *wide_count;  // [1]
    unsigned long long *pass_count;  // [1]
    unsigned long long *alg_cells;   // [1] MPB_FLAG_COUNT_CELLS: algorithmic DP cells of the last mpb_filter_device call
    const double2 *lut;    // [256] {1-p, p'} on device
    int32_t  *nar_count;   // [1] reads the natural-order narrow pass (k_narrow) could not finish
    int32_t  *nar_seg;     // [n + 64] ... as the waves of k_narrow listed them, a segment per wave
    int32_t  *nar_list;    // [n + 64] ... compacted: the dense list the sub-batch is gathered by
    int32_t  *nar_wave_count, *nar_wave_off;   // [MPB_NAR_MAX_WAVES] entries of each wave's segment / where it goes in nar_list
    int32_t  *nar_sample;  // [MPB_NAR_BUCKETS + 2] histogram of the batch sample that picks the pass (k_sample) + the chunks of the
                           // shortest sampled reads that need a third / a fourth row (ragged batches)
    // the narrow pass of RAGGED batches (k_narrow_rg, round 6)
    int2     *rg_ord;      // [n + 64] {read, length} in the order the pass walks them: windows of 4096 reads sorted by length
    int32_t  *rg_gpre;     // [n / 64 + 2] cost of each group of 64 entries (from its longest read), summed up inside its window
    unsigned long long *rg_wsum, *rg_wpre;   // [n / 4096 + 2] cost of each window; exclusive prefix ([nwin] = the total)
    int32_t  *rg_gstart;   // [MPB_NAR_MAX_WAVES + 1] first group of each wave of the persistent grid (written by the pass itself)
};

// ---- natural-order narrow pass (round 5) -------------------------------------------------------
// A batch of GOOD reads (nearly every read needs <= R0 rows of the DP table, R0 = 2..4) is bound by HBM, not by FP64 issue, and
// the two-pass pipeline (classify, sort, DP in class order) then moves 2.7x the algorithmic bytes.  k_narrow<R0> instead walks
// the matrix ONCE, in natural order, one read per lane with R0 rows in registers: rows 0..R0-1 of the table are exact whatever
// the read turns out to need, so every read whose CDF crosses inside them is finished bit for bit; the others -- and every read
// with a lower-case 'n', whose table entry is a NaN in this pass -- go on a list, are gathered into a dense sub-batch and run
// through the ordinary pipeline.  The choice is made per batch from a sample of <= 0.1 % of the reads.
// Two forms of the same pass: k_narrow_rs (row strides that are a multiple of 64 bytes: a lane walks one or two rows as one stream
// of whole 128-byte lines, panels staged in registers) and k_narrow (any stride: 64-byte panels through an LDS-DMA ring).
#define MPB_NAR_MIN_ROWS 2
#define MPB_NAR_MAX_ROWS 4
#define MPB_NAR_BUCKETS 16                // k_sample: [0] reads with a lower-case 'n', [r] reads that need r rows (r = 1..14), [15] more
#define MPB_NAR_MAX_WAVES 8192            // waves of the persistent k_narrow grid (256 CUs x 3 workgroups x 4 waves = 3072 on MI355X)
#define MPB_NAR_AUTO_MIN_READS (1 << 18)  // below this a batch always takes the ordinary pipeline (unless the path is forced)

#define MPB_LUT_BYTES   (256 * 16)
#define MPB_LAMBDA_MAX_STRIDE (1 << 24)     // k_lambda addresses the 64 rows of a wave with 32-bit byte offsets

// Launch wrappers (mpb_kernels.hip).  All asynchronous on `s`.
// list != nullptr (round 6): the pass (and the scatter below) runs over the n reads list[0 .. n) of the matrix, not its first n rows
void mpb_launch_prepass(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                        const MpbDevParams &prm, const MpbWorkspace &ws, int32_t *ns_out,
                        double *ee_out, uint8_t *pass_out, hipStream_t s, const int32_t *list = nullptr);
void mpb_launch_decode_classify(const uint8_t *seq, const uint8_t *qual, int32_t offset, uint8_t *out, int32_t *err,
                                int64_t n, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                                const MpbWorkspace &ws, int32_t *ns_out, double *ee_out, uint8_t *pass_out, hipStream_t s);
void mpb_launch_encode(const uint8_t *q, int64_t n, int64_t stride, int32_t offset, uint8_t *seq, uint8_t *qual, hipStream_t s);
// k_small on inputs / outputs that live in pinned HOST memory (one runtime call per launch: no copies): the device scratch and
// the completion flags that go with it
struct MpbSmallHost {
    uint8_t *stage;        // n x stride bytes of device memory: the rows are parked here by the kernel's statistics pass
    int32_t *ns_dev;       // n ints of device memory: the ambiguity counts for the class body (the report goes to `ns`)
    uint32_t *done;        // n words of pinned host memory, or nullptr: done[i] = token once read i's results are visible
    uint32_t token;
};
void mpb_launch_small(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                      const MpbWorkspace &ws, int32_t *ns, double *ee, uint8_t *pass, hipStream_t s,
                      const MpbSmallHost *host = nullptr);
// The resident one-read server (k_serve; the broker's calls without a launch each): n_ent mailbox entries, one wave each.
// Everything the host and the device exchange lies in pinned HOST memory (mapped); the scratch the class body walks in HBM.
#define MPB_SERVE_STRIDE 2048            // row bytes of a mailbox entry: reads of up to 2047 bases (longer ones take the other paths)
struct alignas(64) MpbServePrm { MpbDevParams p; };      // one 64-byte line per request
struct MpbServeBox {
    // pinned (or registered) host memory: entry e of each array lies e * its byte stride behind the pointer -- dense arrays
    // (the stride is the element's size: the context's own entry, the broker's copies) or fields of the broker's
    // shared-memory slots themselves (every stride is the slot size: the workers' requests are served where they lie)
    const uint8_t *q;                    // the row (written by the host before the door word)
    int64_t stride;                      // bytes a row may hold (the length is clamped to it)
    const MpbServePrm *prm;              // the request's parameters (fixed_len / max_len are set by the wave)
    const unsigned long long *door;      // {length << 32 | token}: a token that differs from done[e] is a request
    uint32_t *done;                      // the token of the last request served
    double *ee; int32_t *ns; uint8_t *pass;      // its results (pass == 2: the row budget was missed, the host runs it alone)
    int64_t q_step, prm_step, door_step, done_step, ee_step, ns_step, pass_step;   // the byte strides of the seven
    const uint32_t *stop;                // [1] non-zero: every wave leaves
    uint32_t *exited;                    // [1] the generation of the last launch that has drained
    // device memory
    uint8_t *stage;                      // [n_ent x stride] the row, parked by the statistics pass
    int32_t *ns_dev; uint8_t *cls; int32_t *ident;   // [n_ent] what the class body reads back
    uint32_t *gone;                      // [1] waves that have left (zeroed before the launch)
    int32_t n_ent;
};
void mpb_launch_serve(const MpbServeBox &box, const double2 *lut, uint32_t generation, uint32_t lifetime_ms, hipStream_t s);
void mpb_launch_scan(int64_t n, const int32_t *len, const MpbWorkspace &ws, hipStream_t s);
void mpb_launch_scatter(int64_t n, const int32_t *len, const int32_t *ns, const MpbDevParams &prm, const MpbWorkspace &ws,
                        hipStream_t s, const int32_t *list = nullptr);
void mpb_launch_dp(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                   const MpbDevParams &prm, const MpbWorkspace &ws, const int32_t *ns,
                   double *ee, uint8_t *pass, hipStream_t s);
void mpb_launch_overflow(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                         const MpbDevParams &prm, const MpbWorkspace &ws, const int32_t *ns,
                         double *ee, uint8_t *pass, hipStream_t s);
// wide reads of the main pass (ws.wide_list); only launched when prm.max_len + 1 > MPB_TILE_MAX_ROWS
void mpb_launch_wide(const uint8_t *q, int64_t stride, const int32_t *len, const MpbDevParams &prm,
                     const MpbWorkspace &ws, const int32_t *ns, double *ee, uint8_t *pass, hipStream_t s);
void mpb_launch_lambda(const uint8_t *q, int64_t n, int64_t stride, const int32_t *len, int32_t fixed_len,
                       const double2 *lut_ap, double *lambda, int32_t *ns, int32_t *bad, hipStream_t s);
void mpb_launch_decode(const uint8_t *seq, const uint8_t *qual, int64_t n, int64_t stride, const int32_t *len,
                       int32_t fixed_len, int32_t offset, uint8_t *out, int32_t *err, hipStream_t s);
void mpb_launch_count(const uint8_t *pass, int64_t n, const MpbWorkspace &ws, hipStream_t s);
// the natural-order narrow pass (fixed-length batches, the context's default table): rows0 in MPB_NAR_MIN_ROWS..MPB_NAR_MAX_ROWS.
// Finished reads get ee / ns (= 0) / pass; the others end up in `list` (dense, in wave order; their number in ws.nar_count).
void mpb_launch_narrow(int rows0, const uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, const MpbDevParams &prm,
                       const MpbWorkspace &ws, double *ee, int32_t *ns, uint8_t *pass, int32_t *list, int grid_blocks, hipStream_t s);
int mpb_narrow_lds_bytes();               // static LDS of one k_narrow workgroup (the host sizes the persistent grid from it)
int mpb_narrow_rs_reads_per_lane(int64_t stride, int rows0);   // k_narrow_rs (whole-line panels staged in registers): reads per lane, 0 = k_narrow
int mpb_narrow_rs_lds_bytes();
// predicted row budgets of `n_sample` reads spread over the batch -> ws.nar_sample (zeroed here)
void mpb_launch_sample(const uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, const int32_t *len, const MpbDevParams &prm,
                       const MpbWorkspace &ws, int n_sample, hipStream_t s);
// the narrow pass of a RAGGED batch (k_rag_sort, k_rag_scan, k_narrow_rg): rows of up to MPB_RG_MAX_STRIDE bytes
#define MPB_RG_MAX_STRIDE 4096
// split_chunks > 0 (rows0 >= 3): groups whose longest read has at most that many 16-byte chunks run with rows0 - 1 rows
void mpb_launch_narrow_ragged(int rows0, int split_chunks, const uint8_t *q, int64_t n, int64_t stride, const int32_t *len,
                              const MpbDevParams &prm, const MpbWorkspace &ws, double *ee, int32_t *ns, uint8_t *pass, int32_t *list,
                              int grid_blocks, hipStream_t s);
void mpb_launch_synth(uint8_t *q, int64_t n, int64_t stride, int32_t fixed_len, int32_t min_len,
                      int32_t max_len, int32_t *len, uint64_t seed, int64_t first_read,
                      hipStream_t s, int32_t profile = 0);

#endif
