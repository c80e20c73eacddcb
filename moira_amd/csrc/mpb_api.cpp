// mpb_api.cpp -- the C ABI of libmoira_pb.so (declared in include/moira_pb.h).
//
// Host side of the path: context (device, stream, LUT, workspace), argument validation with
// the reference's error behaviour (moira/bernoullimodule.c:79-90), packing
// (moira/bernoullimodule.c:92-108,196), and the launch sequence of mpb_kernels.hip.
// There is no CPU compute path in this file: every result comes from the HIP kernels.

#include "../../include/moira_pb.h"
#include "mpb_internal.h"
#include "mpb_host_internal.h"

#include <sched.h>
#include <atomic>
#include <cctype>
#include <chrono>
#include <cmath>
#include <unistd.h>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <thread>
#include <unordered_map>
#include <vector>

#define MPB_VERSION_STR "moira_pb 0.6.0 (gfx950)"

static thread_local char g_err[512] = "";

static int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

#define HIPCHK(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(e_ == hipErrorOutOfMemory ? MPB_E_NOMEM : MPB_E_HIP, "%s failed: %s", \
                        #expr, hipGetErrorString(e_));                                        \
    } while (0)

struct TimedSpan { int kid; hipEvent_t a, b; };

// One of the MPB_HOST_SLOTS chunk buffers of the host pipeline (mpb_filter_host): a device block
// (q | len | ee | ns | pass), a pinned staging block for inputs that arrive in pageable memory, a pinned
// block the outputs land in, and the three events that hand the chunk from stream to stream.
#define MPB_HOST_SLOTS 4
struct HostSlot {
    void *dev = nullptr;      int64_t dev_cap = 0;
    void *pin_in = nullptr;   int64_t pin_in_cap = 0;
    void *pin_out = nullptr;  int64_t pin_out_cap = 0;
    hipEvent_t h2d_done = nullptr, k_done = nullptr, d2h_done = nullptr;
    int64_t off = -1, m = 0;  // chunk in flight (off < 0: none)
};

struct mpb_ctx {
    int device = -1;
    hipStream_t stream = nullptr;        // kernels (and every call that is not the host pipeline)
    hipStream_t copy_stream = nullptr;   // host pipeline: H2D of the next chunk
    hipStream_t out_stream = nullptr;    // host pipeline: D2H of the previous chunk
    HostSlot slot[MPB_HOST_SLOTS];
    int copy_threads = 1;
    double2 *d_lut = nullptr;
    double2 *d_lut_private = nullptr;    // one call's table when a read carries qualities above 254 (mpb_calculate_errors_PB)
    // workspace, grown on demand
    int64_t ws_cap = 0;
    MpbWorkspace ws{};
    void *ws_block = nullptr;
    void *ws_small = nullptr;
    // what the last classify-at-source call produced (consumed by mpb_filter_device_classified; any other call that
    // rebuilds the workspace invalidates it)
    struct Classified {
        bool valid = false; const uint8_t *q = nullptr; int64_t n = 0, stride = 0; const int32_t *len = nullptr;
        int32_t fixed_len = 0; mpb_filter_params params{}; double *ee = nullptr; int32_t *ns = nullptr; uint8_t *pass = nullptr;
    } classified;
    void *ws_wide = nullptr;             // wide-read list + predicted rows, only for batches whose rows hold > 1023 bases
    int64_t ws_wide_cap = 0;
    // timing
    bool timing = false;
    std::vector<TimedSpan> spans;
    std::vector<hipEvent_t> event_pool;
    double acc_ms[MPB_K_COUNT] = {0};
    int64_t acc_n[MPB_K_COUNT] = {0};
    // single-read scratch (device) + host staging for filter_host
    void *one_dev = nullptr;
    int64_t one_cap = 0;
    void *stage_dev = nullptr;
    int64_t stage_cap = 0;
    // pinned host scratch of the small-batch path (inputs and outputs of one call, back to back)
    void *pin_host = nullptr;
    int64_t pin_cap = 0;
    uint32_t small_token = 0;            // completion token of the last zero-copy k_small launch (never 0)
    // ---- natural-order narrow pass (round 5) ----
    int n_cu = 0;                        // compute units of the device (the persistent grid of k_narrow)
    bool narrow_ok = false;              // the default table satisfies a == 1 - b for every score (checked by mpb_create)
    void *ws_nar = nullptr;              // wave segments + dense list + per-wave counts / offsets
    int64_t ws_nar_cap = 0;
    int32_t *pin_words = nullptr;        // pinned host words: [0] list count, [16..32) the sample histogram
    struct NarrowChoice {                // the last decision, reused while the batches keep their shape (it steers speed only)
        bool valid = false; int64_t n = 0, stride = 0; int32_t fixed_len = 0; double alpha = 0; uint32_t flags = 0;
        int rows0 = 0; int split = 0; int calls = 0;
        double expect_back = 0;          // share of the sample (by weight) that needs more rows than rows0 or holds an 'n'
    } nar_choice;
    mpb_path_info last_path{};
    // ---- the per-read entry's resident server (round 5; k_serve with ONE mailbox entry, see serve_one) ----
    struct CtxServe {
        bool tried = false, ok = false, running = false;
        hipStream_t stream = nullptr;
        char *pin = nullptr, *dev = nullptr;
        MpbServeBox box{};
        uint32_t generation = 0, tok = 0;
        double cached_alpha = -1.0;
        MpbDevParams cached_prm;
    } serve;
};

extern "C" {                              // (defined inside the extern "C" block below, next to the entry that uses them)
static void serve_quiesce(mpb_ctx *c);    // the per-read entry's resident kernel leaves (before anything is freed: the runtime
static void serve_free(mpb_ctx *c);       // waits for the whole device there)
}

// Threads that copy a pageable input chunk into its pinned staging block: half of the CPUs this process is
// granted (cgroup quota, else the affinity mask), at most 8 -- one memcpy stream does not saturate PCIe 5.
static int staging_threads()
{
    int n = (int)std::thread::hardware_concurrency();
    if (n < 1) n = 1;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char quota[64]; long long period = 0;
        if (fscanf(f, "%63s %lld", quota, &period) == 2 && strcmp(quota, "max") != 0 && period > 0) {
            const int q = (int)((atoll(quota) + period / 2) / period);
            if (q >= 1 && q < n) n = q;
        }
        fclose(f);
    }
    n = n / 2;
    return n < 1 ? 1 : (n > 8 ? 8 : n);
}

static void parallel_copy(void *dst, const void *src, size_t bytes, int threads)
{
    if (threads <= 1 || bytes < (size_t)(8u << 20)) { memcpy(dst, src, bytes); return; }
    const size_t per = ((bytes + (size_t)threads - 1) / (size_t)threads + 4095) & ~(size_t)4095;
    std::vector<std::thread> th;
    size_t done_to = per < bytes ? per : bytes;      // [0, done_to) is this thread's part; helpers take slices above it
    try {
        th.reserve((size_t)threads);
        for (int t = 1; t < threads; t++) {
            const size_t lo = per * (size_t)t;
            if (lo >= bytes) break;
            const size_t len = bytes - lo < per ? bytes - lo : per;
            th.emplace_back([=] { memcpy((char *)dst + lo, (const char *)src + lo, len); });
            done_to = lo + len;
        }
    } catch (...) {
        // no thread to be had (a container's limit): nothing crosses the C ABI as an exception -- what the helpers
        // did not take is copied here
    }
    memcpy(dst, src, bytes < per ? bytes : per);
    for (auto &t : th) t.join();
    if (done_to < bytes) memcpy((char *)dst + done_to, (const char *)src + done_to, bytes - done_to);
}

extern "C" {

const char *mpb_version(void) { return MPB_VERSION_STR; }
const char *mpb_last_error(void) { return g_err; }

int mpb_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// {1-p, p'} exactly as the reference evaluates them (ref: moira/bernoullimodule.c:202,140-145):
//   p  = pow(10, q / -10.0)
//   a  = prob_j_errors(p, 0, 1) = pow(1 - p, 1)
//   b  = prob_j_errors(p, 1, 1) = ((1-1+1)/(1.0*1)) * (p/(1-p)) * pow(1 - p, 1)
// evaluated on the host with libm (this TU is built with -ffp-contract=off).  Bytes 0 ('N')
// and 255 ('n') are the identity step of the DP: skipping a base == multiplying by {1, 0}.
static void lut_entry(int q, double2 *e)          // any quality the reference's int can hold (p underflows to 0 near Q = 3240)
{
    volatile double p = pow(10, (q / -10.0));
    volatile double a = pow((1 - p), 1);
    volatile double r = p / (1 - p);
    volatile double b1 = ((1 - 1 + 1) / (1.0 * 1)) * r;
    volatile double b = b1 * a;
    e->x = a;
    e->y = b;
}

static void build_lut(double2 *lut)
{
    for (int q = 0; q < 256; q++) {
        if (q == 0 || q == 255) { lut[q].x = 1.0; lut[q].y = 0.0; continue; }
        lut_entry(q, &lut[q]);
    }
}

int mpb_host_lut(double *a_out, double *b_out)
{
    if (!a_out || !b_out) return fail(MPB_E_INVALID, "mpb_host_lut: NULL output");
    double2 h[256];
    build_lut(h);
    for (int q = 0; q < 256; q++) { a_out[q] = h[q].x; b_out[q] = h[q].y; }
    return MPB_OK;
}

int mpb_create(int device_id, mpb_ctx **out)
{
    if (!out) return fail(MPB_E_INVALID, "mpb_create: out is NULL");
    *out = nullptr;
    int n = mpb_device_count();
    if (n <= 0) return fail(MPB_E_NODEVICE, "no HIP device visible (this library has no CPU path)");
    if (device_id < 0 || device_id >= n)
        return fail(MPB_E_NODEVICE, "device %d out of range (0..%d)", device_id, n - 1);
    HIPCHK(hipSetDevice(device_id));
    mpb_ctx *c = new (std::nothrow) mpb_ctx();
    if (!c) return fail(MPB_E_NOMEM, "host allocation failed");
    c->device = device_id;
    hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&c->out_stream, hipStreamNonBlocking);
    for (int k = 0; k < MPB_HOST_SLOTS && e == hipSuccess; k++) {
        e = hipEventCreateWithFlags(&c->slot[k].h2d_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[k].k_done, hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&c->slot[k].d2h_done, hipEventDisableTiming);
    }
    c->copy_threads = staging_threads();
    if (e == hipSuccess) e = hipMalloc((void **)&c->d_lut, 256 * sizeof(double2));
    if (e == hipSuccess) {
        double2 h[256];
        build_lut(h);
        // the narrow pass keeps {p'} alone and recomputes 1 - p' on the device: only sound if the table's a IS that difference
        // (it is: p' == p bit for bit for every encodable score, tests/test_oracle_golden.py::test_lut_pins)
        c->narrow_ok = true;
        for (int qq = 1; qq < 255; qq++) {
            volatile double d = 1.0 - h[qq].y;
            if (memcmp((const void *)&d, &h[qq].x, sizeof(double)) != 0) c->narrow_ok = false;
        }
        e = hipMemcpyAsync(c->d_lut, h, sizeof(h), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);   // h is on this frame
    }
    if (e == hipSuccess) e = hipHostMalloc((void **)&c->pin_words, 64 * sizeof(int32_t), hipHostMallocDefault);
    if (e == hipSuccess) {
        hipDeviceProp_t prop;
        e = hipGetDeviceProperties(&prop, device_id);
        if (e == hipSuccess) c->n_cu = prop.multiProcessorCount;
    }
    if (e != hipSuccess) {
        int rc = fail(MPB_E_HIP, "context setup failed: %s", hipGetErrorString(e));
        mpb_destroy(c);
        return rc;
    }
    *out = c;
    return MPB_OK;
}

int mpb_destroy(mpb_ctx *c)
{
    if (!c) return MPB_OK;
    (void)hipSetDevice(c->device);
    serve_free(c);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    if (c->copy_stream) (void)hipStreamSynchronize(c->copy_stream);
    if (c->out_stream) (void)hipStreamSynchronize(c->out_stream);
    for (auto &sl : c->slot) {
        if (sl.dev) (void)hipFree(sl.dev);
        if (sl.pin_in) (void)hipHostFree(sl.pin_in);
        if (sl.pin_out) (void)hipHostFree(sl.pin_out);
        if (sl.h2d_done) (void)hipEventDestroy(sl.h2d_done);
        if (sl.k_done) (void)hipEventDestroy(sl.k_done);
        if (sl.d2h_done) (void)hipEventDestroy(sl.d2h_done);
    }
    for (auto &s : c->spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
    for (auto ev : c->event_pool) (void)hipEventDestroy(ev);
    if (c->ws_block) (void)hipFree(c->ws_block);
    if (c->ws_small) (void)hipFree(c->ws_small);
    if (c->ws_wide) (void)hipFree(c->ws_wide);
    if (c->one_dev) (void)hipFree(c->one_dev);
    if (c->stage_dev) (void)hipFree(c->stage_dev);
    if (c->pin_host) (void)hipHostFree(c->pin_host);
    if (c->pin_words) (void)hipHostFree(c->pin_words);
    if (c->ws_nar) (void)hipFree(c->ws_nar);
    if (c->d_lut) (void)hipFree(c->d_lut);
    if (c->d_lut_private) (void)hipFree(c->d_lut_private);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->out_stream) (void)hipStreamDestroy(c->out_stream);
    delete c;
    return MPB_OK;
}

#define CTXCHK(c)                                                     \
    do {                                                              \
        if (!(c)) return fail(MPB_E_INVALID, "%s: ctx is NULL", __func__); \
        HIPCHK(hipSetDevice((c)->device));                            \
    } while (0)

int mpb_device_lut(mpb_ctx *c, double *a_out, double *b_out)
{
    CTXCHK(c);
    if (!a_out || !b_out) return fail(MPB_E_INVALID, "mpb_device_lut: NULL output");
    double2 h[256];
    HIPCHK(hipMemcpyAsync(h, c->d_lut, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int q = 0; q < 256; q++) { a_out[q] = h[q].x; b_out[q] = h[q].y; }
    return MPB_OK;
}

int mpb_stream(mpb_ctx *c, void **stream_out)
{
    CTXCHK(c);
    if (!stream_out) return fail(MPB_E_INVALID, "stream_out is NULL");
    *stream_out = (void *)c->stream;
    return MPB_OK;
}

int mpb_synchronize(mpb_ctx *c)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    return MPB_OK;
}

int mpb_malloc(mpb_ctx *c, int64_t bytes, void **dptr_out)
{
    CTXCHK(c);
    if (!dptr_out || bytes < 0) return fail(MPB_E_INVALID, "mpb_malloc: bad arguments");
    *dptr_out = nullptr;
    HIPCHK(hipMalloc(dptr_out, bytes > 0 ? (size_t)bytes : 16));
    return MPB_OK;
}

int mpb_free(mpb_ctx *c, void *dptr)
{
    CTXCHK(c);
    HIPCHK(hipStreamSynchronize(c->stream));
    serve_quiesce(c);
    if (dptr) HIPCHK(hipFree(dptr));
    return MPB_OK;
}

int mpb_memcpy_h2d(mpb_ctx *c, void *dst, const void *src, int64_t bytes)
{
    CTXCHK(c);
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(MPB_E_INVALID, "mpb_memcpy_h2d: bad arguments");
    if (bytes == 0) return MPB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MPB_OK;
}

int mpb_memcpy_d2h(mpb_ctx *c, void *dst, const void *src, int64_t bytes)
{
    CTXCHK(c);
    if (bytes < 0 || (bytes > 0 && (!dst || !src))) return fail(MPB_E_INVALID, "mpb_memcpy_d2h: bad arguments");
    if (bytes == 0) return MPB_OK;
    HIPCHK(hipMemcpyAsync(dst, src, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return MPB_OK;
}

int mpb_memset(mpb_ctx *c, void *dst, int value, int64_t bytes)
{
    CTXCHK(c);
    if (bytes < 0 || (bytes > 0 && !dst)) return fail(MPB_E_INVALID, "mpb_memset: bad arguments");
    if (bytes == 0) return MPB_OK;
    HIPCHK(hipMemsetAsync(dst, value, (size_t)bytes, c->stream));
    return MPB_OK;
}

// ---- packing ---------------------------------------------------------------------------------

static inline uint8_t pack_one(char base, int q)
{
    if (q == 0) q = 1;                       // ref: bernoullimodule.c:104-107
    if (base == 'N') return 0;               // ref: bernoullimodule.c:196 (78)
    if (base == 'n') return 255;             // ref: bernoullimodule.c:196 (110)
    return (uint8_t)q;
}

int mpb_pack_read(const char *seq, const int32_t *quals, int32_t len, uint8_t *row_out, int32_t row_bytes)
{
    if (len < 0 || !row_out || (len > 0 && !quals)) return fail(MPB_E_INVALID, "mpb_pack_read: bad arguments");
    if (len > row_bytes) return fail(MPB_E_INVALID, "read of %d bases does not fit a %d-byte row", len, row_bytes);
    for (int i = 0; i < len; i++) {
        const int q = quals[i];
        if (q < 0) return fail(MPB_E_RANGE, "Qualities must have positive values.");
        if (q > 254) return fail(MPB_E_RANGE, "quality %d at base %d exceeds the encodable maximum 254", q, i);
        row_out[i] = pack_one(seq ? seq[i] : 'A', q);
    }
    memset(row_out + len, 0, (size_t)(row_bytes - len));
    return MPB_OK;
}

int mpb_pack_read_ascii(const char *seq, const char *qual_ascii, int32_t len, int32_t offset,
                        uint8_t *row_out, int32_t row_bytes)
{
    if (len < 0 || !row_out || (len > 0 && !qual_ascii)) return fail(MPB_E_INVALID, "mpb_pack_read_ascii: bad arguments");
    if (len > row_bytes) return fail(MPB_E_INVALID, "read of %d bases does not fit a %d-byte row", len, row_bytes);
    for (int i = 0; i < len; i++) {
        const int q = (int)(unsigned char)qual_ascii[i] - offset;     // ref: moira.py:1177
        if (q < 0) return fail(MPB_E_RANGE, "Qualities must have positive values.");
        if (q > 254) return fail(MPB_E_RANGE, "quality %d at base %d exceeds the encodable maximum 254", q, i);
        row_out[i] = pack_one(seq ? seq[i] : 'A', q);
    }
    memset(row_out + len, 0, (size_t)(row_bytes - len));
    return MPB_OK;
}

int mpb_pack_batch_ascii(const char *seq_cat, const char *qual_cat, const int64_t *off, int64_t n,
                         int32_t fastq_offset, int32_t max_len, int64_t row_stride, uint8_t *out, int32_t *lens_out)
{
    if (n < 0 || row_stride <= 0 || (n > 0 && (!qual_cat || !off || !out))) return fail(MPB_E_INVALID, "mpb_pack_batch_ascii: bad arguments");
    for (int64_t i = 0; i < n; i++) {
        int64_t len = off[i + 1] - off[i];
        if (max_len > 0 && len > max_len) len = max_len;
        if (len < 0 || len > row_stride) return fail(MPB_E_INVALID, "read %lld of %lld bases does not fit a %lld-byte row", (long long)i, (long long)len, (long long)row_stride);
        const char *sq = seq_cat ? seq_cat + off[i] : nullptr;
        const unsigned char *ql = (const unsigned char *)qual_cat + off[i];
        uint8_t *row = out + i * row_stride;
        for (int64_t k = 0; k < len; k++) {
            const int q = (int)ql[k] - fastq_offset;
            if (q < 0) return fail(MPB_E_RANGE, "Qualities must have positive values.");
            if (q > 254) return fail(MPB_E_RANGE, "quality %d exceeds the encodable maximum 254", q);
            row[k] = pack_one(sq ? sq[k] : 'A', q);
        }
        memset(row + len, 0, (size_t)(row_stride - len));
        if (lens_out) lens_out[i] = (int32_t)len;
    }
    return MPB_OK;
}

// ---- workspace ---------------------------------------------------------------------------------

static inline int64_t align_up(int64_t x, int64_t a) { return (x + a - 1) / a * a; }

static int ensure_workspace(mpb_ctx *c, int64_t n)
{
    if (!c->ws_small) {
        // tables (2), overflow counter, bad-length counter, pass counter, overflow total of a host-pipeline call
        const size_t bytes = 2 * align_up(sizeof(MpbTables), 256) + 1024;
        HIPCHK(hipMalloc(&c->ws_small, bytes));
        // on the context's stream, not the null stream: the stream is non-blocking, so a null-stream
        // memset is unordered with the kernels below and can land after they have written the tables
        HIPCHK(hipMemsetAsync(c->ws_small, 0, bytes, c->stream));
        char *p = (char *)c->ws_small;
        c->ws.tables = (MpbTables *)p;
        c->ws.tables2 = (MpbTables *)(p + align_up(sizeof(MpbTables), 256));
        c->ws.ovf_count = (int32_t *)(p + 2 * align_up(sizeof(MpbTables), 256));
        c->ws.bad_len = (int32_t *)(p + 2 * align_up(sizeof(MpbTables), 256) + 64);
        c->ws.pass_count = (unsigned long long *)(p + 2 * align_up(sizeof(MpbTables), 256) + 256);
        c->ws.ovf_total = (long long *)(p + 2 * align_up(sizeof(MpbTables), 256) + 320);
        c->ws.wide_count = (int32_t *)(p + 2 * align_up(sizeof(MpbTables), 256) + 384);
        c->ws.alg_cells = (unsigned long long *)(p + 2 * align_up(sizeof(MpbTables), 256) + 448);
        c->ws.nar_count = (int32_t *)(p + 2 * align_up(sizeof(MpbTables), 256) + 512);
        c->ws.nar_sample = (int32_t *)(p + 2 * align_up(sizeof(MpbTables), 256) + 576);      // MPB_NAR_BUCKETS ints
        c->ws.lut = c->d_lut;
    }
    if (n <= c->ws_cap) return MPB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    serve_quiesce(c);
    if (c->ws_block) { HIPCHK(hipFree(c->ws_block)); c->ws_block = nullptr; c->ws_cap = 0; }
    const int64_t cap = n + n / 8 + 1024;
    const int64_t nb = (cap + MPB_PRE_READS - 1) / MPB_PRE_READS;
    const int64_t b_cls = align_up(cap, 256);
    const int64_t b_perm = align_up((cap + (int64_t)MPB_NCLS * 64) * 4, 256);
    const int64_t b_hist = align_up(nb * MPB_SKEYS * 4, 256);
    const int64_t b_ovf = align_up(cap * 4, 256);
    const int64_t b_pns = align_up((cap + (int64_t)MPB_NCLS * 64) * 2, 256);
    HIPCHK(hipMalloc(&c->ws_block, (size_t)(b_cls + b_perm + b_pns + b_hist + b_ovf)));
    char *p = (char *)c->ws_block;
    c->ws.cls = (uint8_t *)p; p += b_cls;
    c->ws.perm = (int32_t *)p; p += b_perm;
    c->ws.perm_ns = (uint16_t *)p; p += b_pns;
    c->ws.blockhist = (int32_t *)p; p += b_hist;
    c->ws.ovf_list = (int32_t *)p;
    c->ws_cap = cap;
    return MPB_OK;
}

// the wide-read list of a batch whose rows can hold more than MPB_TILE_MAX_ROWS - 1 bases (8 bytes per read)
static int ensure_wide_workspace(mpb_ctx *c, int64_t n)
{
    if (n <= c->ws_wide_cap) return MPB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    serve_quiesce(c);
    if (c->ws_wide) { HIPCHK(hipFree(c->ws_wide)); c->ws_wide = nullptr; c->ws_wide_cap = 0; c->ws.wide_list = c->ws.wide_rows = nullptr; }
    const int64_t cap = n + n / 8 + 1024;
    HIPCHK(hipMalloc(&c->ws_wide, (size_t)(2 * align_up(cap * 4, 256))));
    c->ws.wide_list = (int32_t *)c->ws_wide;
    c->ws.wide_rows = (int32_t *)((char *)c->ws_wide + align_up(cap * 4, 256));
    c->ws_wide_cap = cap;
    return MPB_OK;
}

// ---- timing ------------------------------------------------------------------------------------

static hipEvent_t get_event(mpb_ctx *c)
{
    if (!c->event_pool.empty()) { hipEvent_t e = c->event_pool.back(); c->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

struct Span {
    mpb_ctx *c; int kid; hipEvent_t a = nullptr, b = nullptr;
    Span(mpb_ctx *c_, int kid_) : c(c_), kid(kid_)
    {
        if (c->timing) { a = get_event(c); b = get_event(c); (void)hipEventRecord(a, c->stream); }
    }
    ~Span()
    {
        if (c->timing) { (void)hipEventRecord(b, c->stream); c->spans.push_back({kid, a, b}); }
    }
};

static int resolve_spans(mpb_ctx *c)
{
    if (c->spans.empty()) return MPB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (auto &s : c->spans) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) { c->acc_ms[s.kid] += ms; c->acc_n[s.kid] += 1; }
        c->event_pool.push_back(s.a);
        c->event_pool.push_back(s.b);
    }
    c->spans.clear();
    return MPB_OK;
}

int mpb_timing_enable(mpb_ctx *c, int on) { CTXCHK(c); c->timing = on != 0; return MPB_OK; }

int mpb_timing_reset(mpb_ctx *c)
{
    CTXCHK(c);
    int rc = resolve_spans(c);
    if (rc) return rc;
    for (int k = 0; k < MPB_K_COUNT; k++) { c->acc_ms[k] = 0; c->acc_n[k] = 0; }
    return MPB_OK;
}

int mpb_kernel_time(mpb_ctx *c, int kid, double *total_ms, int64_t *launches)
{
    CTXCHK(c);
    if (kid < 0 || kid >= MPB_K_COUNT) return fail(MPB_E_INVALID, "kernel id %d out of range", kid);
    int rc = resolve_spans(c);
    if (rc) return rc;
    if (total_ms) *total_ms = c->acc_ms[kid];
    if (launches) *launches = c->acc_n[kid];
    return MPB_OK;
}

// ---- the hot path --------------------------------------------------------------------------------

// Phi^-1 (Acklam's rational approximation, |rel err| < 1.2e-9): only steers the row-budget
// prediction, never a result.
static double inv_norm_cdf(double p)
{
    static const double a[] = {-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                               1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00};
    static const double b[] = {-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
                               6.680131188771972e+01, -1.328068155288572e+01};
    static const double cc[] = {-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
                                -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00};
    static const double d[] = {7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
                               3.754408661907416e+00};
    const double pl = 0.02425;
    if (p < pl) {
        double q = sqrt(-2 * log(p));
        return (((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) /
               ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    if (p > 1 - pl) {
        double q = sqrt(-2 * log(1 - p));
        return -(((((cc[0] * q + cc[1]) * q + cc[2]) * q + cc[3]) * q + cc[4]) * q + cc[5]) /
               ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1);
    }
    double q = p - 0.5, r = q * q;
    return (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q /
           (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1);
}

static int check_params(const mpb_filter_params *p)
{
    if (!p) return fail(MPB_E_INVALID, "params is NULL");
    if (!(p->alpha > 0 && p->alpha < 1))                       // ref: bernoullimodule.c:79-83
        return fail(MPB_E_INVALID, "Alpha must be between 0 and 1");
    if (p->ambig_mode < 0 || p->ambig_mode > 2) return fail(MPB_E_INVALID, "unknown ambig_mode %d", p->ambig_mode);
    if ((p->flags & MPB_FLAG_FAST_FMA) && p->alpha < 1e-5)      // the fma error grows like 1/alpha (include/moira_pb.h)
        return fail(MPB_E_INVALID, "MPB_FLAG_FAST_FMA needs alpha >= 1e-5 (its error bound does not hold below)");
    const bool has_me = p->maxerrors == p->maxerrors;
    if (has_me && !(p->maxerrors > 0)) return fail(MPB_E_INVALID, "maxerrors must be > 0");           // moira.py:732
    if (!has_me && !(p->uncert > 0 && p->uncert <= 1)) return fail(MPB_E_INVALID, "uncert must be in (0,1]");  // moira.py:728
    return MPB_OK;
}

static MpbDevParams make_dev_params(const mpb_filter_params *p, int32_t fixed_len, int32_t max_len)
{
    MpbDevParams d;
    d.thr = 1 - p->alpha;                     // same double expression as the reference
    d.uncert = p->uncert;
    d.maxerrors = p->maxerrors;
    const double z = inv_norm_cdf(1 - p->alpha);
    d.z = (float)z;
    d.zq = (float)((z * z - 1) / 6);
    d.clow = (float)(sqrt(2 * log(1 / (1 - p->alpha))) * 1.0001);
    d.ambig_mode = p->ambig_mode;
    d.flags = p->flags;
    d.fixed_len = fixed_len;
    d.max_len = max_len;
    d.len_shift = MPB_LEN_SHIFT;
    while (((max_len - 1) >> d.len_shift) >= MPB_LEN_BINS) d.len_shift++;
    return d;
}

// argument checks shared by mpb_filter_device and the classified-at-source pair; *max_len_out = longest read the batch can hold
static int check_device_batch(const void *d_q, int64_t n, int64_t row_stride, const int32_t *d_len, int32_t fixed_len,
                              const void *d_ee, const void *d_ns, const void *d_pass, int32_t *max_len_out)
{
    if (n < 0) return fail(MPB_E_INVALID, "n < 0");
    if (n > 0x7fffffffll - 4096) return fail(MPB_E_INVALID, "batch of %lld reads exceeds 2^31; split it", (long long)n);
    if (row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "row_stride must be a positive multiple of 16");
    if (row_stride > MPB_MAX_STRIDE) return fail(MPB_E_INVALID, "row_stride %lld exceeds %d (reads longer than %d bases are not supported)", (long long)row_stride, MPB_MAX_STRIDE, MPB_MAX_LEN);
    if (((uintptr_t)d_q & 15) != 0) return fail(MPB_E_INVALID, "quality matrix must be 16-byte aligned");
    if (!d_len && (fixed_len < 0 || fixed_len > row_stride)) return fail(MPB_E_INVALID, "fixed_len %d does not fit row_stride %lld", fixed_len, (long long)row_stride);
    const int32_t max_len = d_len ? (int32_t)(row_stride < MPB_MAX_LEN ? row_stride : MPB_MAX_LEN) : fixed_len;
    if (max_len > MPB_MAX_LEN)
        return fail(MPB_E_INVALID, "reads longer than %d bases are not supported", MPB_MAX_LEN);
    if (n > 0 && (!d_q || !d_ee || !d_ns || !d_pass)) return fail(MPB_E_INVALID, "NULL device buffer");
    *max_len_out = max_len;
    return MPB_OK;
}

// workspace (and, for batches whose rows can hold reads of more than 1023 bases, the wide-read list) of a batch of n reads
static int prepare_batch(mpb_ctx *c, int64_t n, int32_t max_len)
{
    int rc = ensure_workspace(c, n);
    if (rc) return rc;
    // reads of more than 1023 bases may need more DP rows than one wave holds: such reads are listed by the prepass
    // and run by the wide kernel (k_wide).  Batches whose rows cannot hold such a read never see any of it.
    if (max_len + 1 > MPB_TILE_MAX_ROWS) {
        if ((rc = ensure_wide_workspace(c, n))) return rc;
        HIPCHK(hipMemsetAsync(c->ws.wide_count, 0, sizeof(int32_t), c->stream));
    }
    return MPB_OK;
}

// everything after the classification: scan -> scatter -> DP -> wide reads -> overflow pass (-> counts)
static int filter_device_tail(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride, const int32_t *d_len,
                              const MpbDevParams &prm, double *d_ee, int32_t *d_ns, uint8_t *d_pass, mpb_filter_counts *counts,
                              const int32_t *d_list = nullptr)
{
    hipStream_t s = c->stream;
    const int32_t max_len = prm.max_len;
    { Span t(c, MPB_K_SCAN);     mpb_launch_scan(n, d_len, c->ws, s); }
    { Span t(c, MPB_K_SCATTER);  mpb_launch_scatter(n, d_len, d_ns, prm, c->ws, s, d_list); }
    { Span t(c, MPB_K_DP);       mpb_launch_dp(d_q, n, row_stride, d_len, prm, c->ws, d_ns, d_ee, d_pass, s); }
    if (max_len + 1 > MPB_TILE_MAX_ROWS) { Span t(c, MPB_K_WIDE); mpb_launch_wide(d_q, row_stride, d_len, prm, c->ws, d_ns, d_ee, d_pass, s); }
    { Span t(c, MPB_K_OVERFLOW); mpb_launch_overflow(d_q, n, row_stride, d_len, prm, c->ws, d_ns, d_ee, d_pass, s); }
    HIPCHK(hipGetLastError());
    if (counts) {
        mpb_launch_count(d_pass, n, c->ws, s);
        unsigned long long np = 0;
        int32_t novf = 0, bad = 0;
        HIPCHK(hipMemcpyAsync(&np, c->ws.pass_count, sizeof(np), hipMemcpyDeviceToHost, s));
        HIPCHK(hipMemcpyAsync(&novf, c->ws.ovf_count, sizeof(novf), hipMemcpyDeviceToHost, s));
        if (d_len) HIPCHK(hipMemcpyAsync(&bad, c->ws.bad_len, sizeof(bad), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (bad) {
            // a length in d_len was negative or above max_len: those reads were given ee = NaN, pass = 0 by the prepass
            // (never a result computed on a clamped length), and the call that fetches the counts fails
            HIPCHK(hipMemsetAsync(c->ws.bad_len, 0, sizeof(int32_t), s));
            return fail(MPB_E_INVALID, "%d read length(s) in d_len outside 0..%d (row_stride %lld; reads longer than %d bases "
                        "are not supported); those reads were given ee = NaN, pass = 0", bad, max_len, (long long)row_stride, MPB_MAX_LEN);
        }
        counts->n_pass = (int64_t)np;
        counts->n_fail = n - (int64_t)np;
        counts->n_overflow = novf;
    }
    return MPB_OK;
}

// ---- the natural-order narrow pass (round 5; mpb_internal.h, include/moira_pb.h: mpb_path_info) ---------------------------

static int ensure_narrow_workspace(mpb_ctx *c, int64_t n)
{
    if (n <= c->ws_nar_cap) return MPB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    serve_quiesce(c);
    if (c->ws_nar) { HIPCHK(hipFree(c->ws_nar)); c->ws_nar = nullptr; c->ws_nar_cap = 0; }
    const int64_t cap = n + n / 8 + 1024;
    const int64_t b_list = align_up((cap + 64) * 4, 256), b_waves = align_up((int64_t)(MPB_NAR_MAX_WAVES + 1) * 4, 256);
    const int64_t b_groups = align_up((cap / 64 + 2) * 4, 256);
    // (the ragged pass' order entries, 8 bytes per read, + group costs + wave ranges lie behind the lists)
    const int64_t b_wins = align_up((cap / 4096 + 2) * 8, 256);
    HIPCHK(hipMalloc(&c->ws_nar, (size_t)(4 * b_list + 3 * b_waves + b_groups + 2 * b_wins)));
    char *p = (char *)c->ws_nar;
    c->ws.nar_seg = (int32_t *)p; p += b_list;
    c->ws.nar_list = (int32_t *)p; p += b_list;
    c->ws.nar_wave_count = (int32_t *)p; p += b_waves;
    c->ws.nar_wave_off = (int32_t *)p; p += b_waves;
    c->ws.rg_ord = (int2 *)p; p += 2 * b_list;
    c->ws.rg_gpre = (int32_t *)p; p += b_groups;
    c->ws.rg_wsum = (unsigned long long *)p; p += b_wins;
    c->ws.rg_wpre = (unsigned long long *)p; p += b_wins;
    c->ws.rg_gstart = (int32_t *)p;
    c->ws_nar_cap = cap;
    return MPB_OK;
}

// The choice, from the sample's histogram of rows.  Costs per read in units of 0.083 ms per 10 M reads of 300 bases, fitted to
// measured steps (profiles/r05_narrow_rate.txt; round 6: profiles/r06_narrow_mix.txt, r06_ragged_mix.txt): the narrow pass 4 + 3 R for
// every read whatever it needs; the sorted pipeline 9 (classification + sort: its second read of the matrix) + 3.9 per row of the
// read's class (its predictor gives the few-row reads one row more than they need); a read the narrow pass hands back pays the sorted
// pipeline's cost x 1.05 + 1 -- since round 6 it runs through that pipeline where it lies (round 5 gathered it into a dense sub-batch:
// x 1.3 + 3), so the pass wins up to about 40 % handed back (round 5: 20 %).  The pass must finish at least 55 % of the sample and
// promise at least 5 %.
static int narrow_rows_from_sample(const int32_t *hist, int n_sample)
{
    if (n_sample <= 0) return 0;
    auto sorted_cost = [](int r) {
        const int cap = r == MPB_NAR_BUCKETS - 1 ? 30 : r <= 4 ? (r < 1 ? 2 : r + 1) : r;
        return 9.0 + 3.9 * cap;
    };
    const double lower_n_cost = sorted_cost(6);              // a read with a lower-case 'n' (handed back by the pass): a middling class
    double general = hist[0] * lower_n_cost;
    for (int r = 1; r < MPB_NAR_BUCKETS; r++) general += hist[r] * sorted_cost(r);
    int best = 0;
    double best_cost = 0.95 * general;
    for (int R = MPB_NAR_MIN_ROWS; R <= MPB_NAR_MAX_ROWS; R++) {
        int64_t done = 0;
        for (int r = 1; r <= R; r++) done += hist[r];
        if ((double)done < 0.55 * n_sample) continue;
        double cost = (double)n_sample * (4.0 + 3.0 * R) + hist[0] * (1.05 * lower_n_cost + 1.0);
        for (int r = R + 1; r < MPB_NAR_BUCKETS; r++) cost += hist[r] * (1.05 * sorted_cost(r) + 1.0);
        if (cost < best_cost) { best = R; best_cost = cost; }
    }
    return best;
}

// Fixed-length batches, and (round 6) ragged ones whose rows hold up to MPB_RG_MAX_STRIDE bytes (k_narrow_rg).
static bool narrow_eligible(const mpb_ctx *c, int64_t n, int64_t row_stride, const int32_t *d_len, int32_t fixed_len, const mpb_filter_params *p)
{
    const uint32_t forbidden = MPB_FLAG_FAST_FMA | MPB_FLAG_TEST_UNDERPREDICT | MPB_FLAG_DECISION_ONLY | MPB_FLAG_COUNT_CELLS | MPB_FLAG_NO_NARROW;
    if (!(c->narrow_ok && n >= 1 && !(p->flags & forbidden) && c->ws.lut == c->d_lut)) return false;
    return d_len ? row_stride <= MPB_RG_MAX_STRIDE : fixed_len >= 1;
}

// d_list != nullptr: the sorted pipeline over the n reads d_list[0 .. n) of a matrix of n_matrix reads (the reads a narrow pass hands
// back), where they lie: rows, lengths, class bytes and results are addressed by read (so the workspace is sized for the matrix: about
// 12 bytes per read, what the sorted pipeline would have taken for the same batch), the permutation by position; counts must be NULL.
static int filter_device_general(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride, const int32_t *d_len,
                                 int32_t fixed_len, int32_t max_len, const mpb_filter_params *params,
                                 double *d_ee, int32_t *d_ns, uint8_t *d_pass, mpb_filter_counts *counts,
                                 const int32_t *d_list = nullptr, int64_t n_matrix = 0)
{
    int rc;
    if ((rc = prepare_batch(c, d_list && n_matrix > n ? n_matrix : n, max_len))) return rc;
    const MpbDevParams prm = make_dev_params(params, fixed_len, max_len);
    if (params->flags & MPB_FLAG_COUNT_CELLS) HIPCHK(hipMemsetAsync(c->ws.alg_cells, 0, sizeof(unsigned long long), c->stream));
    { Span t(c, MPB_K_PREPASS);  mpb_launch_prepass(d_q, n, row_stride, d_len, prm, c->ws, d_ns, d_ee, d_pass, c->stream, d_list); }
    return filter_device_tail(c, d_q, n, row_stride, d_len, prm, d_ee, d_ns, d_pass, counts, d_list);
}

// 0: the sorted pipeline; 2..4: the narrow pass with that many rows.  May draw a sample (one small launch + a synchronisation).
// *split (ragged batches, *rows0 >= 3): groups whose longest read has at most that many 16-byte chunks run with a row less (0: none)
static int narrow_choose(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride, const int32_t *d_len, int32_t fixed_len,
                         const mpb_filter_params *params, const MpbDevParams &prm, int *rows0, int *split)
{
    *rows0 = 0;
    *split = 0;
    c->last_path.sampled = 0;
    const int forced = (int)((params->flags >> 8) & 15u);
    if (forced) {
        *rows0 = forced < MPB_NAR_MIN_ROWS ? MPB_NAR_MIN_ROWS : forced > MPB_NAR_MAX_ROWS ? MPB_NAR_MAX_ROWS : forced;
        *split = d_len ? (int)((params->flags >> 12) & 255u) : 0;          // MPB_FLAG_NARROW_SPLIT (test / measurement hook)
        return MPB_OK;
    }
    if (n < MPB_NAR_AUTO_MIN_READS) return MPB_OK;
    auto &ch = c->nar_choice;
    if (d_len) fixed_len = -1;                                // (a ragged batch of the same shape is another batch)
    const bool same = ch.valid && ch.n == n && ch.stride == row_stride && ch.fixed_len == fixed_len &&
                      memcmp(&ch.alpha, &params->alpha, sizeof(double)) == 0 && ch.flags == params->flags;
    if (same && ch.calls < 64) { ch.calls++; *rows0 = ch.rows0; *split = ch.split; return MPB_OK; }
    // a sample of <= 0.1 % of the reads: the prepass' row prediction on 256 .. 4096 reads spread over the batch
    int n_sample = (int)(n / 1024 < 256 ? 256 : n / 1024 > 4096 ? 4096 : n / 1024);
    { Span t(c, MPB_K_SAMPLE); mpb_launch_sample(d_q, n, row_stride, fixed_len, d_len, prm, c->ws, n_sample, c->stream); }
    HIPCHK(hipMemcpyAsync(c->pin_words + 16, c->ws.nar_sample, (MPB_NAR_BUCKETS + 2) * sizeof(int32_t), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    c->last_path.sampled = 1;
    int weight = 0;                                           // reads, or (ragged batches) their 16-byte chunks
    for (int k = 0; k < MPB_NAR_BUCKETS; k++) { c->last_path.sample_hist[k] = c->pin_words[16 + k]; weight += c->pin_words[16 + k]; }
    *rows0 = narrow_rows_from_sample(c->pin_words + 16, weight);
    if (d_len && *rows0 >= 3) {
        // mixed rows: the shortest sampled read that needs rows0 rows has `need` chunks; groups safely below it take a row less
        // (a sixteenth of margin and a chunk -- a cut that turns out too bold is dropped after the call that shows it, below; not
        // worth a second code path below four chunks)
        const int need = c->pin_words[16 + MPB_NAR_BUCKETS + (*rows0 - 3)];
        if (need > 0 && need < (1 << 20)) {
            const int cut = need - 1 - need / 16;
            if (cut >= 4) *split = cut > 255 ? 255 : cut;
        }
    }
    ch.split = *split;
    {
        double back = c->pin_words[16];
        for (int r = *rows0 + 1; r < MPB_NAR_BUCKETS; r++) back += c->pin_words[16 + r];
        ch.expect_back = weight > 0 ? back / weight : 0.0;
    }
    ch.valid = true; ch.n = n; ch.stride = row_stride; ch.fixed_len = fixed_len; ch.alpha = params->alpha; ch.flags = params->flags;
    ch.rows0 = *rows0; ch.calls = 0;
    return MPB_OK;
}

static int filter_device_narrow(mpb_ctx *c, int rows0, int split, const uint8_t *d_q, int64_t n, int64_t row_stride, const int32_t *d_len,
                                int32_t fixed_len, const mpb_filter_params *params, const MpbDevParams &prm,
                                double *d_ee, int32_t *d_ns, uint8_t *d_pass, mpb_filter_counts *counts)
{
    int rc;
    hipStream_t s = c->stream;
    if ((rc = ensure_narrow_workspace(c, n))) return rc;
    // persistent grid: as many workgroups as fit the CUs' LDS at once
    const int lds = d_len ? mpb_narrow_rs_lds_bytes()
                          : (mpb_narrow_rs_reads_per_lane(row_stride, rows0) ? mpb_narrow_rs_lds_bytes() : mpb_narrow_lds_bytes());
    const int per_cu = (160 * 1024) / lds;
    const int grid = (c->n_cu > 0 ? c->n_cu : 256) * (per_cu > 0 ? per_cu : 1);
    { Span t(c, MPB_K_NARROW);
      if (d_len) mpb_launch_narrow_ragged(rows0, split, d_q, n, row_stride, d_len, prm, c->ws, d_ee, d_ns, d_pass, c->ws.nar_list, grid, s);
      else mpb_launch_narrow(rows0, d_q, n, row_stride, fixed_len, prm, c->ws, d_ee, d_ns, d_pass, c->ws.nar_list, grid, s); }
    HIPCHK(hipGetLastError());
    HIPCHK(hipMemcpyAsync(c->pin_words, c->ws.nar_count, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    const int64_t m = c->pin_words[0];
    c->last_path.narrow_rows = rows0;
    c->last_path.narrow_split = d_len && rows0 >= 3 ? split : 0;
    c->last_path.n_fallback = m;
    const bool forced = ((params->flags >> 8) & 15u) != 0;
    // a pass that hands back far more than its sample can have promised: look again next time
    if (!forced && m > n / 2) c->nar_choice.valid = false;
    // mixed rows whose cut was too bold (reads of the short groups needed the row they did not get -- the sample's shortest such read
    // was not the batch's -- so that clearly more came back than the sample's own share of unfinished reads): the next calls of this
    // shape run without a cut
    if (!forced && split > 0 && (double)m > (1.5 * c->nar_choice.expect_back + 0.002) * (double)n) c->nar_choice.split = 0;
    int32_t novf = 0;
    if (m > 0) {
        // The reads handed back go through the sorted pipeline WHERE THEY LIE (round 6: the prepass and the scatter walk the list,
        // the DP addresses rows and results by read anyway): no dense copy of their rows, no block of memory for it.  When they
        // are most of the batch (a stale choice met a batch of bad reads) the whole batch takes the sorted pipeline instead:
        // the same results, one pass over everything is cheaper than a list of most of it (break-even near one half).
        if (!forced && m > n / 2) {
            c->last_path.n_fallback = n;
            return filter_device_general(c, d_q, n, row_stride, d_len, fixed_len, prm.max_len, params, d_ee, d_ns, d_pass, counts);
        }
        mpb_filter_params sub = *params;
        sub.flags = (sub.flags & ~((15u << 8) | (255u << 12))) | MPB_FLAG_NO_NARROW;
        if ((rc = filter_device_general(c, d_q, m, row_stride, d_len, fixed_len, prm.max_len, &sub, d_ee, d_ns, d_pass, nullptr, c->ws.nar_list, n))) return rc;
        HIPCHK(hipGetLastError());
        if (counts) HIPCHK(hipMemcpyAsync(&novf, c->ws.ovf_count, sizeof(novf), hipMemcpyDeviceToHost, s));
    }
    if (counts) {
        if ((rc = ensure_workspace(c, 1))) return rc;
        HIPCHK(hipMemsetAsync(c->ws.pass_count, 0, sizeof(unsigned long long), s));
        mpb_launch_count(d_pass, n, c->ws, s);
        unsigned long long np = 0;
        int32_t bad = 0;
        HIPCHK(hipMemcpyAsync(&np, c->ws.pass_count, sizeof(np), hipMemcpyDeviceToHost, s));
        if (d_len) HIPCHK(hipMemcpyAsync(&bad, c->ws.bad_len, sizeof(bad), hipMemcpyDeviceToHost, s));
        HIPCHK(hipStreamSynchronize(s));
        if (bad) {
            // raised by an earlier call that fetched no counts (include/moira_pb.h: "the next call that fetches counts fails")
            HIPCHK(hipMemsetAsync(c->ws.bad_len, 0, sizeof(int32_t), s));
            return fail(MPB_E_INVALID, "%d read length(s) in d_len outside 0..%d (row_stride %lld; reads longer than %d bases "
                        "are not supported); those reads were given ee = NaN, pass = 0", bad, prm.max_len, (long long)row_stride, MPB_MAX_LEN);
        }
        counts->n_pass = (int64_t)np;
        counts->n_fail = n - (int64_t)np;
        counts->n_overflow = novf;
    }
    return MPB_OK;
}

int mpb_filter_device(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride,
                      const int32_t *d_len, int32_t fixed_len, const mpb_filter_params *params,
                      double *d_ee, int32_t *d_ns, uint8_t *d_pass, mpb_filter_counts *counts)
{
    CTXCHK(c);
    int rc = check_params(params);
    if (rc) return rc;
    int32_t max_len = 0;
    if ((rc = check_device_batch(d_q, n, row_stride, d_len, fixed_len, d_ee, d_ns, d_pass, &max_len))) return rc;
    if (counts) { counts->n_reads = n; counts->n_pass = 0; counts->n_fail = 0; counts->n_overflow = 0; }
    if (n == 0) return MPB_OK;
    c->classified.valid = false;                     // the workspace now describes THIS batch
    c->last_path = mpb_path_info{};
    if ((rc = ensure_workspace(c, 1))) return rc;    // the small block (tables, counters) exists from here on
    if (narrow_eligible(c, n, row_stride, d_len, fixed_len, params)) {
        const MpbDevParams prm = make_dev_params(params, fixed_len, max_len);
        int rows0 = 0, split = 0;
        if ((rc = narrow_choose(c, d_q, n, row_stride, d_len, fixed_len, params, prm, &rows0, &split))) return rc;
        if (rows0) return filter_device_narrow(c, rows0, split, d_q, n, row_stride, d_len, fixed_len, params, prm, d_ee, d_ns, d_pass, counts);
    }
    return filter_device_general(c, d_q, n, row_stride, d_len, fixed_len, max_len, params, d_ee, d_ns, d_pass, counts);
}

int mpb_last_path(mpb_ctx *c, mpb_path_info *out)
{
    CTXCHK(c);
    if (!out) return fail(MPB_E_INVALID, "NULL output");
    *out = c->last_path;
    return MPB_OK;
}

int mpb_last_algorithmic_cells(mpb_ctx *c, int64_t *cells)
{
    CTXCHK(c);
    if (!cells) return fail(MPB_E_INVALID, "NULL output");
    if (!c->ws_small) { *cells = 0; return MPB_OK; }
    unsigned long long v = 0;
    HIPCHK(hipMemcpyAsync(&v, c->ws.alg_cells, sizeof(v), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    *cells = (int64_t)v;
    return MPB_OK;
}

// ---- classified at source (SURVEY f-4 + VERDICT r2 #6) -------------------------------------------------------------
// A batch that is PRODUCED on the device -- raw FASTQ text decoded into the packed matrix -- is classified by the kernel
// that produces it: the decode pass already holds every byte in registers, so it also sums the prediction statistics,
// counts the ambiguous bases and fills the class histograms.  mpb_filter_device_classified then starts at the scan;
// the packed matrix is read once (by the DP), not twice.

int mpb_decode_classify_device(mpb_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, int64_t n, int64_t row_stride,
                               const int32_t *d_len, int32_t fixed_len, int32_t fastq_offset,
                               const mpb_filter_params *params, uint8_t *d_q_out, double *d_ee, int32_t *d_ns,
                               uint8_t *d_pass, int32_t *d_err)
{
    CTXCHK(c);
    int rc = check_params(params);
    if (rc) return rc;
    int32_t max_len = 0;
    if ((rc = check_device_batch(d_q_out, n, row_stride, d_len, fixed_len, d_ee, d_ns, d_pass, &max_len))) return rc;
    if ((((uintptr_t)d_seq | (uintptr_t)d_qual) & 15) != 0) return fail(MPB_E_INVALID, "matrices must be 16-byte aligned");
    if (n > 0 && (!d_seq || !d_qual)) return fail(MPB_E_INVALID, "NULL device buffer");
    // the fused pass parks a tile of WHOLE rows in LDS (k_classify_linear: 1,280 chunks): a row of more than 16384 bytes does not
    // fit one -- such text goes through mpb_decode_ascii_device + mpb_filter_device, which take rows of up to 65536 bytes
    if (row_stride > MPB_SMALL_MAX_STRIDE)
        return fail(MPB_E_INVALID, "row_stride %lld: the classify-at-source pass takes rows of up to %d bytes (decode with "
                    "mpb_decode_ascii_device, then mpb_filter_device)", (long long)row_stride, MPB_SMALL_MAX_STRIDE);
    c->classified.valid = false;
    if (n == 0) return MPB_OK;
    if ((rc = prepare_batch(c, n, max_len))) return rc;
    const MpbDevParams prm = make_dev_params(params, fixed_len, max_len);
    { Span t(c, MPB_K_PREPASS);
      mpb_launch_decode_classify(d_seq, d_qual, fastq_offset, d_q_out, d_err, n, row_stride, d_len, prm, c->ws, d_ns, d_ee, d_pass, c->stream); }
    HIPCHK(hipGetLastError());
    c->classified = {true, d_q_out, n, row_stride, d_len, fixed_len, *params, d_ee, d_ns, d_pass};
    return MPB_OK;
}

int mpb_filter_device_classified(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride,
                                 const int32_t *d_len, int32_t fixed_len, const mpb_filter_params *params,
                                 double *d_ee, int32_t *d_ns, uint8_t *d_pass, mpb_filter_counts *counts)
{
    CTXCHK(c);
    int rc = check_params(params);
    if (rc) return rc;
    int32_t max_len = 0;
    if ((rc = check_device_batch(d_q, n, row_stride, d_len, fixed_len, d_ee, d_ns, d_pass, &max_len))) return rc;
    if (counts) { counts->n_reads = n; counts->n_pass = 0; counts->n_fail = 0; counts->n_overflow = 0; }
    if (n == 0) return MPB_OK;
    const auto &k = c->classified;
    const bool same_params = k.valid && memcmp(&k.params.alpha, &params->alpha, sizeof(double)) == 0 &&
                             memcmp(&k.params.uncert, &params->uncert, sizeof(double)) == 0 &&
                             memcmp(&k.params.maxerrors, &params->maxerrors, sizeof(double)) == 0 &&
                             k.params.ambig_mode == params->ambig_mode && k.params.flags == params->flags;
    if (!same_params || k.q != d_q || k.n != n || k.stride != row_stride || k.len != d_len || k.fixed_len != fixed_len ||
        k.ee != d_ee || k.ns != d_ns || k.pass != d_pass)
        return fail(MPB_E_INVALID, "mpb_filter_device_classified: the last call on this context that classified a batch "
                                   "(mpb_decode_classify_device) did not produce THIS batch with THESE parameters and result arrays");
    c->classified.valid = false;                     // consumed: the DP passes overwrite parts of the workspace
    const MpbDevParams prm = make_dev_params(params, fixed_len, max_len);
    return filter_device_tail(c, d_q, n, row_stride, d_len, prm, d_ee, d_ns, d_pass, counts);
}

int mpb_encode_ascii_device(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride, int32_t fastq_offset,
                            uint8_t *d_seq_out, uint8_t *d_qual_out)
{
    CTXCHK(c);
    if (n < 0 || row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "bad matrix shape");
    if ((((uintptr_t)d_q | (uintptr_t)d_seq_out | (uintptr_t)d_qual_out) & 15) != 0) return fail(MPB_E_INVALID, "matrices must be 16-byte aligned");
    if (n == 0) return MPB_OK;
    if (!d_q || !d_seq_out || !d_qual_out) return fail(MPB_E_INVALID, "NULL device buffer");
    if (n * (row_stride / 16) / 256 > 0x7fffffffll) return fail(MPB_E_INVALID, "too large for one launch; split it");
    mpb_launch_encode(d_q, n, row_stride, fastq_offset, d_seq_out, d_qual_out, c->stream);
    HIPCHK(hipGetLastError());
    return MPB_OK;
}

int mpb_last_class_histogram(mpb_ctx *c, int32_t *caps, int64_t *cnts, int32_t max_classes)
{
    CTXCHK(c);
    if (!caps || !cnts || max_classes < 0) return fail(MPB_E_INVALID, "bad arguments");
    if (c->last_path.narrow_rows != 0)      // (the workspace then describes the handed-back sub-batch, or an earlier batch: ADVICE r5)
        return fail(MPB_E_INVALID, "the last filter call took the narrow pass (%d rows): there is no class histogram of its batch; "
                                   "call with MPB_FLAG_NO_NARROW to get one", c->last_path.narrow_rows);
    if (!c->ws.tables) return 0;
    static const MpbClass classes[MPB_NCLS] = MPB_CLASS_TABLE;
    MpbTables h;
    HIPCHK(hipMemcpyAsync(&h, c->ws.tables, sizeof(h), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    int k = 0;
    for (; k < MPB_NCLS && k < max_classes; k++) { caps[k] = classes[k].cap; cnts[k] = h.count[k]; }
    return k;
}

int mpb_last_read_budgets(mpb_ctx *c, int32_t *caps_out, int64_t n)
{
    CTXCHK(c);
    if (!caps_out || n < 0) return fail(MPB_E_INVALID, "bad arguments");
    if (c->last_path.narrow_rows != 0)
        return fail(MPB_E_INVALID, "the last filter call took the narrow pass (%d rows): there are no per-read row budgets of its "
                                   "batch; call with MPB_FLAG_NO_NARROW to get them", c->last_path.narrow_rows);
    if (n > c->ws_cap || !c->ws.cls) return fail(MPB_E_INVALID, "no filter call of at least %lld reads precedes", (long long)n);
    static const MpbClass classes[MPB_NCLS] = MPB_CLASS_TABLE;
    std::vector<uint8_t> cls((size_t)n);
    HIPCHK(hipMemcpyAsync(cls.data(), c->ws.cls, (size_t)n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int64_t i = 0; i < n; i++) {
        const int k = cls[(size_t)i] & 0x7f;
        caps_out[i] = k < MPB_NCLS ? classes[k].cap : 0;
    }
    return MPB_OK;
}

static int ensure_stage(mpb_ctx *c, int64_t bytes)
{
    if (bytes <= c->stage_cap) return MPB_OK;
    HIPCHK(hipStreamSynchronize(c->stream));
    serve_quiesce(c);
    if (c->stage_dev) { HIPCHK(hipFree(c->stage_dev)); c->stage_dev = nullptr; c->stage_cap = 0; }
    HIPCHK(hipMalloc(&c->stage_dev, (size_t)bytes));
    c->stage_cap = bytes;
    return MPB_OK;
}

// Batches of at most MPB_SMALL_N reads (the two paths cost the same at about 8 k reads): one host-to-device copy from pinned memory, one kernel (one read
// per wave), one copy back -- the batched pipeline's eight launches and six copies cost ~150 us whatever
// the size, which is all a per-read caller (bernoulli.calculate_errors_PB) would ever see.
#define MPB_SMALL_N 4096
// small batches whose inputs are at most this many bytes are read by the kernel straight from pinned host memory
// ... and batches of at most this many reads report their completion through a word per read in pinned memory
#define MPB_SMALL_FLAG_N 256
#define MPB_SMALL_ZC_BYTES (1 << 20)
// qualities per chunk of the host pipeline: large enough that a chunk's launch sequence (~0.2 ms fixed) is
// noise beside its 2 ms of PCIe time; the four slots then hold 4 x (128 MiB of qualities + 5.5 MiB of results) of device
// memory, the same again of pinned staging when the input is pageable, and 4 x 5.5 MiB of pinned results
#define MPB_HOST_CHUNK_BYTES (128ll << 20)

// Wait until done[0..n) all hold `token` (k_small writes it once a read's results are visible to the host).  The kernel
// normally takes tens of microseconds: spin; a launch that has not finished after 20 ms is asked about through the runtime
// (which also reports a fault), and its flags are then looked at once more.
int mpbi_wait_flags(const volatile uint32_t *done, int64_t n, uint32_t token, hipStream_t s)
{
    int64_t k = 0;
    const auto t0 = std::chrono::steady_clock::now();
    for (uint32_t spins = 0;; spins++) {
        while (k < n && done[k] == token) k++;
        if (k == n) { std::atomic_thread_fence(std::memory_order_acquire); return MPB_OK; }
        __builtin_ia32_pause();
        if ((spins & 1023u) == 1023u && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) {
            HIPCHK(hipStreamSynchronize(s));
            while (k < n && done[k] == token) k++;
            if (k == n) return MPB_OK;
            return fail(MPB_E_HIP, "k_small finished without reporting read %lld", (long long)k);
        }
    }
}

static int filter_host_small(mpb_ctx *c, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len,
                             int32_t fixed_len, const mpb_filter_params *params, double *ee, int32_t *ns,
                             uint8_t *pass, mpb_filter_counts *counts, bool *done)
{
    *done = false;
    const int32_t max_len = len ? (int32_t)(row_stride < MPB_MAX_LEN ? row_stride : MPB_MAX_LEN) : fixed_len;
    const int64_t b_q = align_up(n * row_stride, 256), b_len = align_up(n * 4, 256);
    const int64_t b_ee = align_up(n * 8, 256), b_ns = align_up(n * 4, 256), b_pass = align_up(n, 256);
    const int64_t in_bytes = b_q + b_len, out_bytes = b_ee + b_ns + b_pass, b_done = align_up(n * 4, 256);
    int rc = ensure_stage(c, in_bytes + out_bytes);
    if (rc) return rc;
    rc = ensure_workspace(c, n);
    if (rc) return rc;
    c->classified.valid = false;                           // k_small rewrites the class bytes
    if (in_bytes + out_bytes + b_done > c->pin_cap) {
        serve_quiesce(c);
        if (c->pin_host) { HIPCHK(hipStreamSynchronize(c->stream)); HIPCHK(hipHostFree(c->pin_host)); c->pin_host = nullptr; c->pin_cap = 0; }
        const int64_t cap = 2 * (in_bytes + out_bytes + b_done);
        HIPCHK(hipHostMalloc(&c->pin_host, (size_t)cap, hipHostMallocMapped));
        c->pin_cap = cap;
    }
    char *h = (char *)c->pin_host, *d = (char *)c->stage_dev;
    memcpy(h, q, (size_t)(n * row_stride));
    if (len) memcpy(h + b_q, len, (size_t)(n * 4));
    uint8_t *d_q = (uint8_t *)d;
    int32_t *d_len = (int32_t *)(d + b_q);
    double *d_ee = (double *)(d + in_bytes);
    int32_t *d_ns = (int32_t *)(d + in_bytes + b_ee);
    uint8_t *d_pass = (uint8_t *)(d + in_bytes + b_ee + b_ns);
    const MpbDevParams prm = make_dev_params(params, fixed_len, max_len);
    char *ho = h + in_bytes;
    constexpr int64_t zc_bytes = MPB_SMALL_ZC_BYTES;
    if (in_bytes <= zc_bytes) {
        // a handful of reads (the per-read entry: ONE): the kernel reads the rows from, and writes the results to, the pinned
        // host block -- one runtime call instead of three dependent ones (copy in, kernel, copy out) -- and the host learns
        // that the results are there from a word per read in the same block, not from the runtime
        // (a word and a system-scope fence per read pay for a few reads -- 1: 33 -> 27 us, 64: 35 -> 32 -- and cost more than
        // the runtime's one completion signal for many: 1,024 reads 64 -> 82 us; hence the limit)
        const bool flags = n <= MPB_SMALL_FLAG_N;
        if (++c->small_token == 0) c->small_token = 1;
        volatile uint32_t *done = (volatile uint32_t *)(ho + out_bytes);
        for (int64_t i = 0; flags && i < n; i++) done[i] = 0;
        const MpbSmallHost hostside{d_q, d_ns, flags ? (uint32_t *)(ho + out_bytes) : nullptr, c->small_token};
        { Span t(c, MPB_K_DP); mpb_launch_small((const uint8_t *)h, n, row_stride, len ? (const int32_t *)(h + b_q) : nullptr, prm, c->ws,
                                                 (int32_t *)(ho + b_ee), (double *)ho, (uint8_t *)(ho + b_ee + b_ns), c->stream, &hostside); }
        HIPCHK(hipGetLastError());
        if (flags) {
            if ((rc = mpbi_wait_flags(done, n, c->small_token, c->stream)) != MPB_OK) return rc;
            if (c->timing) HIPCHK(hipStreamSynchronize(c->stream));      // the spans' events are read after the call
            goto have_results;
        }
    } else {
        HIPCHK(hipMemcpyAsync(d, h, (size_t)(len ? in_bytes : n * row_stride), hipMemcpyHostToDevice, c->stream));
        { Span t(c, MPB_K_DP); mpb_launch_small(d_q, n, row_stride, len ? d_len : nullptr, prm, c->ws, d_ns, d_ee, d_pass, c->stream); }
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(ho, d + in_bytes, (size_t)out_bytes, hipMemcpyDeviceToHost, c->stream));
    }
    HIPCHK(hipStreamSynchronize(c->stream));
have_results:
    const uint8_t *hp = (const uint8_t *)(ho + b_ee + b_ns);
    int64_t np = 0;
    for (int64_t i = 0; i < n; i++) {
        if (hp[i] == 2) return MPB_OK;                     // budget missed: the caller takes the batched path
        np += hp[i];
    }
    memcpy(ee, ho, (size_t)(n * 8));
    memcpy(ns, ho + b_ee, (size_t)(n * 4));
    memcpy(pass, hp, (size_t)n);
    if (counts) { counts->n_reads = n; counts->n_pass = np; counts->n_fail = n - np; counts->n_overflow = 0; }
    *done = true;
    return MPB_OK;
}

// ---- host pipeline (SURVEY a-9: chunked batch, async H2D / compute / D2H) ------------------------
//
// Three streams, MPB_HOST_SLOTS (4) chunk slots.  Chunk k (slot k % 4):
//   copy_stream : H2D of its qualities (+ lengths)            -> h2d_done
//   stream      : waits h2d_done, the five kernels of the pass -> k_done
//   out_stream  : waits k_done, D2H of ee / Ns / pass          -> d2h_done
// so that the H2D of chunk k+1 and the D2H of chunk k-1 run beside the kernels of chunk k.  A slot is
// reused only after its previous chunk's d2h_done (which implies its kernels and its H2D are done); a chunk is
// retired three trips after it was queued, AFTER the current trip's work has been queued, so the copy stream
// never runs dry while the host thread copies results out.
// Inputs in pinned host memory (mpb_host_alloc, or registered by the caller) are DMA-ed where they lie;
// pageable inputs are first copied into the slot's pinned staging block by `copy_threads` threads, which
// happens while the GPU is busy with the chunks before.  Outputs always land in the slot's pinned block and
// are copied out when the slot is retired (13 bytes per read).

static bool is_pinned_host(const void *p, size_t bytes)
{
    if (!p || bytes == 0) return false;
    hipPointerAttribute_t a;
    const void *ends[2] = {p, (const char *)p + bytes - 1};
    for (const void *e : ends) {
        memset(&a, 0, sizeof(a));
        if (hipPointerGetAttributes(&a, e) != hipSuccess) { (void)hipGetLastError(); return false; }
        if (a.type != hipMemoryTypeHost) return false;
    }
    return true;
}

static int grow_block(void **p, int64_t *cap, int64_t bytes, bool pinned)
{
    if (bytes <= *cap) return MPB_OK;
    if (*p) { HIPCHK(pinned ? hipHostFree(*p) : hipFree(*p)); *p = nullptr; *cap = 0; }
    HIPCHK(pinned ? hipHostMalloc(p, (size_t)bytes, hipHostMallocDefault) : hipMalloc(p, (size_t)bytes));
    *cap = bytes;
    return MPB_OK;
}

struct ChunkLayout { int64_t q, len, ee, ns, pass, in_bytes, out_bytes; };

static ChunkLayout chunk_layout(int64_t m, int64_t row_stride)
{
    ChunkLayout L;
    L.q = align_up(m * row_stride, 256); L.len = align_up(m * 4, 256);
    L.ee = align_up(m * 8, 256); L.ns = align_up(m * 4, 256); L.pass = align_up(m, 256);
    L.in_bytes = L.q + L.len; L.out_bytes = L.ee + L.ns + L.pass;
    return L;
}

static void drain_pipeline(mpb_ctx *c)
{
    (void)hipStreamSynchronize(c->copy_stream);
    (void)hipStreamSynchronize(c->stream);
    (void)hipStreamSynchronize(c->out_stream);
    for (auto &sl : c->slot) sl.off = -1;
}

// what a pipeline run computes: the Poisson-binomial pass, or the Poisson approximation (lambda on the device, the
// scalar tail on the host when a chunk is retired -- i.e. beside the GPU work of the chunks after it)
struct PipeJob {
    int poisson;
    const mpb_filter_params *params;
    const int32_t *len;
    int32_t fixed_len;
    double *ee; int32_t *ns; uint8_t *pass;
};

// wait for the chunk parked in `sl` and hand its results to the caller's arrays
static int retire_slot(mpb_ctx *c, HostSlot &sl, int64_t cap_reads, int64_t row_stride, const PipeJob &job, int64_t *n_pass)
{
    if (sl.off < 0) return MPB_OK;
    HIPCHK(hipEventSynchronize(sl.d2h_done));
    const ChunkLayout L = chunk_layout(cap_reads, row_stride);
    const char *h = (const char *)sl.pin_out;
    memcpy(job.ns + sl.off, h + L.ee, (size_t)(sl.m * 4));
    const uint8_t *hp;
    if (job.poisson) {
        // the block's first array holds lambda; ee / pass come from the reference's scalar loop (moira.py:1666-1679)
        int rc = mpb_poisson_finish_host((const double *)h, job.ns + sl.off, job.len ? job.len + sl.off : nullptr, job.fixed_len,
                                         sl.m, job.params, job.ee + sl.off, job.pass + sl.off);
        if (rc) return rc;
        hp = job.pass + sl.off;
    } else {
        memcpy(job.ee + sl.off, h, (size_t)(sl.m * 8));
        hp = (const uint8_t *)(h + L.ee + L.ns);
        memcpy(job.pass + sl.off, hp, (size_t)sl.m);
    }
    int64_t np = 0;
    for (int64_t i = 0; i < sl.m; i++) np += hp[i];
    *n_pass += np;
    sl.off = -1;
    return MPB_OK;
}

static int filter_host_pipeline(mpb_ctx *c, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len,
                                int32_t fixed_len, const mpb_filter_params *params, double *ee, int32_t *ns,
                                uint8_t *pass, mpb_filter_counts *counts, int poisson)
{
    const PipeJob job = {poisson, params, len, fixed_len, ee, ns, pass};
    // chunk: at most MPB_HOST_CHUNK_BYTES of qualities, and at least four chunks per batch where the batch
    // is large enough for a chunk to be worth a launch sequence (overlap needs more than one chunk)
    int64_t chunk = (int64_t)MPB_HOST_CHUNK_BYTES / row_stride;
    if (chunk > (n + 3) / 4) chunk = (n + 3) / 4;
    {   // ... but not below 16384 reads (a launch sequence's worth), nor -- rows of up to 64 KiB -- above 256 MiB for that floor
        int64_t floor_reads = (256ll << 20) / row_stride;
        if (floor_reads > 16384) floor_reads = 16384;
        if (floor_reads < 256) floor_reads = 256;
        if (chunk < floor_reads) chunk = floor_reads;
    }
    if (chunk > n) chunk = n;
    const ChunkLayout L = chunk_layout(chunk, row_stride);
    const bool q_pinned = is_pinned_host(q, (size_t)(n * row_stride));
    const bool len_pinned = len && is_pinned_host(len, (size_t)(n * 4));
    const int64_t nchunks = (n + chunk - 1) / chunk;
    const int nslots = (int)(nchunks < MPB_HOST_SLOTS ? nchunks : MPB_HOST_SLOTS);
    int rc = ensure_workspace(c, chunk);
    if (rc) return rc;
    HIPCHK(hipStreamSynchronize(c->stream));
    for (int k = 0; k < nslots; k++) {                       // a slot that must grow frees memory: the resident one-read kernel leaves first
        const HostSlot &sl = c->slot[k];
        if (L.in_bytes + L.out_bytes > sl.dev_cap || L.out_bytes > sl.pin_out_cap ||
            (!(q_pinned && (!len || len_pinned)) && L.in_bytes > sl.pin_in_cap)) { serve_quiesce(c); break; }
    }
    for (int k = 0; k < nslots; k++) {
        HostSlot &sl = c->slot[k];
        sl.off = -1;
        if ((rc = grow_block(&sl.dev, &sl.dev_cap, L.in_bytes + L.out_bytes, false))) return rc;
        if ((rc = grow_block(&sl.pin_out, &sl.pin_out_cap, L.out_bytes, true))) return rc;
        if (!(q_pinned && (!len || len_pinned)))
            if ((rc = grow_block(&sl.pin_in, &sl.pin_in_cap, L.in_bytes, true))) return rc;
    }
    HIPCHK(hipMemsetAsync(c->ws.ovf_total, 0, sizeof(long long), c->stream));
    // the host has validated every length: a count left behind by an earlier mpb_filter_device call that never
    // fetched its counts must not fail this one
    HIPCHK(hipMemsetAsync(c->ws.bad_len, 0, sizeof(int32_t), c->stream));
    if (poisson) HIPCHK(hipMemsetAsync(c->ws.ovf_count, 0, sizeof(int32_t), c->stream));     // reads with a byte 255, summed over the chunks
    int64_t n_pass = 0;
    int64_t k = 0;
    for (int64_t off = 0; off < n; off += chunk, k++) {
        const int64_t m = (n - off < chunk) ? n - off : chunk;
        HostSlot &sl = c->slot[k % MPB_HOST_SLOTS];          // free: its previous chunk (k - 4) was retired in trip k - 1
        char *d = (char *)sl.dev;
        uint8_t *d_q = (uint8_t *)d;
        int32_t *d_len = (int32_t *)(d + L.q);
        double *d_ee = (double *)(d + L.in_bytes);
        int32_t *d_ns = (int32_t *)(d + L.in_bytes + L.ee);
        uint8_t *d_pass = (uint8_t *)(d + L.in_bytes + L.ee + L.ns);
        // ---- H2D ----
        const uint8_t *src_q = q + off * row_stride;
        if (!q_pinned) {
            parallel_copy(sl.pin_in, src_q, (size_t)(m * row_stride), c->copy_threads);
            src_q = (const uint8_t *)sl.pin_in;
        }
        hipError_t e = hipMemcpyAsync(d_q, src_q, (size_t)(m * row_stride), hipMemcpyHostToDevice, c->copy_stream);
        if (e == hipSuccess && len) {
            const int32_t *src_len = len + off;
            if (!len_pinned) {
                memcpy((char *)sl.pin_in + L.q, src_len, (size_t)(m * 4));
                src_len = (const int32_t *)((char *)sl.pin_in + L.q);
            }
            e = hipMemcpyAsync(d_len, src_len, (size_t)(m * 4), hipMemcpyHostToDevice, c->copy_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(sl.h2d_done, c->copy_stream);
        // ---- kernels ----
        if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, sl.h2d_done, 0);
        if (e != hipSuccess) { drain_pipeline(c); return fail(MPB_E_HIP, "host pipeline (H2D): %s", hipGetErrorString(e)); }
        if (poisson) {
            Span t(c, MPB_K_LAMBDA);
            mpb_launch_lambda(d_q, m, row_stride, len ? d_len : nullptr, fixed_len, c->ws.lut, d_ee, d_ns, c->ws.ovf_count, c->stream);
            rc = hipGetLastError() == hipSuccess ? MPB_OK : fail(MPB_E_HIP, "k_lambda launch failed");
        } else {
            // (the chunks always take the sorted pipeline: the narrow pass synchronises on its count of handed-back reads, which would
            // stall the three-stream overlap of a path that is bound by the link anyway)
            mpb_filter_params chunk_prm = *params;
            chunk_prm.flags = (chunk_prm.flags & ~((15u << 8) | (255u << 12))) | MPB_FLAG_NO_NARROW;
            rc = mpb_filter_device(c, d_q, m, row_stride, len ? d_len : nullptr, fixed_len, &chunk_prm, d_ee, d_ns, d_pass, nullptr);
        }
        if (rc) { drain_pipeline(c); return rc; }
        e = hipEventRecord(sl.k_done, c->stream);
        // ---- D2H ----
        if (e == hipSuccess) e = hipStreamWaitEvent(c->out_stream, sl.k_done, 0);
        if (e == hipSuccess) e = hipMemcpyAsync(sl.pin_out, d + L.in_bytes, (size_t)(L.ee + L.ns + align_up(m, 256)),
                                                hipMemcpyDeviceToHost, c->out_stream);
        if (e == hipSuccess) e = hipEventRecord(sl.d2h_done, c->out_stream);
        if (e != hipSuccess) { drain_pipeline(c); return fail(MPB_E_HIP, "host pipeline (D2H): %s", hipGetErrorString(e)); }
        sl.off = off; sl.m = m;
        // With this chunk queued behind the two before it, hand chunk k - 3 to the caller: the host-side copy
        // of its results (and, next trip, the staging copy of chunk k + 1) runs while the copy engines are busy
        // with queued work, and frees the slot the next trip fills.
        if ((rc = retire_slot(c, c->slot[(k + 1) % MPB_HOST_SLOTS], chunk, row_stride, job, &n_pass))) { drain_pipeline(c); return rc; }
        // ... and any younger chunk whose results have already landed (in order), so that little is left for the end
        for (int64_t j = k + 2; j <= k + MPB_HOST_SLOTS - 1; j++) {
            HostSlot &o = c->slot[j % MPB_HOST_SLOTS];
            if (o.off < 0) continue;
            if (hipEventQuery(o.d2h_done) != hipSuccess) { (void)hipGetLastError(); break; }
            if ((rc = retire_slot(c, o, chunk, row_stride, job, &n_pass))) { drain_pipeline(c); return rc; }
        }
    }
    // retire what is still in flight, oldest first
    for (int64_t j = k; j < k + MPB_HOST_SLOTS; j++)
        if ((rc = retire_slot(c, c->slot[j % MPB_HOST_SLOTS], chunk, row_stride, job, &n_pass))) { drain_pipeline(c); return rc; }
    long long ovf = 0;
    int32_t bad = 0, bad255 = 0;
    HIPCHK(hipMemcpyAsync(&ovf, c->ws.ovf_total, sizeof(ovf), hipMemcpyDeviceToHost, c->stream));
    if (len && !poisson) HIPCHK(hipMemcpyAsync(&bad, c->ws.bad_len, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    if (poisson) HIPCHK(hipMemcpyAsync(&bad255, c->ws.ovf_count, sizeof(bad255), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad255) return fail(MPB_E_INVALID, "%d read(s) contain byte 255 ('n'): the Poisson path follows the Python reference, "
                            "which scores lower-case n as a normal base -- pack it as one", bad255);
    if (poisson) ovf = 0;
    if (bad) {
        HIPCHK(hipMemsetAsync(c->ws.bad_len, 0, sizeof(int32_t), c->stream));
        return fail(MPB_E_INVALID, "%d read length(s) outside 0..%d", bad, MPB_MAX_LEN);
    }
    if (counts) { counts->n_reads = n; counts->n_pass = n_pass; counts->n_fail = n - n_pass; counts->n_overflow = ovf; }
    return MPB_OK;
}

int mpb_filter_host(mpb_ctx *c, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len,
                    int32_t fixed_len, const mpb_filter_params *params, double *ee, int32_t *ns,
                    uint8_t *pass, mpb_filter_counts *counts)
{
    CTXCHK(c);
    int rc = check_params(params);
    if (rc) return rc;
    if (n < 0) return fail(MPB_E_INVALID, "n < 0");
    if (row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "row_stride must be a positive multiple of 16");
    if (row_stride > MPB_MAX_STRIDE) return fail(MPB_E_INVALID, "row_stride %lld exceeds %d (reads longer than %d bases are not supported)", (long long)row_stride, MPB_MAX_STRIDE, MPB_MAX_LEN);
    if (n > 0 && (!q || !ee || !ns || !pass)) return fail(MPB_E_INVALID, "NULL host buffer");
    if (!len && (fixed_len < 0 || fixed_len > row_stride)) return fail(MPB_E_INVALID, "fixed_len %d does not fit row_stride %lld", fixed_len, (long long)row_stride);
    if (!len && fixed_len > MPB_MAX_LEN) return fail(MPB_E_INVALID, "reads longer than %d bases are not supported", MPB_MAX_LEN);
    if (len) {
        // a length is never clamped silently: the reference scores every base it is given
        const int64_t lim = row_stride < MPB_MAX_LEN ? row_stride : MPB_MAX_LEN;
        for (int64_t i = 0; i < n; i++)
            if (len[i] < 0 || len[i] > lim)
                return fail(MPB_E_INVALID, len[i] > row_stride || len[i] < 0 ? "read %lld: length %d does not fit the %lld-byte row"
                                                                             : "read %lld: %d bases; reads longer than 65535 bases are not supported (row of %lld bytes)",
                            (long long)i, len[i], (long long)row_stride);
    }
    if (counts) { counts->n_reads = n; counts->n_pass = 0; counts->n_fail = 0; counts->n_overflow = 0; }
    if (n == 0) return MPB_OK;
    if (n <= MPB_SMALL_N && n * row_stride <= (8ll << 20) && row_stride <= MPB_SMALL_MAX_STRIDE && !(params->flags & MPB_FLAG_BATCHED_ONLY)) {
        bool done = false;
        rc = filter_host_small(c, q, n, row_stride, len, fixed_len, params, ee, ns, pass, counts, &done);
        if (rc || done) return rc;
    }
    return filter_host_pipeline(c, q, n, row_stride, len, fixed_len, params, ee, ns, pass, counts, 0);
}

int mpb_host_alloc(mpb_ctx *c, int64_t bytes, void **hptr_out)
{
    CTXCHK(c);
    if (!hptr_out || bytes < 0) return fail(MPB_E_INVALID, "mpb_host_alloc: bad arguments");
    *hptr_out = nullptr;
    HIPCHK(hipHostMalloc(hptr_out, bytes > 0 ? (size_t)bytes : 16, hipHostMallocDefault));
    return MPB_OK;
}

int mpb_host_free(mpb_ctx *c, void *hptr)
{
    CTXCHK(c);
    drain_pipeline(c);
    serve_quiesce(c);
    if (hptr) HIPCHK(hipHostFree(hptr));
    return MPB_OK;
}

// One read of the per-read entries -> a packed row.  A score the byte codes cannot express borrows a code the read does
// not use, from 254 down (the row predictor reads codes as scores, and up there every code means "practically never
// wrong"), and h -- a copy of the context's table -- gets {1-p, p'} of the real score under that code (*priv = true):
// the reference takes any int (moira/bernoullimodule.c:92-108, moira/moira.py:1637-1679) and one read holds few
// distinct values.  poisson: the Python function's rules -- 'n' is a base like any other (moira.py:1660), Q0 is p = 1
// (no clamp inside the function; it too gets a private code), and the second table component is p itself.
int mpbi_pack_one_read(const char *contig, const int32_t *quals, int32_t len, bool poisson, uint8_t *row, int32_t row_bytes,
                       double2 *h, bool *priv)
{
    *priv = false;
    auto direct = [&](int32_t q) { return poisson ? (q >= 1 && q <= 254) : q <= 254; };
    bool used[256] = {};
    for (int32_t i = 0; i < len; i++) {
        const int32_t q = quals[i];
        if (q < 0) return fail(MPB_E_RANGE, "Qualities must have positive values.");
        if (direct(q)) used[q == 0 ? 1 : q] = true; else *priv = true;
    }
    if (*priv) build_lut(h);
    std::unordered_map<int32_t, int> code_of;
    int next = 254;
    for (int32_t i = 0; i < len; i++) {
        const int32_t q = quals[i];
        char base = contig ? contig[i] : 'A';
        if (poisson && base == 'n') base = 'A';
        if (direct(q)) { row[i] = pack_one(base, q); continue; }
        auto it = code_of.find(q);
        if (it == code_of.end()) {
            while (next >= 1 && used[next]) next--;
            if (next < 1) return fail(MPB_E_RANGE, "more than 254 distinct quality values in one read, some outside 1..254");
            used[next] = true;
            lut_entry(q, &h[next]);
            if (poisson) { volatile double p = pow(10, (q / -10.0)); h[next].y = p; }     // (1 - p is 0 at Q0: p' is not p there)
            it = code_of.emplace(q, next).first;
        }
        row[i] = pack_one(base, it->second);
    }
    memset(row + len, 0, (size_t)(row_bytes - len));
    return MPB_OK;
}

// The kernels of one call run on a private copy of the table; the context's own is back in place when the call returns.
struct PrivateTable {
    mpb_ctx *c;
    bool on = false;
    explicit PrivateTable(mpb_ctx *ctx) : c(ctx) {}
    int install(const double2 *h)
    {
        int rc = ensure_workspace(c, 1);
        if (rc) return rc;
        if (!c->d_lut_private) HIPCHK(hipMalloc((void **)&c->d_lut_private, 256 * sizeof(double2)));
        HIPCHK(hipMemcpyAsync(c->d_lut_private, h, 256 * sizeof(double2), hipMemcpyHostToDevice, c->stream));
        HIPCHK(hipStreamSynchronize(c->stream));             // h lives on the caller's stack
        c->ws.lut = c->d_lut_private;
        on = true;
        return MPB_OK;
    }
    ~PrivateTable() { if (on) c->ws.lut = c->d_lut; }
};

// ---- batches whose reads carry scores above 254 (missing 3: the reference takes any int) ----------------
// The byte matrix has 254 score codes.  A batch rarely uses them all, so -- exactly as the per-read entry does for one read --
// the packer gives every distinct out-of-range score of the BATCH a code the batch does not use (from 254 down: the row
// predictor reads codes as scores, and up there every code means "practically never wrong") and reports what each code
// stands for; mpb_filter_host_coded runs the ordinary pipeline on a private copy of the table built from that list.

int mpb_pack_batch_coded(const char *seq_cat, const int32_t *qual_cat, const int64_t *off, int64_t n, int32_t max_len,
                         int64_t row_stride, uint8_t *q_out, int32_t *len_out, int32_t *code_scores)
{
    if (!qual_cat || !off || !q_out || !len_out || !code_scores || n < 0) return fail(MPB_E_INVALID, "bad arguments");
    if (row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "row_stride must be a positive multiple of 16");
    bool used[256] = {};
    std::unordered_map<int32_t, int> code_of;
    std::vector<int32_t> big;                         // distinct out-of-range scores in order of first appearance
    for (int64_t i = 0; i < n; i++) {
        int64_t L = off[i + 1] - off[i];
        if (L < 0) return fail(MPB_E_INVALID, "offsets must not decrease");
        if (max_len > 0 && L > max_len) L = max_len;
        if (L > row_stride) return fail(MPB_E_INVALID, "read %lld (%lld bases) does not fit row_stride %lld", (long long)i, (long long)L, (long long)row_stride);
        if (L > MPB_MAX_LEN) return fail(MPB_E_INVALID, "read %lld has %lld bases: reads longer than %d bases are not supported", (long long)i, (long long)L, MPB_MAX_LEN);
        for (int64_t k = 0; k < L; k++) {
            const int32_t q = qual_cat[off[i] + k];
            if (q < 0) return fail(MPB_E_RANGE, "Qualities must have positive values.");
            if (q <= 254) used[q == 0 ? 1 : q] = true;
            else if (code_of.emplace(q, -1).second) big.push_back(q);
        }
    }
    for (int c = 0; c < 256; c++) code_scores[c] = c;
    int next = 254;
    size_t assigned = 0;
    for (int32_t q : big) {
        while (next >= 1 && used[next]) next--;
        if (next < 1)
            return fail(MPB_E_RANGE, "the batch holds %zu distinct scores above 254 but uses all but %zu of the 254 byte codes: "
                        "split it (or score such reads through mpb_calculate_errors_PB)", big.size(), assigned);
        assigned++;
        used[next] = true;
        code_of[q] = next;
        code_scores[next] = q;
    }
    for (int64_t i = 0; i < n; i++) {
        int64_t L = off[i + 1] - off[i];
        if (max_len > 0 && L > max_len) L = max_len;
        uint8_t *row = q_out + i * row_stride;
        for (int64_t k = 0; k < L; k++) {
            const int32_t q = qual_cat[off[i] + k];
            const char base = seq_cat ? seq_cat[off[i] + k] : 'A';
            row[k] = pack_one(base, q <= 254 ? q : code_of[q]);
        }
        memset(row + L, 0, (size_t)(row_stride - L));
        len_out[i] = (int32_t)L;
    }
    return MPB_OK;
}

int mpb_filter_host_coded(mpb_ctx *c, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len, int32_t fixed_len,
                          const mpb_filter_params *params, const int32_t *code_scores, double *ee, int32_t *ns, uint8_t *pass,
                          mpb_filter_counts *counts)
{
    CTXCHK(c);
    if (!code_scores) return mpb_filter_host(c, q, n, row_stride, len, fixed_len, params, ee, ns, pass, counts);
    double2 h[256];
    build_lut(h);
    bool any = false;
    for (int code = 1; code <= 254; code++) {
        if (code_scores[code] == code) continue;
        if (code_scores[code] < 1) return fail(MPB_E_RANGE, "code %d stands for score %d: scores are >= 1 (Q0 is clamped to 1 at pack time)", code, code_scores[code]);
        lut_entry(code_scores[code], &h[code]);
        any = true;
    }
    PrivateTable guard(c);
    int rc;
    if (any && (rc = guard.install(h)) != MPB_OK) return rc;
    return mpb_filter_host(c, q, n, row_stride, len, fixed_len, params, ee, ns, pass, counts);
}

// argument rules of the per-read entry (moira/bernoullimodule.c:79-90), shared with the broker's client side
int mpbi_check_one_read(const char *contig, const int32_t *contig_quals, int32_t len, double alpha, const void *ee, const void *ns)
{
    if (!ee || !ns) return fail(MPB_E_INVALID, "NULL output");
    if (!(alpha > 0 && alpha < 1)) return fail(MPB_E_INVALID, "Alpha must be between 0 and 1");   // bernoullimodule.c:79-83
    if (len < 0 || (len > 0 && !contig_quals)) return fail(MPB_E_INVALID, "bad arguments");
    if (contig && (int32_t)strlen(contig) != len)                                               // bernoullimodule.c:85-90
        return fail(MPB_E_INVALID, "contig and contig_quals must have the same length");
    if (len > MPB_MAX_LEN) return fail(MPB_E_INVALID, "reads longer than %d bases are not supported", MPB_MAX_LEN);
    return MPB_OK;
}

// one packed row (and, when the read carries scores above 254, its private table h) -> (ee, Ns): the GPU half of the
// per-read entry.  The broker calls it for the reads it cannot put into a micro-batch.
// ---- the per-read entry without a launch per call (round 5) ----------------------------------------------------------
// mpb_calculate_errors_PB from a process of its own (moira.py --processors 1: one call per read) paid a k_small launch per
// call: 26 us of which the kernel is 12.  While such calls keep coming the context keeps k_serve resident with ONE mailbox
// entry (mpb_kernels.hip; the broker's form has one per slot): the call writes row + parameters + door word into pinned
// memory and spins on done[0].  The kernel leaves by itself 100 ms after its launch -- so 100 ms after the last call at
// the latest -- and the next call launches it again.  Anything that frees device or pinned memory (a device-wide wait in the
// runtime) asks it to leave first.  MPB_SERVE=0 keeps the launch per call.
static inline int64_t mono_us()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}

static void serve_quiesce(mpb_ctx *c)
{
    auto &sv = c->serve;
    if (!sv.ok || !sv.running) return;
    __atomic_store_n((uint32_t *)sv.box.stop, 1u, __ATOMIC_SEQ_CST);
    (void)hipStreamSynchronize(sv.stream);
    __atomic_store_n((uint32_t *)sv.box.stop, 0u, __ATOMIC_SEQ_CST);
    sv.running = false;
}

static void serve_free(mpb_ctx *c)
{
    auto &sv = c->serve;
    serve_quiesce(c);
    if (sv.stream) (void)hipStreamDestroy(sv.stream);
    if (sv.pin) (void)hipHostFree(sv.pin);
    if (sv.dev) (void)hipFree(sv.dev);
    sv = mpb_ctx::CtxServe{};
}

static bool serve_init(mpb_ctx *c)
{
    auto &sv = c->serve;
    if (sv.tried) return sv.ok;
    sv.tried = true;
    const char *e = getenv("MPB_SERVE");
    if (e && atoi(e) == 0) return false;
    // pinned: row | parameters | door | done | ee | ns | pass | stop | exited, a 256-byte step each
    const size_t o_q = 0, o_prm = MPB_SERVE_STRIDE, o_door = o_prm + 256, o_done = o_door + 256, o_ee = o_done + 256,
                 o_ns = o_ee + 256, o_pass = o_ns + 256, o_stop = o_pass + 256, o_exited = o_stop + 256, pin_bytes = o_exited + 256;
    const size_t d_stage = 0, d_ns = MPB_SERVE_STRIDE + 256, d_cls = d_ns + 256, d_ident = d_cls + 256, d_gone = d_ident + 256,
                 dev_bytes = d_gone + 256;
    if (hipStreamCreateWithFlags(&sv.stream, hipStreamNonBlocking) != hipSuccess ||
        hipHostMalloc((void **)&sv.pin, pin_bytes, hipHostMallocMapped) != hipSuccess ||
        hipMalloc((void **)&sv.dev, dev_bytes) != hipSuccess || hipMemset(sv.dev, 0, dev_bytes) != hipSuccess) {
        (void)hipGetLastError();
        if (sv.stream) (void)hipStreamDestroy(sv.stream);
        if (sv.pin) (void)hipHostFree(sv.pin);
        if (sv.dev) (void)hipFree(sv.dev);
        sv.stream = nullptr; sv.pin = sv.dev = nullptr;
        return false;                                   // the launch per call remains
    }
    memset(sv.pin, 0, pin_bytes);
    MpbServeBox &x = sv.box;
    x.q = (const uint8_t *)(sv.pin + o_q); x.stride = MPB_SERVE_STRIDE; x.prm = (const MpbServePrm *)(sv.pin + o_prm);
    x.door = (const unsigned long long *)(sv.pin + o_door); x.done = (uint32_t *)(sv.pin + o_done);
    x.ee = (double *)(sv.pin + o_ee); x.ns = (int32_t *)(sv.pin + o_ns); x.pass = (uint8_t *)(sv.pin + o_pass);
    x.stop = (const uint32_t *)(sv.pin + o_stop); x.exited = (uint32_t *)(sv.pin + o_exited);
    x.stage = (uint8_t *)(sv.dev + d_stage); x.ns_dev = (int32_t *)(sv.dev + d_ns); x.cls = (uint8_t *)(sv.dev + d_cls);
    x.ident = (int32_t *)(sv.dev + d_ident); x.gone = (uint32_t *)(sv.dev + d_gone);
    x.q_step = MPB_SERVE_STRIDE; x.prm_step = sizeof(MpbServePrm); x.door_step = 8; x.done_step = 4; x.ee_step = 8; x.ns_step = 4; x.pass_step = 1;
    x.n_ent = 1;
    sv.ok = true;
    return true;
}

static int serve_launch(mpb_ctx *c)
{
    auto &sv = c->serve;
    if (++sv.generation == 0) sv.generation = 1;
    mpb_launch_serve(sv.box, c->d_lut, sv.generation, 100, sv.stream);
    HIPCHK(hipGetLastError());
    sv.running = true;
    return MPB_OK;
}

// one packed read (default table, at most MPB_SERVE_STRIDE - 1 bases) through the resident server; *served = false: not
// taken (no server, or the read missed its row budget there): the caller goes the ordinary way
static int serve_one(mpb_ctx *c, const uint8_t *row, int32_t len, double alpha, double *ee, int32_t *ns, bool *served)
{
    *served = false;
    if (!serve_init(c)) return MPB_OK;
    auto &sv = c->serve;
    const MpbServeBox &x = sv.box;
    if (sv.running && __atomic_load_n(x.exited, __ATOMIC_ACQUIRE) == sv.generation) sv.running = false;
    memcpy((void *)x.q, row, (size_t)((len + 15) & ~15));
    if (alpha != sv.cached_alpha) { mpbi_small_params(alpha, &sv.cached_prm); sv.cached_alpha = alpha; }
    ((MpbServePrm *)x.prm)->p = sv.cached_prm;
    if (++sv.tok == 0) sv.tok = 1;
    __atomic_store_n((unsigned long long *)x.door, ((unsigned long long)(uint32_t)len << 32) | sv.tok, __ATOMIC_RELEASE);
    int rc;
    if (!sv.running && (rc = serve_launch(c))) return rc;
    const int64_t t0 = mono_us();
    int64_t asked = t0;
    for (unsigned spins = 0;; spins++) {
        if (__atomic_load_n(x.done, __ATOMIC_ACQUIRE) == sv.tok) break;
        __builtin_ia32_pause();
        if ((spins & 1023u) != 1023u) continue;
        // a wave that left just before the door word arrived: the launch that follows serves it (it starts from done[0])
        if (__atomic_load_n(x.exited, __ATOMIC_ACQUIRE) == sv.generation) {
            if (__atomic_load_n(x.done, __ATOMIC_ACQUIRE) == sv.tok) break;
            if ((rc = serve_launch(c))) return rc;
        }
        // a GPU that is busy with somebody else's long kernel answers late, as a launch per call would: wait (off the CPU
        // between looks after two milliseconds), and ask the runtime every two seconds -- only a fault ends the wait
        const int64_t now = mono_us();
        if (now - t0 > 2000) usleep(50);
        if (now - asked > 2000000) {
            asked = now;
            const hipError_t q = hipStreamQuery(sv.stream);
            if (q != hipSuccess && q != hipErrorNotReady) {
                sv.running = false;
                return fail(MPB_E_HIP, "the resident per-read kernel failed: %s", hipGetErrorString(q));
            }
        }
    }
    if (*x.pass == 2) return MPB_OK;                    // row budget missed / a wide read: the ordinary path
    *ee = *x.ee;
    *ns = *x.ns;
    *served = true;
    return MPB_OK;
}

int mpbi_run_packed_read(mpb_ctx *c, const uint8_t *row, int32_t len, int32_t stride, const double2 *h, double alpha,
                         double *ee, int32_t *ns)
{
    mpb_filter_params prm;
    prm.alpha = alpha; prm.uncert = 1.0; prm.maxerrors = NAN; prm.ambig_mode = MPB_AMBIG_IGNORE; prm.flags = 0;
    uint8_t pass = 0;
    PrivateTable guard(c);
    int rc;
    if (h && (rc = guard.install(h)) != MPB_OK) return rc;
    return mpb_filter_host(c, row, 1, stride, nullptr, len, &prm, ee, ns, &pass, nullptr);
}

int mpb_calculate_errors_PB(mpb_ctx *c, const char *contig, const int32_t *contig_quals, int32_t len,
                            double alpha, double *ee, int32_t *ns)
{
    CTXCHK(c);
    int rc = mpbi_check_one_read(contig, contig_quals, len, alpha, ee, ns);
    if (rc) return rc;
    const int32_t stride = (int32_t)align_up(len > 0 ? len : 1, 16);
    std::vector<uint8_t> row((size_t)stride);
    bool priv = false;
    double2 h[256];
    if ((rc = mpbi_pack_one_read(contig, contig_quals, len, false, row.data(), stride, h, &priv))) return rc;
    if (!priv && len <= MPB_SERVE_STRIDE - 1 && !c->timing) {
        bool served = false;
        if ((rc = serve_one(c, row.data(), len, alpha, ee, ns, &served)) || served) return rc;
    }
    return mpbi_run_packed_read(c, row.data(), len, stride, priv ? h : nullptr, alpha, ee, ns);
}

// One micro-batch of the broker: m packed rows that already lie in device memory -> one k_small launch on stream s
// (one read per wave), results into device arrays.  Nothing here synchronises; a read that misses its row budget comes
// back with pass == 2 and the broker re-runs it alone.  cls / ident are the launch's own scratch (m bytes / m int32),
// so that several micro-batches can be in flight on different streams.
int mpbi_small_async(mpb_ctx *c, const uint8_t *d_q, int64_t m, int64_t stride, const int32_t *d_len, double alpha,
                     double *d_ee, int32_t *d_ns, uint8_t *d_pass, uint8_t *d_cls, int32_t *d_ident, hipStream_t s,
                     const MpbSmallHost *host)
{
    mpb_filter_params prm;
    prm.alpha = alpha; prm.uncert = 1.0; prm.maxerrors = NAN; prm.ambig_mode = MPB_AMBIG_IGNORE; prm.flags = 0;
    const int32_t max_len = (int32_t)(stride < MPB_MAX_LEN ? stride : MPB_MAX_LEN);
    const MpbDevParams dp = make_dev_params(&prm, 0, max_len);
    int rc = ensure_workspace(c, m);               // (the broker sizes it once, before anything is in flight)
    if (rc) return rc;
    MpbWorkspace ws = c->ws;                       // only lut / cls / perm are read by the launch
    ws.lut = c->d_lut;
    ws.cls = d_cls;
    ws.perm = d_ident;
    mpb_launch_small(d_q, m, stride, d_len, dp, ws, d_ns, d_ee, d_pass, s, host);
    HIPCHK(hipGetLastError());
    return MPB_OK;
}

void mpbi_small_params(double alpha, MpbDevParams *out)
{
    mpb_filter_params prm;
    prm.alpha = alpha; prm.uncert = 1.0; prm.maxerrors = NAN; prm.ambig_mode = MPB_AMBIG_IGNORE; prm.flags = 0;
    *out = make_dev_params(&prm, 0, MPB_SERVE_STRIDE);
}

int mpbi_serve_launch(mpb_ctx *c, const MpbServeBox *box, uint32_t generation, uint32_t lifetime_ms, hipStream_t s)
{
    mpb_launch_serve(*box, c->d_lut, generation, lifetime_ms, s);
    HIPCHK(hipGetLastError());
    return MPB_OK;
}

int mpbi_ctx_device(const mpb_ctx *c) { return c ? c->device : -1; }

int mpbi_fail(int code, const char *msg) { return fail(code, "%s", msg); }

int mpb_decode_ascii_device(mpb_ctx *c, const uint8_t *d_seq, const uint8_t *d_qual, int64_t n, int64_t row_stride,
                            const int32_t *d_len, int32_t fixed_len, int32_t fastq_offset, uint8_t *d_out, int32_t *d_err)
{
    CTXCHK(c);
    if (n < 0 || row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "bad matrix shape");
    if ((((uintptr_t)d_seq | (uintptr_t)d_qual | (uintptr_t)d_out) & 15) != 0) return fail(MPB_E_INVALID, "matrices must be 16-byte aligned");
    if (!d_len && (fixed_len < 0 || fixed_len > row_stride)) return fail(MPB_E_INVALID, "fixed_len does not fit row_stride");
    if (n == 0) return MPB_OK;
    if (!d_seq || !d_qual || !d_out) return fail(MPB_E_INVALID, "NULL device buffer");
    if (n * (row_stride / 16) / 256 > 0x7fffffffll) return fail(MPB_E_INVALID, "too large for one launch; split it");
    mpb_launch_decode(d_seq, d_qual, n, row_stride, d_len, fixed_len, fastq_offset, d_out, d_err, c->stream);
    HIPCHK(hipGetLastError());
    return MPB_OK;
}

// ---- Poisson approximation (SURVEY f-3) ------------------------------------------------------

int mpb_poisson_lambda_device(mpb_ctx *c, const uint8_t *d_q, int64_t n, int64_t row_stride,
                              const int32_t *d_len, int32_t fixed_len, double *d_lambda, int32_t *d_ns)
{
    CTXCHK(c);
    if (n < 0 || row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "bad matrix shape");
    if (row_stride > MPB_LAMBDA_MAX_STRIDE) return fail(MPB_E_INVALID, "row_stride %lld exceeds %d", (long long)row_stride, MPB_LAMBDA_MAX_STRIDE);
    if (((uintptr_t)d_q & 15) != 0) return fail(MPB_E_INVALID, "quality matrix must be 16-byte aligned");
    if (!d_len && (fixed_len < 0 || fixed_len > row_stride)) return fail(MPB_E_INVALID, "fixed_len does not fit row_stride");
    if (n == 0) return MPB_OK;
    if (!d_q || !d_lambda || !d_ns) return fail(MPB_E_INVALID, "NULL device buffer");
    int rc = ensure_workspace(c, 1);
    if (rc) return rc;
    HIPCHK(hipMemsetAsync(c->ws.ovf_count, 0, sizeof(int32_t), c->stream));
    { Span t(c, MPB_K_LAMBDA);
      mpb_launch_lambda(d_q, n, row_stride, d_len, fixed_len, c->ws.lut, d_lambda, d_ns, c->ws.ovf_count, c->stream); }
    HIPCHK(hipGetLastError());
    int32_t bad = 0;
    HIPCHK(hipMemcpyAsync(&bad, c->ws.ovf_count, sizeof(bad), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    if (bad) return fail(MPB_E_INVALID, "%d read(s) contain byte 255 ('n'): the Poisson path follows the Python reference, "
                         "which scores lower-case n as a normal base -- pack it as one", bad);
    return MPB_OK;
}

// float(math.factorial(j)) for j = 0..170: exact up to 22!, correctly rounded above (built from
// the exact integer with round-half-even on first use).
struct FactorialTable {
    double tab[171];
    FactorialTable()
    {
        // exact big-integer product in base 2^32, then correctly rounded conversion
        std::vector<uint32_t> big(1, 1u);
        tab[0] = 1.0;
        for (int j = 1; j <= 170; j++) {
            uint64_t carry = 0;
            for (size_t k = 0; k < big.size(); k++) {
                uint64_t v = (uint64_t)big[k] * (uint64_t)j + carry;
                big[k] = (uint32_t)v; carry = v >> 32;
            }
            if (carry) big.push_back((uint32_t)carry);
            // top 64+ bits -> double with round-half-even
            int top = (int)big.size() - 1;
            int hb = 31; while (!((big[top] >> hb) & 1u)) hb--;
            const long nbits = (long)top * 32 + hb + 1;
            auto bit = [&](long pos) -> int { return pos < 0 ? 0 : (int)((big[pos / 32] >> (pos % 32)) & 1u); };
            uint64_t mant = 0;
            for (int k = 0; k < 53; k++) mant = (mant << 1) | (uint64_t)bit(nbits - 1 - k);
            if (nbits > 53) {
                const int guard = bit(nbits - 54);
                bool sticky = false;
                for (long pos = nbits - 55; pos >= 0 && !sticky; pos--) sticky = bit(pos) != 0;
                if (guard && (sticky || (mant & 1u))) mant++;
            }
            tab[j] = ldexp((double)mant, nbits > 53 ? (int)(nbits - 53) : 0);
            if (nbits <= 53) tab[j] = (double)mant / ldexp(1.0, (int)(53 - nbits));
        }
    }
};

static const double *factorial_table()
{
    // function-local static object: initialised once, thread-safe (C++11 magic static) -- contexts on different
    // threads may reach the Poisson path for the first time concurrently
    static const FactorialTable t;
    return t.tab;
}

// ref: moira/moira.py:1666-1679 -- one read
static double poisson_tail(double lam, double alpha, const double *fact)
{
    double acc_prev = 0, acc = 0;
    int j = 0;
    const double em = exp(-lam);                                  // the reference re-evaluates it per term: same value
    for (;;) {
        if (j > 170) return NAN;                                  // Python: int too large to convert to float
        const double pw = pow(lam, (double)j);
        if (std::isinf(pw)) return NAN;                           // Python: OverflowError from float pow
        const double prob = (em * pw) / fact[j];
        acc_prev = acc;
        acc = acc_prev + prob;
        if (acc > (1 - alpha)) break;
        j++;
    }
    double r = (j - 1) + ((j - (j - 1)) * ((1 - alpha) - acc_prev) / (acc - acc_prev));
    if (r < 0) r = 0;
    return r;
}

int mpb_poisson_finish_host(const double *lambda, const int32_t *ns, const int32_t *len, int32_t fixed_len,
                            int64_t n, const mpb_filter_params *p, double *ee, uint8_t *pass)
{
    int rc = check_params(p);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!lambda || !ns || !ee || !pass))) return fail(MPB_E_INVALID, "bad arguments");
    const double *fact = factorial_table();
    const bool has_me = p->maxerrors == p->maxerrors;
    auto run = [=](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; i++) {
            double e = poisson_tail(lambda[i], p->alpha, fact);
            if (p->ambig_mode == MPB_AMBIG_TREAT_AS_ERRORS) e = e + ns[i];          // moira.py:827-828
            if (p->flags & MPB_FLAG_ROUND) e = floor(e);                              // moira.py:830-831
            const int li = len ? len[i] : fixed_len;
            bool keep;
            if (p->ambig_mode == MPB_AMBIG_DISALLOW && ns[i] > 0) keep = false;       // moira.py:911
            else if (has_me) keep = e <= p->maxerrors;
            else keep = e <= li * p->uncert;
            ee[i] = e;
            pass[i] = keep ? 1 : 0;
        }
    };
    // the scalar tail (one libm pow per CDF term) is what a GPU-fed Poisson run waits for: reads are independent,
    // so large batches are split over the CPUs this process is granted (each read's arithmetic is unchanged)
    const int threads = n >= 16384 ? 2 * staging_threads() : 1;
    if (threads <= 1) { run(0, n); return MPB_OK; }
    std::vector<std::thread> th;
    const int64_t per = (n + threads - 1) / threads;
    int64_t taken = per < n ? per : n;                 // helpers take [per, taken); the rest is done here
    try {
        th.reserve((size_t)threads);
        for (int t = 1; t < threads; t++) {
            const int64_t lo = per * t, hi = lo + per < n ? lo + per : n;
            if (lo >= n) break;
            th.emplace_back(run, lo, hi);
            taken = hi;
        }
    } catch (...) {
        // thread creation failed: never an exception across the C ABI, the remainder runs on this thread
    }
    run(0, per < n ? per : n);
    if (taken < n) run(taken, n);
    for (auto &t : th) t.join();
    return MPB_OK;
}

int mpb_filter_poisson_host(mpb_ctx *c, const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len,
                            int32_t fixed_len, const mpb_filter_params *params, double *ee, int32_t *ns,
                            uint8_t *pass, mpb_filter_counts *counts)
{
    CTXCHK(c);
    int rc = check_params(params);
    if (rc) return rc;
    if (n < 0 || row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "bad matrix shape");
    if (row_stride > MPB_LAMBDA_MAX_STRIDE) return fail(MPB_E_INVALID, "row_stride %lld exceeds %d", (long long)row_stride, MPB_LAMBDA_MAX_STRIDE);
    if (n > 0 && (!q || !ee || !ns || !pass)) return fail(MPB_E_INVALID, "NULL host buffer");
    if (!len && (fixed_len < 0 || fixed_len > row_stride)) return fail(MPB_E_INVALID, "fixed_len does not fit row_stride");
    if (len)
        for (int64_t i = 0; i < n; i++)
            if (len[i] < 0 || len[i] > row_stride)
                return fail(MPB_E_INVALID, "read %lld: length %d does not fit the %lld-byte row", (long long)i, len[i], (long long)row_stride);
    if (counts) { counts->n_reads = n; counts->n_pass = 0; counts->n_fail = 0; counts->n_overflow = 0; }
    if (n == 0) return MPB_OK;
    // the same four-slot pipeline as mpb_filter_host: H2D of chunk k+1 | k_lambda of chunk k | D2H of chunk k-1, and the
    // scalar tail of a chunk runs on the host while the GPU is busy with the chunks after it
    return filter_host_pipeline(c, q, n, row_stride, len, fixed_len, params, ee, ns, pass, counts, 1);
}

// One read, the twin of moira.py's calculate_errors_poisson(sequence, quals, alpha) -> (expected_errors, Ns)
// (moira/moira.py:1637-1679): any non-negative int is a score (see pack_one_read), 'n' is a base, Q0 is p = 1.
// ee is NaN where the Python function raises OverflowError (Lambda ** j or the factorial leave the float range).
int mpb_calculate_errors_poisson(mpb_ctx *c, const char *sequence, const int32_t *quals, int32_t len, double alpha,
                                 double *ee, int32_t *ns)
{
    CTXCHK(c);
    if (!ee || !ns) return fail(MPB_E_INVALID, "NULL output");
    if (!(alpha > 0 && alpha < 1)) return fail(MPB_E_INVALID, "Alpha must be between 0 and 1");
    if (len < 0 || (len > 0 && !quals)) return fail(MPB_E_INVALID, "bad arguments");
    if (sequence && (int32_t)strlen(sequence) != len) return fail(MPB_E_INVALID, "sequence and quals must have the same length");
    const int32_t stride = (int32_t)align_up(len > 0 ? len : 1, 16);
    std::vector<uint8_t> row((size_t)stride);
    mpb_filter_params prm;
    prm.alpha = alpha; prm.uncert = 1.0; prm.maxerrors = NAN; prm.ambig_mode = MPB_AMBIG_IGNORE; prm.flags = 0;
    uint8_t pass = 0;
    bool priv = false;
    double2 h[256];
    int rc = mpbi_pack_one_read(sequence, quals, len, true, row.data(), stride, h, &priv);
    if (rc) return rc;
    PrivateTable guard(c);
    if (priv && (rc = guard.install(h)) != MPB_OK) return rc;
    return mpb_filter_poisson_host(c, row.data(), 1, stride, nullptr, len, &prm, ee, ns, &pass, nullptr);
}

// ---- one host process, several GPUs ("one host thread (or process) + one HIP stream per device") ----
//
// The batch is cut into n_ctx contiguous, balanced shards in read order (shard r = reads [r*n/W + min(r, n%W), ...), the
// same bounds as moira_amd/shard.py:shard_bounds); one host thread per context runs the ordinary host pipeline on its
// shard and writes straight into the caller's arrays at the shard's offset, so the results come back gathered in read
// order with no copy and no exchange step.  Every context keeps its own streams, slots and workspace (contexts are
// independent: tests/test_gpu_parity.py::test_two_contexts_from_two_threads); a host-fed caller thereby drives one
// PCIe link per GPU from one process -- what moira's `Pool(args.processors)` (moira/moira.py:398-399) was for.
// ---- NUMA placement of a shard's host thread (8 x 55 GB/s out of one socket does not scale) ------------------
// The PCI device of a GPU says which NUMA node it hangs off (/sys/bus/pci/devices/<id>/numa_node) and the node says which CPUs
// are its own (/sys/devices/system/node/node<k>/cpulist).  A shard thread of mpb_filter_host_multi restricts itself to those
// CPUs before it starts its pipeline: the pinned staging blocks of its slots are then allocated and first touched from that
// node, and the threads that copy a pageable input into them inherit the mask.  `sysfs_root` ("" = the real one) exists for
// the parser's test.  node -1 (no NUMA information, one node, a container that hides it) leaves the thread where it is.
static bool read_small_file(const char *path, char *buf, size_t len)
{
    FILE *f = fopen(path, "r");
    if (!f) return false;
    const size_t k = fread(buf, 1, len - 1, f);
    fclose(f);
    buf[k] = 0;
    return k > 0;
}

// "0-7,16-23,40" -> cpu_set; false on anything else
static bool parse_cpulist(const char *s, cpu_set_t *set, int *count)
{
    CPU_ZERO(set);
    int n = 0;
    const char *p = s;
    while (*p && *p != '\n') {
        char *end = nullptr;
        const long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') {
            b = strtol(p + 1, &end, 10);
            if (end == p + 1 || b < a) return false;
            p = end;
        }
        for (long c = a; c <= b; c++)
            if (c < CPU_SETSIZE) { CPU_SET((int)c, set); n++; }
        if (*p == ',') p++;
        else if (*p && *p != '\n') return false;
    }
    if (count) *count = n;
    return n > 0;
}

int mpb_numa_cpulist_for_pci(const char *sysfs_root, const char *pci_bus_id, int32_t *node_out, char *cpulist_out, int32_t cpulist_len)
{
    if (!pci_bus_id || !node_out) return fail(MPB_E_INVALID, "mpb_numa_cpulist_for_pci: NULL argument");
    const char *root = sysfs_root ? sysfs_root : "";
    char path[512], buf[4096], id[64];
    // HIP prints the bus id in upper-case hex ("0000:C1:00.0"); sysfs names are lower-case
    size_t k = 0;
    for (; pci_bus_id[k] && k + 1 < sizeof(id); k++) id[k] = (char)tolower((unsigned char)pci_bus_id[k]);
    id[k] = 0;
    *node_out = -1;
    if (cpulist_out && cpulist_len > 0) cpulist_out[0] = 0;
    snprintf(path, sizeof(path), "%s/sys/bus/pci/devices/%s/numa_node", root, id);
    if (!read_small_file(path, buf, sizeof(buf))) return MPB_OK;          // no such file: no NUMA information
    char *end = nullptr;
    const long node = strtol(buf, &end, 10);
    if (end == buf || node < 0) return MPB_OK;                               // "-1": the platform does not say
    snprintf(path, sizeof(path), "%s/sys/devices/system/node/node%ld/cpulist", root, node);
    if (!read_small_file(path, buf, sizeof(buf))) return MPB_OK;
    cpu_set_t set;
    int cnt = 0;
    if (!parse_cpulist(buf, &set, &cnt)) return fail(MPB_E_INVALID, "cannot parse %s: '%s'", path, buf);
    *node_out = (int32_t)node;
    if (cpulist_out && cpulist_len > 0) {
        snprintf(cpulist_out, (size_t)cpulist_len, "%s", buf);
        for (char *q = cpulist_out; *q; q++) if (*q == '\n') *q = 0;
    }
    return MPB_OK;
}

// restrict the calling thread to the CPUs of `c`'s GPU's NUMA node that this process may use; returns the node or -1 (not moved)
static int pin_thread_near_device(const mpb_ctx *c)
{
    if (getenv("MOIRA_PB_NO_NUMA")) return -1;
    char bus[64] = "";
    if (hipDeviceGetPCIBusId(bus, (int)sizeof(bus), c->device) != hipSuccess) return -1;
    int32_t node = -1;
    char list[4096];
    if (mpb_numa_cpulist_for_pci("", bus, &node, list, (int32_t)sizeof(list)) != MPB_OK || node < 0) return -1;
    cpu_set_t near, allowed, both;
    if (!parse_cpulist(list, &near, nullptr)) return -1;
    if (sched_getaffinity(0, sizeof(allowed), &allowed) != 0) return -1;
    CPU_AND(&both, &near, &allowed);
    if (CPU_COUNT(&both) == 0) return -1;                                    // the cgroup grants nothing on that node: stay
    if (sched_setaffinity(0, sizeof(both), &both) != 0) return -1;
    return node;
}

int mpb_shard_bounds(int64_t n, int32_t world, int32_t rank, int64_t *lo, int64_t *hi)
{
    if (n < 0 || world < 1 || rank < 0 || rank >= world || !lo || !hi) return fail(MPB_E_INVALID, "mpb_shard_bounds: bad arguments");
    const int64_t base = n / world, rem = n % world;
    *lo = rank * base + (rank < rem ? rank : rem);
    *hi = *lo + base + (rank < rem ? 1 : 0);
    return MPB_OK;
}

int mpb_filter_host_multi(mpb_ctx *const *ctxs, int32_t n_ctx, const uint8_t *q, int64_t n, int64_t row_stride,
                          const int32_t *len, int32_t fixed_len, const mpb_filter_params *params, double *ee,
                          int32_t *ns, uint8_t *pass, mpb_filter_counts *counts, int32_t poisson)
{
    if (!ctxs || n_ctx < 1) return fail(MPB_E_INVALID, "mpb_filter_host_multi: no context");
    for (int k = 0; k < n_ctx; k++) {
        if (!ctxs[k]) return fail(MPB_E_INVALID, "mpb_filter_host_multi: context %d is NULL", k);
        for (int j = 0; j < k; j++)
            if (ctxs[j] == ctxs[k]) return fail(MPB_E_INVALID, "mpb_filter_host_multi: context %d is listed twice (calls on one context must not overlap)", k);
    }
    if (n < 0) return fail(MPB_E_INVALID, "n < 0");
    if (counts) { counts->n_reads = n; counts->n_pass = 0; counts->n_fail = 0; counts->n_overflow = 0; }
    struct Shard { int64_t lo = 0, hi = 0; int rc = MPB_OK; mpb_filter_counts c{}; char err[sizeof(g_err)] = ""; };
    std::vector<Shard> sh((size_t)n_ctx);
    for (int r = 0; r < n_ctx; r++) (void)mpb_shard_bounds(n, n_ctx, r, &sh[(size_t)r].lo, &sh[(size_t)r].hi);
    auto work = [&](int r) {
        Shard &S = sh[(size_t)r];
        const int64_t m = S.hi - S.lo;
        // the shard's thread (and the copy threads it starts, and the pinned blocks it first touches) next to its GPU; the
        // calling thread (shard 0, and shards that got no thread) gets its own mask back afterwards
        cpu_set_t before;
        const bool have_before = n_ctx > 1 && sched_getaffinity(0, sizeof(before), &before) == 0;
        if (n_ctx > 1) (void)pin_thread_near_device(ctxs[r]);
        const uint8_t *qs = q ? q + S.lo * row_stride : q;
        const int32_t *ls = len ? len + S.lo : nullptr;
        if (poisson)
            S.rc = mpb_filter_poisson_host(ctxs[r], qs, m, row_stride, ls, fixed_len, params, ee ? ee + S.lo : ee,
                                           ns ? ns + S.lo : ns, pass ? pass + S.lo : pass, &S.c);
        else
            S.rc = mpb_filter_host(ctxs[r], qs, m, row_stride, ls, fixed_len, params, ee ? ee + S.lo : ee,
                                   ns ? ns + S.lo : ns, pass ? pass + S.lo : pass, &S.c);
        if (S.rc != MPB_OK) snprintf(S.err, sizeof(S.err), "%s", g_err);      // g_err is thread-local: carry it out
        if (have_before) (void)sched_setaffinity(0, sizeof(before), &before);
    };
    std::vector<std::thread> th;
    int started = 1;                                   // shard 0 runs on the calling thread
    try {
        th.reserve((size_t)n_ctx);
        for (int r = 1; r < n_ctx; r++) { th.emplace_back(work, r); started = r + 1; }
    } catch (...) {
        // no thread to be had: the shards that got none run on this thread, one after the other
    }
    work(0);
    for (int r = started; r < n_ctx; r++) work(r);
    for (auto &t : th) t.join();
    for (int r = 0; r < n_ctx; r++) {
        const Shard &S = sh[(size_t)r];
        if (S.rc != MPB_OK) return fail(S.rc, "shard %d of %d (reads %lld..%lld): %s", r, n_ctx, (long long)S.lo, (long long)S.hi - 1, S.err);
        if (counts) { counts->n_pass += S.c.n_pass; counts->n_fail += S.c.n_fail; counts->n_overflow += S.c.n_overflow; }
    }
    return MPB_OK;
}

int mpb_synth_fill_device(mpb_ctx *c, uint8_t *d_q, int64_t n, int64_t row_stride, int32_t fixed_len,
                          int32_t min_len, int32_t max_len, int32_t *d_len, uint64_t seed, int64_t first_read)
{
    return mpb_synth_fill_device_profile(c, d_q, n, row_stride, fixed_len, min_len, max_len, d_len, seed, first_read, 0);
}

int mpb_synth_fill_device_profile(mpb_ctx *c, uint8_t *d_q, int64_t n, int64_t row_stride, int32_t fixed_len,
                                  int32_t min_len, int32_t max_len, int32_t *d_len, uint64_t seed, int64_t first_read,
                                  int32_t profile)
{
    CTXCHK(c);
    if (profile != 0 && profile != 1) return fail(MPB_E_INVALID, "unknown synthetic profile %d", profile);
    if (n < 0 || row_stride <= 0 || row_stride % 16 != 0) return fail(MPB_E_INVALID, "bad matrix shape");
    if (fixed_len > 0) { if (fixed_len > row_stride) return fail(MPB_E_INVALID, "fixed_len exceeds row_stride"); }
    else if (min_len < 1 || max_len < min_len || max_len > row_stride || !d_len)
        return fail(MPB_E_INVALID, "ragged fill needs 1 <= min_len <= max_len <= row_stride and d_len");
    if (n == 0) return MPB_OK;
    if (n * (row_stride / 16) / 256 > 0x7fffffffll) return fail(MPB_E_INVALID, "fill too large for one launch; split it");
    mpb_launch_synth(d_q, n, row_stride, fixed_len, min_len, max_len, d_len, seed, first_read, c->stream, profile);
    HIPCHK(hipGetLastError());
    return MPB_OK;
}

}  // extern "C"
