// mpb_broker.cpp -- one GPU-owning process serves the per-read calls of many worker processes.
//
// Reference shape served: moira/moira.py:398-399,431-454 -- `Pool(args.processors)` worker processes, each calling
// bernoulli.calculate_errors_PB(contig, quals, alpha) once per read (moira/moira.py:817) and blocking for the answer.
// If every worker opens its own GPU context those one-read launches time-share the card (2.4 x 10^4 calls/s in all,
// whatever P is: profiles/r03_per_read_concurrency.txt).  Here the workers never touch the GPU: a call packs its read into
// a slot of a shared-memory segment and waits; the broker process gathers whatever is pending -- at most one read per
// worker, they block -- into ONE launch of the one-read-per-wave kernel (k_small), several such micro-batches in flight on
// streams of their own, and hands the results back through the slots.  Since round 5 the default is no launch per call at
// all: a resident kernel (k_serve), a wave per slot polling a mailbox entry in pinned host memory (`Server` below).
//
// Segment (POSIX shared memory, name "/moira_pb_<name>"): a header and n_slots slots of fixed size.  A client owns one
// slot for as long as it is attached (claimed by a compare-and-swap of its pid); a slot whose owner has died is reclaimed
// by the broker.  Slot state machine (one 32-bit word, also the futex the client sleeps on):
//     IDLE -> (client packs the read) SUBMITTED -> (broker) RUNNING -> DONE -> (client copies the result) IDLE
// Waiting is spin-then-futex on both sides, so that in steady state nobody makes a system call.
//
// Results are those of mpb_calculate_errors_PB bit for bit: same packer (client side, host only), same kernel, same
// fallbacks (a read that misses its predicted row budget, or carries scores above 254, is run alone by the broker through
// the ordinary per-read path).

#include "../../include/moira_pb.h"
#include "mpb_internal.h"
#include "mpb_host_internal.h"

#include <atomic>
#include <cerrno>
#include <cmath>
#include <csignal>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fcntl.h>
#include <linux/futex.h>
#include <new>
#include <sys/file.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <sys/syscall.h>
#include <unistd.h>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t BRK_MAGIC = 0x4d504252u;        // "MPBR"
constexpr uint32_t BRK_VERSION = 3;
constexpr int BRK_LANES = 4;                        // micro-batches in flight
constexpr int BRK_MAX_SLOTS = 256;

enum : uint32_t { ST_IDLE = 0, ST_SUBMITTED = 1, ST_RUNNING = 2, ST_DONE = 3 };
enum : int32_t { BS_STARTING = 0, BS_SERVING = 1, BS_EXITING = 2 };

struct alignas(64) BrkHeader {
    uint32_t magic, version;
    int32_t n_slots, slot_bytes;
    std::atomic<int32_t> state;
    std::atomic<int32_t> pid;
    std::atomic<int32_t> device;
    std::atomic<int32_t> stop;                      // set by mpb_broker_shutdown
    alignas(64) std::atomic<uint32_t> submit_seq;   // futex word: bumped by every submit
    std::atomic<int32_t> sleeping;                  // the broker is (about to be) asleep on submit_seq
    alignas(64) std::atomic<int64_t> served;
    std::atomic<int64_t> batches, solo;
    std::atomic<int64_t> heartbeat_ms;
    // direct serving (round 5): the resident kernel reads the requests and writes the results IN the slots (the segment is
    // registered with the runtime): a call does not pass through the broker thread at all.  `direct`: the clients may post
    // door words; `server_up`: a launch of the kernel is out (a client that finds it down kicks the broker);
    // `direct_seq`: bumped by every direct call (how the broker knows that calls keep coming).
    alignas(64) std::atomic<int32_t> direct;
    std::atomic<int32_t> server_up;
    alignas(64) std::atomic<uint32_t> direct_seq;
};

// One cache line per direction: the client spins on the first (state), the request and the result each have a line of their
// own, so that the broker's result stores do not fight the spinning reader for the line, and one store of `state` hands over.
struct alignas(64) BrkSlot {
    std::atomic<int32_t> owner;                     // pid of the attached client, 0 = free
    std::atomic<uint32_t> state;                    // ST_*, futex word of the client
    std::atomic<int32_t> waiting;                   // the client sleeps on `state`
    alignas(64) int32_t len;                        // request (written by the client before SUBMITTED)
    int32_t priv;
    double alpha;
    alignas(64) int32_t rc;                         // result (written by the broker before DONE)
    int32_t ns;
    double ee;
    char err[160];
    // direct serving: the client's line (door word, its count of direct calls), the kernel's line (token served, results),
    // the request's parameters
    alignas(64) unsigned long long door;            // {length << 32 | token}, written by the client after row + parameters
    int64_t n_direct;                               // direct calls of this slot's owners (summed by mpb_broker_stats)
    alignas(64) uint32_t d_done;                    // written by the kernel: the token served ...
    int32_t d_ns;
    double d_ee;
    uint8_t d_pass;                                 // ... 2: row budget missed (or a wide read): the client submits it to the broker
    alignas(64) MpbServePrm d_prm;
    // then: uint8_t row[MPB_MAX_STRIDE]; double2 lut[256] (only read when priv != 0)
};

constexpr size_t SLOT_ROW_OFF = (sizeof(BrkSlot) + 63) & ~(size_t)63;
constexpr size_t SLOT_LUT_OFF = SLOT_ROW_OFF + MPB_MAX_STRIDE;
constexpr size_t SLOT_BYTES = SLOT_LUT_OFF + 256 * sizeof(double2);

inline long futex(void *addr, int op, uint32_t val, const timespec *ts)
{
    return syscall(SYS_futex, addr, op, val, ts, nullptr, 0);
}
inline void futex_wait(void *addr, uint32_t val, int ms)
{
    timespec ts{ms / 1000, (long)(ms % 1000) * 1000000L};
    futex(addr, FUTEX_WAIT, val, &ts);
}
inline void futex_wait_us(void *addr, uint32_t val, int64_t us)
{
    timespec ts{(time_t)(us / 1000000), (long)(us % 1000000) * 1000L};
    futex(addr, FUTEX_WAIT, val, &ts);
}
inline void futex_wake(void *addr, int n) { futex(addr, FUTEX_WAKE, (uint32_t)n, nullptr); }

inline int64_t now_ms()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000 + ts.tv_nsec / 1000000;
}
inline int64_t now_us()
{
    timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (int64_t)ts.tv_sec * 1000000 + ts.tv_nsec / 1000;
}
inline void cpu_relax() { __builtin_ia32_pause(); }

inline bool pid_alive(int pid) { return pid > 0 && (kill(pid, 0) == 0 || errno == EPERM); }

int shm_path(const char *name, char *out, size_t cap)
{
    if (!name || !*name || strlen(name) > 100) return mpbi_fail(MPB_E_INVALID, "broker name must be 1..100 characters");
    for (const char *p = name; *p; p++)
        if (!((*p >= 'a' && *p <= 'z') || (*p >= 'A' && *p <= 'Z') || (*p >= '0' && *p <= '9') || *p == '_' || *p == '-' || *p == '.'))
            return mpbi_fail(MPB_E_INVALID, "broker name may hold letters, digits, '_', '-' and '.' only");
    snprintf(out, cap, "/moira_pb_%s", name);
    return MPB_OK;
}

struct Mapping {
    void *base = nullptr;
    size_t bytes = 0;
    BrkHeader *hdr() const { return (BrkHeader *)base; }
    BrkSlot *slot(int i) const { return (BrkSlot *)((char *)base + sizeof(BrkHeader) + (size_t)i * SLOT_BYTES); }
    static uint8_t *row(BrkSlot *s) { return (uint8_t *)s + SLOT_ROW_OFF; }
    static double2 *lut(BrkSlot *s) { return (double2 *)((char *)s + SLOT_LUT_OFF); }
    void unmap() { if (base) munmap(base, bytes); base = nullptr; bytes = 0; }
};

int map_existing(const char *name, Mapping *m)
{
    char path[128];
    int rc = shm_path(name, path, sizeof(path));
    if (rc) return rc;
    const int fd = shm_open(path, O_RDWR, 0600);
    if (fd < 0) return mpbi_fail(MPB_E_INVALID, "no broker segment of this name");
    struct stat st;
    if (fstat(fd, &st) != 0 || (size_t)st.st_size < sizeof(BrkHeader)) { close(fd); return mpbi_fail(MPB_E_INVALID, "broker segment is not initialised yet"); }
    void *p = mmap(nullptr, (size_t)st.st_size, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    close(fd);
    if (p == MAP_FAILED) return mpbi_fail(MPB_E_NOMEM, "mmap of the broker segment failed");
    m->base = p;
    m->bytes = (size_t)st.st_size;
    const BrkHeader *h = m->hdr();
    if (h->magic != BRK_MAGIC || h->version != BRK_VERSION || h->slot_bytes != (int32_t)SLOT_BYTES ||
        m->bytes < sizeof(BrkHeader) + (size_t)h->n_slots * SLOT_BYTES) {
        m->unmap();
        return mpbi_fail(MPB_E_INVALID, "broker segment has another layout (another library version?)");
    }
    return MPB_OK;
}

// ---- broker side -------------------------------------------------------------------------------------------------

#define BHIP(expr)                                                                         \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess) {                                                            \
            char b_[256];                                                                  \
            snprintf(b_, sizeof(b_), "%s failed: %s", #expr, hipGetErrorString(e_));       \
            return mpbi_fail(e_ == hipErrorOutOfMemory ? MPB_E_NOMEM : MPB_E_HIP, b_);     \
        }                                                                                  \
    } while (0)

// one micro-batch in flight: pinned input block (len | q), device block, pinned output block (ee | ns | pass)
struct Lane {
    hipStream_t stream = nullptr;
    char *pin_in = nullptr, *pin_out = nullptr, *dev = nullptr;
    size_t in_cap = 0, out_cap = 0, dev_cap = 0;
    std::atomic<int> busy{0};   // 0: free (the launch thread may fill it), 1: a launch is out (the retire thread's, when there is one)
    int m = 0;
    int64_t stride = 0;
    int slots[BRK_MAX_SLOTS];
    uint32_t token = 0;         // completion token of the launch in flight (zero-copy lanes: k_small reports through pin_out)
    int reported = 0;           // reads of it whose flag has been seen
    int64_t launched_us = 0;
};

struct Broker {
    mpb_ctx *ctx;
    Mapping map;
    int n_slots;
    Lane lane[BRK_LANES];
    size_t off_q, off_ee, off_ns, off_pass, off_done, off_cls, off_ident, off_nsdev;   // offsets inside a lane's blocks (n_slots reads)
    bool zero_copy = true;                                             // MPB_BROKER_COPIES=1: stage through HBM with two async copies
    // What a slot asked for, read ONCE when it goes SUBMITTED -> RUNNING and checked there (the segment is writable by
    // every client; a worker killed in mid-write, or a buggy one, must not be able to send the one process that serves everybody
    // into a wild memcpy): everything after that uses these copies, never the slot's own fields.
    int32_t s_len[BRK_MAX_SLOTS];
    int32_t s_priv[BRK_MAX_SLOTS];
    double s_alpha[BRK_MAX_SLOTS];
    // reads a finished micro-batch hands back (row budget missed): queued by whoever retires, run alone by the launch thread
    // (calls on one context must not overlap)
    std::atomic<int> solo_q[BRK_MAX_SLOTS];
    std::atomic<uint32_t> solo_head{0}, solo_tail{0};

    // ---- the resident server (k_serve, mpb_kernels.hip): calls without a launch each -------------------------------------
    // One mailbox entry per slot in pinned host memory; the broker thread copies a request in (row, parameters, then the
    // door word) and polls done[e]; the kernel's waves poll the door words over the link.  The kernel leaves by itself
    // when its lifetime is over (the grid always drains) and says so in *exited: while calls keep coming it is launched
    // again at once, otherwise by the next call.  MPB_BROKER_SERVER=0 keeps the launch-per-micro-batch lanes.
    struct Server {
        bool enabled = false, running = false;
        hipStream_t stream = nullptr;
        char *pin = nullptr, *dev = nullptr;
        MpbServeBox box{};
        uint8_t *h_q = nullptr;
        MpbServePrm *h_prm = nullptr;
        volatile unsigned long long *h_door = nullptr;
        volatile uint32_t *h_done = nullptr, *h_stop = nullptr, *h_exited = nullptr;
        double *h_ee = nullptr;
        int32_t *h_ns = nullptr;
        uint8_t *h_pass = nullptr;
        uint32_t generation = 0;
        uint32_t tok[BRK_MAX_SLOTS];
        bool posted[BRK_MAX_SLOTS];
        int64_t posted_us[BRK_MAX_SLOTS];
        int n_posted = 0;
        int64_t launched_us = 0, last_post_us = 0;
        double cached_alpha = -1.0;
        MpbDevParams cached_prm;
        uint32_t lifetime_ms = 100;
        bool direct = false;                                // the kernel serves the slots themselves (the segment is registered)
        uint32_t seen_direct_seq = 0;
        int64_t turn_us = 0, turn_n = 0;                    // MPB_BROKER_TRACE: door word -> results seen, summed
    } srv;

    int init_server()
    {
        const size_t ns = (size_t)n_slots;
        size_t o = 0;
        auto take_bytes = [&](size_t bytes) { const size_t at = o; o = (o + bytes + 255) & ~(size_t)255; return at; };
        const size_t o_q = take_bytes(ns * MPB_SERVE_STRIDE), o_prm = take_bytes(ns * sizeof(MpbServePrm)), o_door = take_bytes(ns * 8),
                     o_done = take_bytes(ns * 4), o_ee = take_bytes(ns * 8), o_ns = take_bytes(ns * 4), o_pass = take_bytes(ns),
                     o_stop = take_bytes(64), o_exited = take_bytes(64);
        const size_t pin_bytes = o;
        o = 0;
        const size_t d_stage = take_bytes(ns * MPB_SERVE_STRIDE), d_ns = take_bytes(ns * 4), d_cls = take_bytes(ns), d_ident = take_bytes(ns * 4),
                     d_gone = take_bytes(64);
        const size_t dev_bytes = o;
        BHIP(hipStreamCreateWithFlags(&srv.stream, hipStreamNonBlocking));
        BHIP(hipHostMalloc((void **)&srv.pin, pin_bytes, hipHostMallocMapped));
        BHIP(hipMalloc((void **)&srv.dev, dev_bytes));
        BHIP(hipMemset(srv.dev, 0, dev_bytes));
        memset(srv.pin, 0, pin_bytes);
        srv.h_q = (uint8_t *)(srv.pin + o_q);
        srv.h_prm = (MpbServePrm *)(srv.pin + o_prm);
        srv.h_door = (volatile unsigned long long *)(srv.pin + o_door);
        srv.h_done = (volatile uint32_t *)(srv.pin + o_done);
        srv.h_ee = (double *)(srv.pin + o_ee);
        srv.h_ns = (int32_t *)(srv.pin + o_ns);
        srv.h_pass = (uint8_t *)(srv.pin + o_pass);
        srv.h_stop = (volatile uint32_t *)(srv.pin + o_stop);
        srv.h_exited = (volatile uint32_t *)(srv.pin + o_exited);
        MpbServeBox &x = srv.box;
        x.q = srv.h_q; x.stride = MPB_SERVE_STRIDE; x.prm = srv.h_prm; x.door = (const unsigned long long *)srv.h_door;
        x.done = (uint32_t *)srv.h_done; x.ee = srv.h_ee; x.ns = srv.h_ns; x.pass = srv.h_pass;
        x.stop = (const uint32_t *)srv.h_stop; x.exited = (uint32_t *)srv.h_exited;
        x.stage = (uint8_t *)(srv.dev + d_stage); x.ns_dev = (int32_t *)(srv.dev + d_ns); x.cls = (uint8_t *)(srv.dev + d_cls);
        x.ident = (int32_t *)(srv.dev + d_ident); x.gone = (uint32_t *)(srv.dev + d_gone);
        x.q_step = MPB_SERVE_STRIDE; x.prm_step = sizeof(MpbServePrm); x.door_step = 8; x.done_step = 4; x.ee_step = 8; x.ns_step = 4; x.pass_step = 1;
        x.n_ent = n_slots;
        for (int i = 0; i < BRK_MAX_SLOTS; i++) { srv.tok[i] = 0; srv.posted[i] = false; srv.posted_us[i] = 0; }
        // Direct serving: the shared-memory segment itself registered with the runtime, the kernel's entry e = slot e.  Then a
        // call is: the worker writes row + parameters + door word into its slot, the wave answers there -- this thread only
        // keeps the kernel resident.  (Registration refused, or MPB_BROKER_DIRECT=0: the copies above stay.)
        const bool want_direct = !(getenv("MPB_BROKER_DIRECT") && atoi(getenv("MPB_BROKER_DIRECT")) == 0);
        if (want_direct && hipHostRegister(map.base, map.bytes, hipHostRegisterMapped) == hipSuccess) {
            void *dbase = nullptr;
            if (hipHostGetDevicePointer(&dbase, map.base, 0) == hipSuccess && dbase) {
                BrkSlot *s0 = (BrkSlot *)((char *)dbase + sizeof(BrkHeader));
                x.q = (const uint8_t *)s0 + SLOT_ROW_OFF; x.prm = &s0->d_prm; x.door = &s0->door; x.done = &s0->d_done;
                x.ee = &s0->d_ee; x.ns = &s0->d_ns; x.pass = &s0->d_pass;
                x.q_step = x.prm_step = x.door_step = x.done_step = x.ee_step = x.ns_step = x.pass_step = (int64_t)SLOT_BYTES;
                srv.direct = true;
            } else {
                (void)hipHostUnregister(map.base);
            }
        }
        (void)hipGetLastError();
        srv.enabled = true;
        return MPB_OK;
    }

    uint32_t exited_generation() const { return __atomic_load_n((const uint32_t *)srv.h_exited, __ATOMIC_ACQUIRE); }

    int server_launch()
    {
        if (++srv.generation == 0) srv.generation = 1;
        const int rc = mpbi_serve_launch(ctx, &srv.box, srv.generation, srv.lifetime_ms, srv.stream);
        if (rc) return rc;
        srv.running = true;
        srv.launched_us = now_us();
        map.hdr()->batches.fetch_add(1, std::memory_order_relaxed);
        map.hdr()->server_up.store(1, std::memory_order_release);
        return MPB_OK;
    }

    // the request of slot si (taken: s_len / s_alpha are the checked copies) -> its mailbox entry
    int server_post(int si)
    {
        BrkSlot *s = map.slot(si);
        const int32_t len = s_len[si];
        memcpy(srv.h_q + (size_t)si * MPB_SERVE_STRIDE, Mapping::row(s), (size_t)((len + 15) & ~15));
        if (s_alpha[si] != srv.cached_alpha) { mpbi_small_params(s_alpha[si], &srv.cached_prm); srv.cached_alpha = s_alpha[si]; }
        srv.h_prm[si].p = srv.cached_prm;
        uint32_t t = srv.tok[si] + 1;
        if (t == 0) t = 1;
        srv.tok[si] = t;
        __atomic_store_n((unsigned long long *)&srv.h_door[si], ((unsigned long long)(uint32_t)len << 32) | t, __ATOMIC_RELEASE);
        srv.posted[si] = true;
        srv.posted_us[si] = srv.last_post_us = now_us();
        srv.n_posted++;
        if (!srv.running) return server_launch();
        return MPB_OK;
    }

    // results of the entries that have been served; true if any
    bool server_collect()
    {
        bool any = false;
        for (int i = 0; i < n_slots && srv.n_posted > 0; i++) {
            if (!srv.posted[i] || __atomic_load_n((const uint32_t *)&srv.h_done[i], __ATOMIC_ACQUIRE) != srv.tok[i]) continue;
            srv.posted[i] = false;
            srv.n_posted--;
            any = true;
            srv.turn_us += now_us() - srv.posted_us[i];
            srv.turn_n++;
            if (srv.h_pass[i] == 2) run_solo(i);           // row budget missed (or a wide read): the ordinary per-read path
            else finish(i, MPB_OK, srv.h_ee[i], srv.h_ns[i], nullptr);
        }
        return any;
    }

    // the launch has drained (lifetime over): again at once while calls are pending or were a moment ago.
    // A request that has been out for 50 ms: ask the runtime (a fault shows there).
    int server_watch()
    {
        if (srv.running && exited_generation() == srv.generation) { srv.running = false; map.hdr()->server_up.store(0, std::memory_order_release); }
        const int64_t t = now_us();
        if (srv.direct) {                                   // calls that do not pass through this thread: their counter says they keep coming
            const uint32_t ds = map.hdr()->direct_seq.load(std::memory_order_acquire);
            if (ds != srv.seen_direct_seq) { srv.seen_direct_seq = ds; srv.last_post_us = t; }
        }
        if (!srv.running && (srv.n_posted > 0 || t - srv.last_post_us < 20000)) return server_launch();
        if (srv.running && srv.n_posted > 0) {
            int64_t oldest = t;
            for (int i = 0; i < n_slots; i++) if (srv.posted[i] && srv.posted_us[i] < oldest) oldest = srv.posted_us[i];
            if (t - oldest > 50000 && t - srv.launched_us > 50000) {
                const hipError_t q = hipStreamQuery(srv.stream);
                if (q != hipSuccess && q != hipErrorNotReady) return mpbi_fail(MPB_E_HIP, hipGetErrorString(q));
                if (q == hipSuccess && exited_generation() != srv.generation) srv.running = false;   // gone without a word: launch again
            }
        }
        return MPB_OK;
    }

    void free_server()
    {
        if (!srv.stream) return;
        if (srv.h_stop) __atomic_store_n((uint32_t *)srv.h_stop, 1u, __ATOMIC_SEQ_CST);
        if (getenv("MPB_BROKER_TRACE") && srv.turn_n) {
            fprintf(stderr, "[broker] resident server: %lld requests, %.1f us from the door word to the results on average, %u launches\n",
                    (long long)srv.turn_n, (double)srv.turn_us / (double)srv.turn_n, srv.generation);
        }
        (void)hipStreamSynchronize(srv.stream);
        if (srv.running && srv.h_exited) {                  // (a runtime that completes at once -- the test stub -- still has its server out)
            const int64_t until = now_ms() + 2000;
            while (exited_generation() != srv.generation && now_ms() < until) usleep(100);
        }
        map.hdr()->direct.store(0);
        map.hdr()->server_up.store(0);
        if (srv.direct) (void)hipHostUnregister(map.base);
        (void)hipStreamDestroy(srv.stream);
        if (srv.pin) (void)hipHostFree(srv.pin);
        if (srv.dev) (void)hipFree(srv.dev);
        srv.stream = nullptr; srv.pin = srv.dev = nullptr; srv.enabled = srv.running = false;
    }

    bool take(int si)                                                  // SUBMITTED -> RUNNING with a checked copy of the request
    {
        BrkSlot *s = map.slot(si);
        const int32_t len = s->len, priv = s->priv;
        const double alpha = s->alpha;
        s->state.store(ST_RUNNING, std::memory_order_relaxed);
        if (len < 0 || len > MPB_MAX_LEN || !(alpha > 0 && alpha < 1) || (priv != 0 && priv != 1)) {
            mpbi_fail(MPB_E_INVALID, "malformed request in the broker slot (length, alpha or table flag out of range)");
            finish(si, MPB_E_INVALID, 0, 0, mpb_last_error());
            return false;
        }
        s_len[si] = len; s_priv[si] = priv; s_alpha[si] = alpha;
        return true;
    }

    int init_lanes()
    {
        const size_t ns = (size_t)n_slots;
        off_q = (ns * 4 + 255) & ~(size_t)255;                        // input block: len[n_slots] | q[m x stride]
        const size_t in_cap = off_q + ns * (size_t)MPB_SMALL_MAX_STRIDE;   // (a longer read is run alone, through the pipeline)
        off_ee = 0;                                                   // output block: ee | ns | pass
        off_ns = ns * 8;
        off_pass = off_ns + ns * 4;
        off_done = (off_pass + ns + 63) & ~(size_t)63;                // ... | completion word per read (zero-copy lanes)
        const size_t out_cap = (off_done + ns * 4 + 255) & ~(size_t)255;
        off_cls = in_cap + out_cap;                                   // device block: input | output | cls | ident | ns
        off_ident = (off_cls + ns + 255) & ~(size_t)255;
        off_nsdev = off_ident + ns * 4;
        const size_t dev_cap = off_nsdev + ns * 4;
        for (Lane &l : lane) {
            BHIP(hipStreamCreateWithFlags(&l.stream, hipStreamNonBlocking));
            BHIP(hipHostMalloc((void **)&l.pin_in, in_cap, hipHostMallocMapped));
            BHIP(hipHostMalloc((void **)&l.pin_out, out_cap, hipHostMallocMapped));
            BHIP(hipMalloc((void **)&l.dev, dev_cap));
            l.in_cap = in_cap; l.out_cap = out_cap; l.dev_cap = dev_cap;
        }
        return MPB_OK;
    }

    void free_lanes()
    {
        for (Lane &l : lane) {
            if (l.stream) { (void)hipStreamSynchronize(l.stream); (void)hipStreamDestroy(l.stream); }
            if (l.pin_in) (void)hipHostFree(l.pin_in);
            if (l.pin_out) (void)hipHostFree(l.pin_out);
            if (l.dev) (void)hipFree(l.dev);
            l.stream = nullptr; l.pin_in = l.pin_out = l.dev = nullptr;
        }
    }

    void finish(int si, int rc, double ee, int32_t ns, const char *err)
    {
        BrkSlot *s = map.slot(si);
        s->rc = rc; s->ee = ee; s->ns = ns;
        if (err) { strncpy(s->err, err, sizeof(s->err) - 1); s->err[sizeof(s->err) - 1] = 0; } else s->err[0] = 0;
        s->state.store(ST_DONE, std::memory_order_seq_cst);
        if (s->waiting.load(std::memory_order_seq_cst)) futex_wake(&s->state, 1);
        map.hdr()->served.fetch_add(1, std::memory_order_relaxed);
    }

    // a read the micro-batch cannot take (private table), or that came back with pass == 2: alone, synchronously
    void run_solo(int si)
    {
        BrkSlot *s = map.slot(si);
        double ee = 0;
        int32_t ns = 0;
        const int32_t len = s_len[si];
        const int32_t stride = (int32_t)(((len > 0 ? len : 1) + 15) & ~15);
        const int rc = mpbi_run_packed_read(ctx, Mapping::row(s), len, stride, s_priv[si] ? Mapping::lut(s) : nullptr, s_alpha[si], &ee, &ns);
        map.hdr()->solo.fetch_add(1, std::memory_order_relaxed);
        finish(si, rc, ee, ns, rc ? mpb_last_error() : nullptr);
    }

    int launch(Lane &l, const int *cand, int m, double alpha)
    {
        int32_t maxlen = 1;
        for (int k = 0; k < m; k++) maxlen = std::max(maxlen, s_len[cand[k]]);
        const int64_t stride = (maxlen + 15) & ~15;
        int32_t *h_len = (int32_t *)l.pin_in;
        uint8_t *h_q = (uint8_t *)l.pin_in + off_q;
        for (int k = 0; k < m; k++) {
            BrkSlot *s = map.slot(cand[k]);
            const int32_t len = s_len[cand[k]];                       // checked by take(): 0 .. MPB_SMALL_MAX_STRIDE - 1 here
            h_len[k] = len;
            memcpy(h_q + (size_t)k * stride, Mapping::row(s), (size_t)((len + 15) & ~15));     // bytes past len are never looked at
            l.slots[k] = cand[k];
        }
        l.m = m; l.stride = stride;
        char *d = l.dev;
        if (zero_copy) {
            // ONE runtime call per micro-batch: the kernel reads the rows from, and writes the results to, the lane's pinned
            // host blocks (mapped into the device's address space); only its scratch (class bytes, identity list) is in HBM.
            // A few hundred bytes per read over the link cost less than two more asynchronous copies cost the broker thread,
            // which is what bounds the call rate at P = 16.
            if (++l.token == 0) l.token = 1;
            volatile uint32_t *done = (volatile uint32_t *)(l.pin_out + off_done);
            for (int k = 0; k < m; k++) done[k] = 0;
            l.reported = 0;
            const MpbSmallHost hostside{(uint8_t *)d + off_q, (int32_t *)(d + off_nsdev), (uint32_t *)(l.pin_out + off_done), l.token};
            int rc = mpbi_small_async(ctx, (const uint8_t *)l.pin_in + off_q, m, stride, (const int32_t *)l.pin_in, alpha,
                                      (double *)(l.pin_out + off_ee), (int32_t *)(l.pin_out + off_ns), (uint8_t *)(l.pin_out + off_pass),
                                      (uint8_t *)d + off_cls, (int32_t *)(d + off_ident), l.stream, &hostside);
            if (rc) return rc;
            l.launched_us = now_us();
        } else {
            const size_t in_bytes = off_q + (size_t)m * stride;
            BHIP(hipMemcpyAsync(d, l.pin_in, in_bytes, hipMemcpyHostToDevice, l.stream));
            char *d_out = d + l.in_cap;
            int rc = mpbi_small_async(ctx, (const uint8_t *)d + off_q, m, stride, (const int32_t *)d, alpha, (double *)(d_out + off_ee),
                                      (int32_t *)(d_out + off_ns), (uint8_t *)(d_out + off_pass), (uint8_t *)d + off_cls,
                                      (int32_t *)(d + off_ident), l.stream, nullptr);
            if (rc) return rc;
            BHIP(hipMemcpyAsync(l.pin_out, d_out, off_pass + (size_t)n_slots, hipMemcpyDeviceToHost, l.stream));
        }
        map.hdr()->batches.fetch_add(1, std::memory_order_relaxed);
        l.busy.store(1, std::memory_order_release);                   // everything above is the retiring thread's from here on
        return MPB_OK;
    }

    // hipSuccess: the lane's launch has delivered; hipErrorNotReady: not yet; anything else: the runtime's error.
    // Zero-copy lanes are told by the kernel itself (a word per read in pinned memory: no runtime call per iteration, and no
    // wait for the completion signal, which trails the last wave by microseconds); a launch that has been out for 20 ms is
    // asked about through the runtime, which is also where a fault would show.
    hipError_t finished(Lane &l)
    {
        if (!zero_copy) return hipStreamQuery(l.stream);
        const volatile uint32_t *done = (const volatile uint32_t *)(l.pin_out + off_done);
        while (l.reported < l.m && done[l.reported] == l.token) l.reported++;
        if (l.reported == l.m) { std::atomic_thread_fence(std::memory_order_acquire); return hipSuccess; }
        if (now_us() - l.launched_us > 20000) {
            const hipError_t q = hipStreamQuery(l.stream);
            if (q != hipSuccess) return q;
            while (l.reported < l.m && done[l.reported] == l.token) l.reported++;
            return l.reported == l.m ? hipSuccess : hipErrorUnknown;
        }
        return hipErrorNotReady;
    }

    void retire(Lane &l)
    {
        const double *ee = (const double *)(l.pin_out + off_ee);
        const int32_t *ns = (const int32_t *)(l.pin_out + off_ns);
        const uint8_t *pass = (const uint8_t *)(l.pin_out + off_pass);
        for (int k = 0; k < l.m; k++) {
            if (pass[k] == 2) {                                // row budget missed (or more than 1024 rows): the ordinary per-read path,
                const uint32_t t = solo_tail.load(std::memory_order_relaxed);      // on the launch thread
                solo_q[t % BRK_MAX_SLOTS].store(l.slots[k], std::memory_order_relaxed);
                solo_tail.store(t + 1, std::memory_order_release);
            } else finish(l.slots[k], MPB_OK, ee[k], ns[k], nullptr);
        }
        l.busy.store(0, std::memory_order_release);
    }

    bool run_queued_solos()
    {
        bool any = false;
        for (;;) {
            const uint32_t h = solo_head.load(std::memory_order_relaxed);
            if (h == solo_tail.load(std::memory_order_acquire)) return any;
            run_solo(solo_q[h % BRK_MAX_SLOTS].load(std::memory_order_relaxed));
            solo_head.store(h + 1, std::memory_order_release);
            any = true;
        }
    }

    void fail_lane(Lane &l, int rc)
    {
        for (int k = 0; k < l.m; k++) finish(l.slots[k], rc, 0, 0, mpb_last_error());
        l.busy.store(0, std::memory_order_release);
    }
};

}  // namespace

extern "C" {

// Same object?  (the name may have been unlinked and created again while somebody waited for a lock on the old one)
static bool same_object(int fd, const char *path)
{
    struct stat a, b;
    const int fd2 = shm_open(path, O_RDWR, 0600);
    if (fd2 < 0) return false;
    const bool same = fstat(fd, &a) == 0 && fstat(fd2, &b) == 0 && a.st_ino == b.st_ino && a.st_dev == b.st_dev;
    close(fd2);
    return same;
}

// The segment of `path`, ours: created, sized, header written (pid, BS_STARTING; magic still 0) -- all of it under an exclusive
// flock on the object itself, so that of any number of brokers started at the same moment exactly one gets through (ADVICE r4,
// the C entry no longer relies on the Python starter's lock file).  Whoever holds the lock and finds
//   - a header whose pid is alive and whose state is not BS_EXITING (starting or serving): leaves it alone -> MPB_E_INVALID;
//   - anything else of non-zero size (a dead broker's segment, one that is on its way out): unlinks the NAME and starts over
//     with a fresh object -- processes still attached to the old one keep their mapping and find its pid dead;
//   - an empty object: initialises it.
// The descriptor stays open (and unlocked) for the broker's lifetime: leaving, it unlinks the name only if it still names this
// object (a successor may have replaced it while this broker was on its way out).
static int claim_segment(const char *path, size_t bytes, int32_t n_slots, int device, int *fd_out, void **base_out)
{
    for (int attempt = 0; attempt < 100; attempt++) {
        const int fd = shm_open(path, O_RDWR | O_CREAT, 0600);
        if (fd < 0) return mpbi_fail(MPB_E_INVALID, "cannot open or create the broker segment");
        if (flock(fd, LOCK_EX) != 0) { close(fd); return mpbi_fail(MPB_E_INVALID, "cannot lock the broker segment"); }
        if (!same_object(fd, path)) { close(fd); continue; }            // replaced while we waited: look again
        struct stat st;
        if (fstat(fd, &st) != 0) { close(fd); return mpbi_fail(MPB_E_INVALID, "cannot stat the broker segment"); }
        if ((size_t)st.st_size >= sizeof(BrkHeader)) {
            void *p = mmap(nullptr, sizeof(BrkHeader), PROT_READ, MAP_SHARED, fd, 0);
            bool live = false;
            if (p != MAP_FAILED) {
                const BrkHeader *h = (const BrkHeader *)p;
                const int opid = h->pid.load();
                live = h->state.load() != BS_EXITING && pid_alive(opid) && opid != (int)getpid();
                munmap(p, sizeof(BrkHeader));
            }
            if (live) { close(fd); return mpbi_fail(MPB_E_INVALID, "a broker of this name is already serving (or starting)"); }
            shm_unlink(path);                                            // a corpse: the name goes, a fresh object comes
            close(fd);
            continue;
        }
        if (st.st_size != 0) { shm_unlink(path); close(fd); continue; } // a half-sized leftover
        if (ftruncate(fd, (off_t)bytes) != 0) { shm_unlink(path); close(fd); return mpbi_fail(MPB_E_NOMEM, "cannot size the broker segment"); }
        void *p = mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
        if (p == MAP_FAILED) { shm_unlink(path); close(fd); return mpbi_fail(MPB_E_NOMEM, "mmap of the broker segment failed"); }
        BrkHeader *h = new (p) BrkHeader();         // ftruncate zero-filled the pages: every slot is free and idle
        h->n_slots = n_slots;
        h->slot_bytes = (int32_t)SLOT_BYTES;
        h->version = BRK_VERSION;
        h->device.store(device);
        h->heartbeat_ms.store(now_ms());
        h->state.store(BS_STARTING);
        h->pid.store((int32_t)getpid());            // from here on a second starter finds a live owner
        flock(fd, LOCK_UN);
        *fd_out = fd;
        *base_out = p;
        return MPB_OK;
    }
    return mpbi_fail(MPB_E_INVALID, "the broker segment kept changing hands; giving up");
}

int mpb_broker_serve(mpb_ctx *ctx, const char *name, int32_t n_slots, int32_t idle_exit_ms)
{
    if (!ctx) return mpbi_fail(MPB_E_INVALID, "ctx is NULL");
    if (n_slots < 1 || n_slots > BRK_MAX_SLOTS) return mpbi_fail(MPB_E_INVALID, "n_slots must be 1..256");
    char path[128];
    int rc = shm_path(name, path, sizeof(path));
    if (rc) return rc;
    BHIP(hipSetDevice(mpbi_ctx_device(ctx)));
    const size_t bytes = sizeof(BrkHeader) + (size_t)n_slots * SLOT_BYTES;
    int seg_fd = -1;
    void *p = nullptr;
    if ((rc = claim_segment(path, bytes, n_slots, mpbi_ctx_device(ctx), &seg_fd, &p))) return rc;
    Broker b;
    b.ctx = ctx;
    b.map.base = p;
    b.map.bytes = bytes;
    b.n_slots = n_slots;
    BrkHeader *h = b.map.hdr();
    b.zero_copy = !(getenv("MPB_BROKER_COPIES") && atoi(getenv("MPB_BROKER_COPIES")) != 0);
    // Two knobs of the serving loop, both measured (profiles/r05_per_read_concurrency.txt).  Gather window: with more workers
    // attached than have a read in, wait this many microseconds for the stragglers before launching -- it halves the launches
    // (4.2 instead of 2.1 reads per launch at 16 workers).  Retire thread (MPB_BROKER_RETIRE_THREAD=1): a second thread that only
    // watches the lanes' completion words and hands the results back.  Neither moves the call rate of 16 Python workers
    // (3.3-3.5 x 10^5 calls/s either way): that rate is bound by the workers' own cycle -- 28 us per call alone, 45 us with 16 of
    // them spinning on the 16 CPUs the box grants, which the kernel's CPU-quota throttling counters show -- not by this thread.
    // So the window stays (fewer launches) and the second thread is off by default (it is one more spinning thread on the quota).
    const int gather_us = getenv("MPB_BROKER_GATHER_US") ? atoi(getenv("MPB_BROKER_GATHER_US")) : 5;
    const bool retire_thread = getenv("MPB_BROKER_RETIRE_THREAD") && atoi(getenv("MPB_BROKER_RETIRE_THREAD")) != 0;
    rc = b.init_lanes();
    const bool want_server = !(getenv("MPB_BROKER_SERVER") && atoi(getenv("MPB_BROKER_SERVER")) == 0) && b.zero_copy;
    if (rc == MPB_OK && want_server) {
        // The context's one-read workspaces are sized BEFORE the resident server is first launched: a workspace that
        // grows later frees memory, the runtime then waits for the whole device, and the broker's own server is a kernel the
        // context's quiesce cannot see -- the broker thread would stall for the rest of that kernel's lifetime (<= 100 ms).
        // One read of the longest row the server takes through the path run_solo uses covers the reads it hands back; only a read
        // of more than 16384 bases (the chunked host pipeline) can still grow something, once.
        std::vector<uint8_t> row(MPB_SERVE_STRIDE, 30);
        double ee = 0;
        int32_t ns = 0;
        (void)mpbi_run_packed_read(ctx, row.data(), MPB_SERVE_STRIDE - 1, MPB_SERVE_STRIDE, nullptr, 0.005, &ee, &ns);
        rc = b.init_server();
    }
    if (rc == MPB_OK) {
        std::atomic_thread_fence(std::memory_order_seq_cst);
        h->direct.store(b.srv.enabled && b.srv.direct ? 1 : 0);
        h->magic = BRK_MAGIC;                        // clients accept the segment from here on
        h->state.store(BS_SERVING);
    }
    std::atomic<int> fatal{0}, quit{0};
    std::thread retirer;
    if (rc == MPB_OK && retire_thread) {
        const int device = mpbi_ctx_device(ctx);
        retirer = std::thread([&b, &fatal, &quit, device] {
            (void)hipSetDevice(device);
            int idle = 0;
            while (!quit.load(std::memory_order_acquire)) {
                bool any = false, did = false;
                for (Lane &l : b.lane) {
                    if (!l.busy.load(std::memory_order_acquire)) continue;
                    any = true;
                    const hipError_t q = b.finished(l);
                    if (q == hipSuccess) { b.retire(l); did = true; }
                    else if (q != hipErrorNotReady) {
                        mpbi_fail(MPB_E_HIP, hipGetErrorString(q));
                        b.fail_lane(l, MPB_E_HIP);
                        fatal.store(1, std::memory_order_release);
                    }
                }
                if (did) { idle = 0; continue; }
                if (any) { cpu_relax(); idle = 0; continue; }
                if (++idle < 2000) cpu_relax();       // a launch is usually microseconds away; after that, off the CPU
                else usleep(50);
            }
        });
    }
    int64_t last_work = now_ms(), last_house = last_work;
    int cand[BRK_MAX_SLOTS];
    while (rc == MPB_OK && !h->stop.load(std::memory_order_relaxed)) {
        bool progress = false;
        bool any_busy = false;
        if (fatal.load(std::memory_order_acquire)) { rc = MPB_E_HIP; break; }   // the context is gone: stop serving (clients see BS_EXITING)
        // 1. retire the micro-batches that have finished (every busy lane is asked: launches on streams of their own do not
        //    finish in launch order, and asking only the oldest one -- tried -- costs a fifth of the call rate) -- unless the
        //    retire thread does that
        for (Lane &l : b.lane) {
            if (!l.busy.load(std::memory_order_acquire)) continue;
            any_busy = true;
            if (retire_thread) continue;
            const hipError_t q = b.finished(l);
            if (q == hipSuccess) { b.retire(l); progress = true; }
            else if (q != hipErrorNotReady) {
                mpbi_fail(MPB_E_HIP, hipGetErrorString(q));
                b.fail_lane(l, MPB_E_HIP);
                rc = MPB_E_HIP;
                break;
            }
        }
        if (rc) break;
        if (b.run_queued_solos()) progress = true;
        if (b.srv.enabled) {
            // 1s. results of the entries the resident server has finished; 2s. whatever has been submitted goes into its entry
            if (b.server_collect()) progress = true;
            for (int i = 0; i < n_slots && rc == MPB_OK; i++) {
                BrkSlot *s = b.map.slot(i);
                if (s->state.load(std::memory_order_acquire) != ST_SUBMITTED) continue;
                progress = true;
                if (!b.take(i)) continue;                      // malformed: answered with MPB_E_INVALID
                // its own table / a long row -- and, when the kernel serves the slots themselves, whatever a client still submits
                // here (what the kernel handed back: a missed row budget)
                if (b.srv.direct || b.s_priv[i] || b.s_len[i] > MPB_SERVE_STRIDE - 1) { b.run_solo(i); continue; }
                const int prc = b.server_post(i);
                if (prc) { b.finish(i, prc, 0, 0, mpb_last_error()); b.srv.posted[i] = false; b.srv.n_posted--; if (prc == MPB_E_HIP) rc = prc; }
            }
            if (rc == MPB_OK) { const int wrc = b.server_watch(); if (wrc == MPB_E_HIP) rc = wrc; }
            if (rc) break;
            if (b.srv.n_posted > 0) any_busy = true;
        }
        // 2. whatever has been submitted goes into the next free lane
        Lane *free_lane = nullptr;
        for (Lane &l : b.lane) if (!l.busy.load(std::memory_order_acquire)) { free_lane = &l; break; }
        if (free_lane && !b.srv.enabled) {
            int m = 0, attached = 0, running = 0;
            double alpha = 0;
            auto collect = [&](bool first) {
                for (int i = 0; i < n_slots; i++) {
                    BrkSlot *s = b.map.slot(i);
                    const uint32_t st = s->state.load(std::memory_order_acquire);
                    if (first) {
                        if (s->owner.load(std::memory_order_relaxed)) attached++;
                        if (st == ST_RUNNING) running++;
                    }
                    if (st != ST_SUBMITTED) continue;
                    if (m && s->alpha != alpha) continue;          // another alpha: the next micro-batch
                    if (!b.take(i)) { progress = true; continue; } // malformed: answered with MPB_E_INVALID
                    if (b.s_priv[i] || b.s_len[i] > MPB_SMALL_MAX_STRIDE - 1) {   // its own code table, or a row too long for k_small
                        b.run_solo(i);
                        progress = true;
                        continue;
                    }
                    if (m == 0) alpha = b.s_alpha[i];
                    else if (b.s_alpha[i] != alpha) {              // (the slot's alpha changed under us: its own launch)
                        b.run_solo(i);
                        progress = true;
                        continue;
                    }
                    cand[m++] = i;
                }
            };
            collect(true);
            if (m && gather_us > 0 && m < attached - running) {
                const int64_t until = now_us() + gather_us;
                while (m < attached - running && now_us() < until) { cpu_relax(); collect(false); }
            }
            if (m) {
                const int lrc = b.launch(*free_lane, cand, m, alpha);
                if (lrc) {
                    for (int k = 0; k < m; k++) b.finish(cand[k], lrc, 0, 0, mpb_last_error());
                    if (lrc == MPB_E_HIP) { rc = lrc; break; }
                }
                progress = true;
                any_busy = true;
            }
        }
        const int64_t t = now_ms();
        if (progress) { last_work = t; continue; }
        // 3. nothing to do right now
        if (any_busy) { cpu_relax(); continue; }
        if (t - last_house >= 20) {
            last_house = t;
            h->heartbeat_ms.store(t, std::memory_order_relaxed);
            bool attached = false;
            for (int i = 0; i < n_slots; i++) {     // a slot whose owner has died goes back to the pool
                BrkSlot *s = b.map.slot(i);
                const int o = s->owner.load(std::memory_order_relaxed);
                if (!o) continue;
                if (pid_alive(o)) { attached = true; continue; }
                s->state.store(ST_IDLE);
                s->waiting.store(0);
                s->owner.store(0);
            }
            if (attached) last_work = t;             // attached clients keep the broker alive even when they are quiet
            if (idle_exit_ms > 0 && t - last_work > idle_exit_ms) break;
        }
        // spin for a little while (a worker is usually back within tens of microseconds), then sleep on the submit word.
        // Direct serving: the calls do not come through here, this thread only has to notice that the kernel has left while
        // they keep coming -- a look every 200 us, hardly any spinning (a spinning thread is one CPU less for the workers).
        const bool watching = b.srv.enabled && b.srv.direct && (b.srv.running || now_us() - b.srv.last_post_us < 20000);
        const uint32_t seq = h->submit_seq.load(std::memory_order_acquire);
        const int64_t spin_until = now_us() + (b.srv.enabled && b.srv.direct ? 2 : 100);
        bool woke = false;
        while (now_us() < spin_until) {
            if (h->submit_seq.load(std::memory_order_acquire) != seq) { woke = true; break; }
            cpu_relax();
        }
        if (woke) continue;
        h->sleeping.store(1, std::memory_order_seq_cst);
        bool pending = false;
        for (int i = 0; i < n_slots && !pending; i++) pending = b.map.slot(i)->state.load(std::memory_order_seq_cst) == ST_SUBMITTED;
        if (!pending && !h->stop.load()) { if (watching) futex_wait_us(&h->submit_seq, seq, 200); else futex_wait(&h->submit_seq, seq, 20); }
        h->sleeping.store(0, std::memory_order_seq_cst);
    }
    // leave: nobody may wait for an answer that will not come
    h->state.store(BS_EXITING);
    quit.store(1, std::memory_order_release);
    if (retirer.joinable()) retirer.join();
    for (Lane &l : b.lane) if (l.busy.load()) { if (hipStreamSynchronize(l.stream) == hipSuccess) b.retire(l); else b.fail_lane(l, MPB_E_HIP); }
    if (b.srv.enabled && rc == MPB_OK) {                  // what the resident server still has: a moment to finish, then its results
        const int64_t until = now_ms() + 200;
        while (b.srv.n_posted > 0 && now_ms() < until) { if (!b.server_collect()) cpu_relax(); if (b.server_watch()) break; }
    }
    b.free_server();
    if (rc == MPB_OK) (void)b.run_queued_solos();
    for (int i = 0; i < n_slots; i++) {
        BrkSlot *s = b.map.slot(i);
        const uint32_t st = s->state.load();
        if (st == ST_SUBMITTED || st == ST_RUNNING) { mpbi_fail(MPB_E_HIP, "the broker is shutting down"); b.finish(i, MPB_E_HIP, 0, 0, mpb_last_error()); }
    }
    b.free_lanes();
    // the name goes only if it still names THIS object: a successor that found us exiting may have replaced it already
    if (flock(seg_fd, LOCK_EX) == 0) {
        if (same_object(seg_fd, path)) shm_unlink(path);
        flock(seg_fd, LOCK_UN);
    }
    close(seg_fd);
    b.map.unmap();
    return rc;
}

struct mpb_broker_client {
    Mapping map;
    int slot = -1;
    int32_t pid = 0;
    uint32_t tok = 0;                 // direct serving: the last door token of this attachment
    double prm_alpha = -1.0;          // ... and the parameters of the last alpha
    MpbDevParams prm;
};

int mpb_broker_attach(const char *name, int32_t wait_ms, mpb_broker_client **out)
{
    if (!out) return mpbi_fail(MPB_E_INVALID, "NULL output");
    *out = nullptr;
    const int64_t deadline = now_ms() + (wait_ms > 0 ? wait_ms : 0);
    Mapping m;
    for (;;) {
        int rc = map_existing(name, &m);
        if (rc == MPB_OK) {
            BrkHeader *h = m.hdr();
            const int st = h->state.load();
            if (st == BS_SERVING && pid_alive(h->pid.load())) break;
            m.unmap();
            if (st == BS_SERVING) { mpbi_fail(MPB_E_INVALID, "the broker process of this segment is gone"); }
            else mpbi_fail(MPB_E_INVALID, "the broker is not serving");
        }
        if (now_ms() >= deadline) return MPB_E_INVALID;
        usleep(2000);
    }
    const int32_t me = (int32_t)getpid();
    int got = -1;
    for (int i = 0; i < m.hdr()->n_slots && got < 0; i++) {
        int32_t expect = 0;
        if (m.slot(i)->owner.compare_exchange_strong(expect, me)) got = i;
    }
    if (got < 0) { m.unmap(); return mpbi_fail(MPB_E_NOMEM, "every slot of the broker is taken (more worker processes than slots)"); }
    BrkSlot *s = m.slot(got);
    s->waiting.store(0);
    s->state.store(ST_IDLE);
    mpb_broker_client *cl = new (std::nothrow) mpb_broker_client();
    if (!cl) { s->owner.store(0); m.unmap(); return mpbi_fail(MPB_E_NOMEM, "out of memory"); }
    cl->map = m;
    cl->slot = got;
    cl->pid = me;
    // direct serving: start behind whatever a former owner of the slot left (a door word still unserved, the token served last)
    const uint32_t t_door = (uint32_t)__atomic_load_n(&s->door, __ATOMIC_ACQUIRE), t_done = __atomic_load_n(&s->d_done, __ATOMIC_ACQUIRE);
    cl->tok = (int32_t)(t_door - t_done) > 0 ? t_door : t_done;
    *out = cl;
    return MPB_OK;
}

// Direct serving: the request goes to the resident kernel where it lies (the worker's slot); nothing of it passes through the
// broker thread.  *served = false: the kernel handed the read back (row budget missed / a wide read) -- the caller submits it.
static int direct_call(mpb_broker_client *cl, BrkHeader *h, BrkSlot *s, int32_t len, double alpha, double *ee, int32_t *ns, bool *served)
{
    *served = false;
    if (alpha != cl->prm_alpha) { mpbi_small_params(alpha, &cl->prm); cl->prm_alpha = alpha; }
    s->d_prm.p = cl->prm;
    uint32_t t = cl->tok + 1;
    if (t == 0) t = 1;
    cl->tok = t;
    __atomic_store_n(&s->door, ((unsigned long long)(uint32_t)len << 32) | t, __ATOMIC_RELEASE);
    h->direct_seq.fetch_add(1, std::memory_order_relaxed);
    auto kick = [&] {                                   // the kernel is not out: the broker launches it when it looks
        h->submit_seq.fetch_add(1, std::memory_order_seq_cst);
        if (h->sleeping.load(std::memory_order_seq_cst)) futex_wake(&h->submit_seq, 1);
    };
    if (!h->server_up.load(std::memory_order_acquire)) kick();
    const int64_t t0 = now_us();
    int64_t last_kick = t0, last_check = t0;
    for (unsigned spins = 1;; spins++) {
        if (__atomic_load_n(&s->d_done, __ATOMIC_ACQUIRE) == t) break;
        cpu_relax();
        if (spins & 255u) continue;
        const int64_t now = now_us();
        if (now - last_kick > 300 && !h->server_up.load(std::memory_order_acquire)) { kick(); last_kick = now; }
        if (now - last_check > 1000) {
            last_check = now;
            if (h->state.load(std::memory_order_acquire) != BS_SERVING || !h->direct.load(std::memory_order_acquire) ||
                (now - t0 > 1000000 && !pid_alive(h->pid.load()))) {
                if (__atomic_load_n(&s->d_done, __ATOMIC_ACQUIRE) == t) break;
                return mpbi_fail(MPB_E_HIP, "the broker went away while a read was pending");
            }
        }
        if (now - t0 > 2000) usleep(50);                // something is slow (a launch, a loaded box): off the CPU between looks
    }
    if (s->d_pass == 2) return MPB_OK;
    *ee = s->d_ee;
    *ns = s->d_ns;
    *served = true;
    __atomic_store_n(&s->n_direct, s->n_direct + 1, __ATOMIC_RELAXED);      // (a read handed back is counted by the broker, which serves it)
    return MPB_OK;
}

int mpb_broker_call(mpb_broker_client *cl, const char *contig, const int32_t *contig_quals, int32_t len, double alpha,
                    double *ee, int32_t *ns)
{
    if (!cl || !cl->map.base) return mpbi_fail(MPB_E_INVALID, "not attached to a broker");
    if (cl->pid != (int32_t)getpid()) return mpbi_fail(MPB_E_INVALID, "this attachment belongs to another process (forked after attaching): attach again");
    int rc = mpbi_check_one_read(contig, contig_quals, len, alpha, ee, ns);
    if (rc) return rc;
    BrkHeader *h = cl->map.hdr();
    BrkSlot *s = cl->map.slot(cl->slot);
    if (h->state.load(std::memory_order_acquire) != BS_SERVING) return mpbi_fail(MPB_E_HIP, "the broker has stopped serving");
    bool priv = false;
    const int32_t stride = (int32_t)(((len > 0 ? len : 1) + 15) & ~15);
    if ((rc = mpbi_pack_one_read(contig, contig_quals, len, false, Mapping::row(s), stride, Mapping::lut(s), &priv))) return rc;
    if (!priv && len <= MPB_SERVE_STRIDE - 1 && h->direct.load(std::memory_order_acquire)) {
        bool served = false;
        if ((rc = direct_call(cl, h, s, len, alpha, ee, ns, &served)) || served) return rc;
    }
    s->len = len;
    s->alpha = alpha;
    s->priv = priv ? 1 : 0;
    s->state.store(ST_SUBMITTED, std::memory_order_seq_cst);
    h->submit_seq.fetch_add(1, std::memory_order_seq_cst);
    if (h->sleeping.load(std::memory_order_seq_cst)) futex_wake(&h->submit_seq, 1);
    // wait: spin first (the answer is usually tens of microseconds away), then sleep on the slot's state word
    const int64_t spin_until = now_us() + 300;
    int spins = 0;
    for (;;) {
        if (s->state.load(std::memory_order_acquire) == ST_DONE) break;
        cpu_relax();
        if ((++spins & 63) == 0 && now_us() >= spin_until) {
            int64_t waited_ms = 0;
            for (;;) {
                s->waiting.store(1, std::memory_order_seq_cst);
                const uint32_t st = s->state.load(std::memory_order_seq_cst);
                if (st == ST_DONE) { s->waiting.store(0); break; }
                futex_wait(&s->state, st, 50);
                s->waiting.store(0, std::memory_order_seq_cst);
                if (s->state.load(std::memory_order_acquire) == ST_DONE) break;
                waited_ms += 50;
                if (h->state.load() != BS_SERVING || (waited_ms >= 1000 && !pid_alive(h->pid.load()))) {
                    if (s->state.load() == ST_DONE) break;
                    s->state.store(ST_IDLE);
                    return mpbi_fail(MPB_E_HIP, "the broker went away while a read was pending");
                }
            }
            break;
        }
    }
    rc = s->rc;
    if (rc == MPB_OK) { *ee = s->ee; *ns = s->ns; }
    else mpbi_fail(rc, s->err);
    s->state.store(ST_IDLE, std::memory_order_release);
    return rc;
}

int mpb_broker_detach(mpb_broker_client *cl)
{
    if (!cl) return MPB_OK;
    if (cl->map.base) {
        if (cl->pid == (int32_t)getpid()) {
            BrkSlot *s = cl->map.slot(cl->slot);
            s->state.store(ST_IDLE);
            s->owner.store(0);
        }
        cl->map.unmap();
    }
    delete cl;
    return MPB_OK;
}

int mpb_broker_shutdown(const char *name)
{
    Mapping m;
    int rc = map_existing(name, &m);
    if (rc) return rc;
    m.hdr()->stop.store(1);
    m.hdr()->submit_seq.fetch_add(1);
    futex_wake(&m.hdr()->submit_seq, 1);
    m.unmap();
    return MPB_OK;
}

int mpb_broker_stats(const char *name, int64_t *served, int64_t *batches, int64_t *solo, int32_t *pid, int32_t *attached)
{
    Mapping m;
    int rc = map_existing(name, &m);
    if (rc) return rc;
    const BrkHeader *h = m.hdr();
    if (served) {
        int64_t v = h->served.load();
        for (int i = 0; i < h->n_slots; i++) v += __atomic_load_n(&m.slot(i)->n_direct, __ATOMIC_RELAXED);     // direct calls: counted by the callers
        *served = v;
    }
    if (batches) *batches = h->batches.load();
    if (solo) *solo = h->solo.load();
    if (pid) *pid = (h->state.load() == BS_SERVING && pid_alive(h->pid.load())) ? h->pid.load() : 0;
    if (attached) {
        int a = 0;
        for (int i = 0; i < h->n_slots; i++) a += m.slot(i)->owner.load() != 0;
        *attached = a;
    }
    m.unmap();
    return MPB_OK;
}

}  // extern "C"
