// broker_stub.cpp -- TEST INFRASTRUCTURE, never shipped: lets the broker's inter-process protocol
// (moira_amd/csrc/mpb_broker.cpp: shared-memory slots, futex hand-over, micro-batch lanes, dead-owner reclaim,
// shutdown) run WITHOUT a GPU.  The product source file is compiled unchanged; this file supplies
//   * the dozen HIP runtime entry points it calls, as host-memory operations that complete at once, and
//   * the mpbi_* hooks of mpb_api.cpp, with the ORACLE (oracle/pb_oracle.c) doing the arithmetic,
// and the test library is linked from the two.  A read whose length is a multiple of 7 is reported "row budget missed"
// (pass == 2) by the stub's micro-batch, so that the broker's run-alone fallback is exercised as well.
// What this does NOT test: the kernels (tests/test_gpu_broker.py does, on the GPU).
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include "../../include/moira_pb.h"
#include "../../moira_amd/csrc/mpb_internal.h"

extern "C" {
typedef struct pbo_params { double alpha, uncert, maxerrors; int32_t ambig_mode; uint32_t flags; } pbo_params;
int pbo_filter_batch(const uint8_t *q, int64_t n, int64_t row_stride, const int32_t *len, int32_t fixed_len,
                     const pbo_params *prm, int shape, int threads, double *ee, int32_t *ns, uint8_t *pass, int32_t *rows);
int pbo_pack_read(const char *seq, const int32_t *quals, int32_t len, uint8_t *row_out, int32_t row_bytes);

static thread_local char g_err[512] = "";
const char *mpb_last_error(void) { return g_err; }
int mpbi_fail(int code, const char *msg) { snprintf(g_err, sizeof(g_err), "%s", msg); return code; }
int mpbi_ctx_device(const mpb_ctx *) { return 0; }

int mpbi_check_one_read(const char *contig, const int32_t *q, int32_t len, double alpha, const void *ee, const void *ns)
{
    if (!ee || !ns) return mpbi_fail(MPB_E_INVALID, "NULL output");
    if (!(alpha > 0 && alpha < 1)) return mpbi_fail(MPB_E_INVALID, "Alpha must be between 0 and 1");
    if (contig && (int32_t)strlen(contig) != len) return mpbi_fail(MPB_E_INVALID, "contig and contig_quals must have the same length");
    if (len > 65535) return mpbi_fail(MPB_E_INVALID, "reads longer than 65535 bases are not supported");
    return MPB_OK;
}

int mpbi_pack_one_read(const char *contig, const int32_t *quals, int32_t len, bool, uint8_t *row, int32_t row_bytes, double2 *, bool *priv)
{
    *priv = false;
    for (int32_t i = 0; i < len; i++) {
        if (quals[i] < 0) return mpbi_fail(MPB_E_RANGE, "Qualities must have positive values.");
        if (quals[i] > 254) *priv = true;            // the stub has no private tables: it only carries the flag to the broker
    }
    if (*priv) {                                     // clamp so that the oracle can score it (the flag routes it to run_solo)
        int32_t *tmp = (int32_t *)malloc(sizeof(int32_t) * (size_t)(len ? len : 1));
        for (int32_t i = 0; i < len; i++) tmp[i] = quals[i] > 254 ? 254 : quals[i];
        const int rc = pbo_pack_read(contig, tmp, len, row, row_bytes);
        free(tmp);
        return rc ? mpbi_fail(MPB_E_RANGE, "pack failed") : MPB_OK;
    }
    return pbo_pack_read(contig, quals, len, row, row_bytes) ? mpbi_fail(MPB_E_RANGE, "pack failed") : MPB_OK;
}

static int oracle_rows(const uint8_t *q, int64_t m, int64_t stride, const int32_t *len, double alpha, double *ee, int32_t *ns, uint8_t *pass)
{
    pbo_params p{alpha, 1.0, NAN, 1 /* ignore */, 0};
    return pbo_filter_batch(q, m, stride, len, 0, &p, 0, 1, ee, ns, pass, nullptr) ? mpbi_fail(MPB_E_INVALID, "oracle failed") : MPB_OK;
}

int mpbi_run_packed_read(mpb_ctx *, const uint8_t *row, int32_t len, int32_t stride, const double2 *, double alpha, double *ee, int32_t *ns)
{
    uint8_t pass;
    return oracle_rows(row, 1, stride, &len, alpha, ee, ns, &pass);
}

int mpbi_small_async(mpb_ctx *, const uint8_t *d_q, int64_t m, int64_t stride, const int32_t *d_len, double alpha,
                     double *d_ee, int32_t *d_ns, uint8_t *d_pass, uint8_t *, int32_t *, hipStream_t, const MpbSmallHost *host)
{
    int rc = oracle_rows(d_q, m, stride, d_len, alpha, d_ee, d_ns, d_pass);
    for (int64_t i = 0; i < m && !rc; i++)
        if (d_len[i] % 7 == 0) { d_pass[i] = 2; d_ee[i] = -12345.0; }    // "row budget missed": the broker must re-run it alone
    if (host && host->done)                                              // what k_small does last: one word per read
        for (int64_t i = m - 1; i >= 0; i--) __atomic_store_n(host->done + i, host->token, __ATOMIC_RELEASE);
    return rc;
}
int mpbi_wait_flags(const volatile uint32_t *, int64_t, uint32_t, hipStream_t) { return MPB_OK; }

// ---- the resident server (k_serve), as a host thread that keeps k_serve's side of the mailbox protocol: it polls the door
// words, answers with the oracle, stores the token to done[e], leaves on *stop or when its lifetime is over and says so in
// *exited.
}   // extern "C"
#include <chrono>
#include <mutex>
#include <thread>
#include <vector>
extern "C" {
void mpbi_small_params(double alpha, MpbDevParams *out)
{
    memset(out, 0, sizeof(*out));
    out->thr = 1 - alpha; out->uncert = 1.0; out->maxerrors = NAN; out->ambig_mode = 1;
}
// The parameters carry thr = fl(1 - alpha), and with direct serving they are made in the CLIENT's process -- which may be the
// real library (tests/test_reference_pipeline_with_dropins.py).  The stub's oracle wants alpha itself: an alpha whose
// fl(1 - alpha) is that thr gives the same arithmetic (the reference only ever uses 1 - alpha: bernoullimodule.c:244).
static double alpha_of_thr(double thr)
{
    double a = 1 - thr;
    for (int k = 0; k < 8 && 1 - a != thr; k++) a = nextafter(a, 1 - a < thr ? 0.0 : 1.0);
    return a;
}
int mpbi_serve_launch(mpb_ctx *, const MpbServeBox *boxp, uint32_t generation, uint32_t lifetime_ms, hipStream_t)
{
    const MpbServeBox box = *boxp;
    std::thread([box, generation, lifetime_ms] {
        const auto t0 = std::chrono::steady_clock::now();
        std::vector<uint32_t> last(box.n_ent);
        auto at = [](const void *base, int e, int64_t step) { return (char *)const_cast<void *>(base) + (int64_t)e * step; };
        for (int e = 0; e < box.n_ent; e++) last[e] = __atomic_load_n((uint32_t *)at(box.done, e, box.done_step), __ATOMIC_ACQUIRE);
        for (;;) {
            bool any = false;
            for (int e = 0; e < box.n_ent; e++) {
                const unsigned long long door = __atomic_load_n((const unsigned long long *)at(box.door, e, box.door_step), __ATOMIC_ACQUIRE);
                const uint32_t token = (uint32_t)door;
                if (token == last[e]) continue;
                any = true;
                int32_t len = (int32_t)(door >> 32);
                if (len < 0) len = 0;
                if (len > (int32_t)box.stride) len = (int32_t)box.stride;              // as the kernel: clamped, never trusted
                const double alpha = alpha_of_thr(((const MpbServePrm *)at(box.prm, e, box.prm_step))->p.thr);
                double ee = 0; int32_t ns = 0; uint8_t pass = 0;
                oracle_rows((const uint8_t *)at(box.q, e, box.q_step), 1, box.stride, &len, alpha, &ee, &ns, &pass);
                if (len % 7 == 0) { pass = 2; ee = -12345.0; }                   // "row budget missed", as the micro-batch stub
                *(double *)at(box.ee, e, box.ee_step) = ee; *(int32_t *)at(box.ns, e, box.ns_step) = ns; *(uint8_t *)at(box.pass, e, box.pass_step) = pass;
                __atomic_store_n((uint32_t *)at(box.done, e, box.done_step), token, __ATOMIC_RELEASE);
                last[e] = token;
            }
            if (any) continue;
            if (__atomic_load_n(box.stop, __ATOMIC_ACQUIRE)) break;
            if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(lifetime_ms)) break;
            std::this_thread::yield();
        }
        __atomic_store_n(box.exited, generation, __ATOMIC_RELEASE);
    }).detach();
    return MPB_OK;
}

// ---- the HIP entry points mpb_broker.cpp calls, on host memory ----
hipError_t hipSetDevice(int) { return hipSuccess; }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned int) { *s = (hipStream_t)malloc(8); return hipSuccess; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
hipError_t hipStreamQuery(hipStream_t) { return hipSuccess; }
hipError_t hipHostMalloc(void **p, size_t n, unsigned int) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipHostFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMalloc(void **p, size_t n) { *p = malloc(n); return *p ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipFree(void *p) { free(p); return hipSuccess; }
hipError_t hipMemset(void *p, int v, size_t n) { memset(p, v, n); return hipSuccess; }
hipError_t hipHostRegister(void *, size_t, unsigned int) { return getenv("MPB_STUB_NO_REGISTER") ? hipErrorInvalidValue : hipSuccess; }
hipError_t hipHostUnregister(void *) { return hipSuccess; }
hipError_t hipHostGetDevicePointer(void **d, void *h, unsigned int) { *d = h; return hipSuccess; }
hipError_t hipGetLastError(void) { return hipSuccess; }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
const char *hipGetErrorString(hipError_t) { return "stub"; }
}
