// mpb_host_internal.h -- what the broker's translation unit (mpb_broker.cpp) needs from the C-ABI layer
// (mpb_api.cpp).  Not installed; none of these symbols is exported (the version script keeps `mpb_*` only).
#ifndef MPB_HOST_INTERNAL_H
#define MPB_HOST_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

struct mpb_ctx;
struct MpbSmallHost;

extern "C" {      // (defined inside mpb_api.cpp's extern "C" block; hidden: the version script exports `mpb_*` only)

int mpbi_fail(int code, const char *msg);                // sets mpb_last_error() of the calling thread, returns code
int mpbi_ctx_device(const mpb_ctx *c);
int mpbi_check_one_read(const char *contig, const int32_t *contig_quals, int32_t len, double alpha, const void *ee, const void *ns);
int mpbi_pack_one_read(const char *contig, const int32_t *quals, int32_t len, bool poisson, uint8_t *row, int32_t row_bytes,
                       double2 *h, bool *priv);
int mpbi_run_packed_read(mpb_ctx *c, const uint8_t *row, int32_t len, int32_t stride, const double2 *h, double alpha,
                         double *ee, int32_t *ns);
int mpbi_small_async(mpb_ctx *c, const uint8_t *d_q, int64_t m, int64_t stride, const int32_t *d_len, double alpha,
                     double *d_ee, int32_t *d_ns, uint8_t *d_pass, uint8_t *d_cls, int32_t *d_ident, hipStream_t s,
                     const MpbSmallHost *host /* nullptr, or the device scratch + completion flags that go with inputs and
                                                 outputs in pinned host memory (mpb_internal.h) */);
int mpbi_wait_flags(const volatile uint32_t *done, int64_t n, uint32_t token, hipStream_t s);
// the resident one-read server (k_serve): parameters of a request with this alpha (what mpbi_small_async gives its launch);
// one launch of the server over `box` (generation, lifetime); is a launch on `s` still out?
struct MpbServeBox;
struct MpbDevParams;
void mpbi_small_params(double alpha, MpbDevParams *out);
int mpbi_serve_launch(mpb_ctx *c, const MpbServeBox *box, uint32_t generation, uint32_t lifetime_ms, hipStream_t s);

}

#endif
