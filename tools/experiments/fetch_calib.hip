// fetch_calib.hip -- calibrates rocprofv3 FETCH_SIZE for k_dp's access pattern (one row per lane,
// 64 bytes per lane and step, rows of 320 B) against a known byte count.
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- ./tools/fetch_calib
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int PIECE>   // bytes per lane per step: 16, 64 or 128
__global__ __launch_bounds__(256) void k_rows(const uint8_t *q, int64_t n, int stride, int len, int spin, unsigned *out)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const uint8_t *row = q + i * stride;
    unsigned acc = 0;
    double d = 1.0 + threadIdx.x;
    for (int off = 0; off < len; off += PIECE) {
#pragma unroll
        for (int p = 0; p < PIECE / 16; p++) {
            if (off + p * 16 < len) {
                const uint4 x = *reinterpret_cast<const uint4 *>(row + off + p * 16);
                acc ^= x.x ^ x.y ^ x.z ^ x.w;
            }
        }
        for (int s = 0; s < spin * (PIECE / 16); s++) d = d * 1.0000001 + 1e-9;   // stands in for the DP arithmetic
    }
    out[i] = acc ^ (unsigned)d;
}

int main()
{
    const int64_t n = 4000000; const int stride = 320, len = 304;
    uint8_t *q; unsigned *out;
    hipMalloc(&q, n * stride); hipMalloc(&out, n * 4);
    hipMemset(q, 1, n * stride);
    for (int spin : {0, 64}) {
        hipLaunchKernelGGL(k_rows<16>, dim3((n + 255) / 256), dim3(256), 0, 0, q, n, stride, len, spin, out);
        hipLaunchKernelGGL(k_rows<64>, dim3((n + 255) / 256), dim3(256), 0, 0, q, n, stride, len, spin, out);
        hipLaunchKernelGGL(k_rows<128>, dim3((n + 255) / 256), dim3(256), 0, 0, q, n, stride, len, spin, out);
        hipDeviceSynchronize();
    }
    printf("rows=%lld stride=%d len=%d: useful bytes per launch = %.3f GB, whole rows = %.3f GB, touched 128-B lines = %.3f GB\n",
           (long long)n, stride, len, n * 304.0 / 1e9, n * 320.0 / 1e9, n * 384.0 / 1e9);
    return 0;
}
