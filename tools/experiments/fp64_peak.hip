// fp64_peak.hip -- microbenchmark: what does the FP64 VALU of this MI355X sustain?
// Used to anchor DESIGN.md's "binding roof" claim.  hipcc --offload-arch=gfx950 -O3 tools/fp64_peak.hip -o tools/fp64_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#pragma clang fp contract(off)

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b)
{
    double v[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) v[i] = 1.0 + threadIdx.x * 1e-9 + i;
    unsigned x = threadIdx.x;
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int i = 0; i < NACC; i++) {
            if (MODE == 0) v[i] = v[i] * a;                       // v_mul_f64 only
            if (MODE == 1) v[i] = v[i] * a + b;                   // mul + add (contract off)
            if (MODE == 2) { v[i] = v[i] * a; x = (x << 4) ^ (x >> 3); }   // 1 mul + 2 int ops
            if (MODE == 3) v[i] = __builtin_fma(v[i], a, b);      // fma
        }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + x;
}

template <int NACC, int MODE>
void run(const char *name, int blocks_per_cu, double ops_per_iter_per_acc)
{
    int cus = 256, iters = 4096;
    double *out;
    hipMalloc(&out, sizeof(double) * 256 * cus * blocks_per_cu);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, 64, 1.0000001, 1e-9);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double ops = (double)cus * blocks_per_cu * 256 * iters * NACC * ops_per_iter_per_acc;
    printf("%-28s acc=%2d blocks/CU=%d : %8.3f ms  %.3e FP64 lane-ops/s\n", name, NACC, blocks_per_cu, ms, ops / (ms * 1e-3));
    hipFree(out);
}

int main()
{
    for (int b = 1; b <= 8; b *= 2) run<8, 0>("v_mul_f64", b, 1);
    for (int b = 1; b <= 8; b *= 2) run<8, 1>("mul+add (2 ops)", b, 2);
    for (int b = 1; b <= 8; b *= 2) run<8, 3>("v_fma_f64 (1 op)", b, 1);
    for (int b = 2; b <= 8; b *= 2) run<8, 2>("mul + 2 int VALU", b, 1);
    run<16, 1>("mul+add (2 ops)", 4, 2);
    return 0;
}
