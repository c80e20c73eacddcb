// fp64_latency.hip -- how many independent FP64 chains does ONE wave per SIMD need to keep the vector ALU issuing every 4 cycles?
// (round 5: the narrow pass runs 2..4 rows per lane; its per-base chain is mul -> add.)
//   hipcc --offload-arch=gfx950 -O3 tools/experiments/fp64_latency.hip -o /tmp/fp64_latency && /tmp/fp64_latency
#include <hip/hip_runtime.h>
#include <cstdio>
#pragma clang fp contract(off)

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(double *out, int iters, double a, double b, long long *cyc)
{
    double v[NACC];
#pragma unroll
    for (int i = 0; i < NACC; i++) v[i] = 1.0 + threadIdx.x * 1e-9 + i;
    const long long t0 = clock64();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int u = 0; u < 8; u++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (MODE == 0) v[i] = v[i] * a;                          // dependent v_mul_f64
                if (MODE == 1) { double x = v[i] * a; v[i] = x + b; }    // dependent mul -> add
            }
    }
    const long long t1 = clock64();
    double s = 0;
#pragma unroll
    for (int i = 0; i < NACC; i++) s += v[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, int MODE>
void run(const char *name, int waves_per_simd)
{
    const int cus = 256, iters = 2048;
    double *out; long long *cyc, h = 0;
    hipMalloc(&out, sizeof(double) * 256 * cus * waves_per_simd);
    hipMalloc(&cyc, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(cus * waves_per_simd), dim3(256), 0, 0, out, 64, 1.0000001, 1e-9, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, MODE>), dim3(cus * waves_per_simd), dim3(256), 0, 0, out, iters, 1.0000001, 1e-9, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double ops_per_wave = (double)iters * 8 * NACC * (MODE == 1 ? 2 : 1);
    printf("%-22s chains=%d waves/SIMD=%d : %7.3f ms   %.2f ns per wave-instruction   (s_memtime ticks per instruction: %.2f)\n",
           name, NACC, waves_per_simd, ms, ms * 1e6 / (ops_per_wave * waves_per_simd), (double)h / ops_per_wave);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, 0>("mul chain", 1); run<2, 0>("mul chain", 1); run<3, 0>("mul chain", 1); run<4, 0>("mul chain", 1); run<8, 0>("mul chain", 1);
    run<1, 1>("mul->add chain", 1); run<2, 1>("mul->add chain", 1); run<3, 1>("mul->add chain", 1); run<4, 1>("mul->add chain", 1);
    run<1, 0>("mul chain", 2); run<1, 0>("mul chain", 3); run<1, 0>("mul chain", 4);
    run<1, 1>("mul->add chain", 2); run<1, 1>("mul->add chain", 3); run<1, 1>("mul->add chain", 4);
    return 0;
}
