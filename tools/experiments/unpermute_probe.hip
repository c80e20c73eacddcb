// What VERDICT r1 item 6(a) would add to a step: results written by k_dp in class (sorted) order and
// brought back to read order by a separate kernel -- coalesced reads of the inverse permutation, gathers
// of 8 + 1 bytes from the sorted arrays (90 MB: Infinity-Cache resident), coalesced writes.  The
// permutation has the shape of the real one: 32 classes, stable inside a class.
//   hipcc -O3 --offload-arch=gfx950 tools/unpermute_probe.hip -o tools/unpermute_probe && tools/unpermute_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <numeric>
#include <algorithm>

__global__ __launch_bounds__(256) void k_unpermute(const int *__restrict__ inv, const double *__restrict__ ee_s,
                                                   const unsigned char *__restrict__ pass_s, double *__restrict__ ee,
                                                   unsigned char *__restrict__ pass, long n)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int k = inv[i];
    ee[i] = ee_s[k];
    pass[i] = pass_s[k];
}

int main()
{
    const long n = 10000000;
    std::vector<int> cls(n), inv(n);
    // class mix of config 2 (row-budget histogram of profiles/r02_bench.json, rounded)
    const double share[] = {0.256, 0.174, 0.101, 0.075, 0.057, 0.046, 0.038, 0.033, 0.054, 0.044, 0.038, 0.014, 0.005, 0.004,
                            0.004, 0.007, 0.007, 0.007, 0.008, 0.009, 0.010, 0.009};
    const int nc = sizeof(share) / sizeof(share[0]);
    unsigned long long s = 12345;
    std::vector<long> count(nc, 0);
    for (long i = 0; i < n; i++) {
        s = s * 6364136223846793005ull + 1442695040888963407ull;
        double u = (double)(s >> 11) / 9007199254740992.0, acc = 0;
        int c = nc - 1;
        for (int k = 0; k < nc; k++) { acc += share[k]; if (u < acc) { c = k; break; } }
        cls[i] = c; count[c]++;
    }
    std::vector<long> base(nc, 0);
    for (int k = 1; k < nc; k++) base[k] = base[k - 1] + count[k - 1];
    for (long i = 0; i < n; i++) inv[i] = (int)base[cls[i]]++;
    int *d_inv; double *d_es, *d_e; unsigned char *d_ps, *d_p;
    hipMalloc(&d_inv, n * 4); hipMalloc(&d_es, n * 8); hipMalloc(&d_e, n * 8); hipMalloc(&d_ps, n); hipMalloc(&d_p, n);
    hipMemcpy(d_inv, inv.data(), n * 4, hipMemcpyHostToDevice);
    hipMemset(d_es, 0, n * 8); hipMemset(d_ps, 0, n);
    // something large in between, as the 3.2 GB matrix pass of the DP would be: evict the sorted arrays from L2
    char *d_big; hipMalloc(&d_big, 1l << 30);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    float best = 1e9f;
    for (int rep = 0; rep < 10; rep++) {
        hipMemset(d_big, rep, 1l << 30);
        hipEventRecord(a);
        hipLaunchKernelGGL(k_unpermute, dim3((n + 255) / 256), dim3(256), 0, 0, d_inv, d_es, d_ps, d_e, d_p, n);
        hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (ms < best) best = ms;
    }
    printf("un-permute of %ld results (8 + 1 bytes each) by a separate kernel: %.3f ms (best of 10)\n", n, best);
    return 0;
}
