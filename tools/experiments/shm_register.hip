// Run ON THE GPU BOX: can a POSIX shared-memory mapping be registered with the runtime and polled / written by a kernel?
// (the broker's slots served by k_serve directly, without the broker thread copying requests)
//   hipcc --offload-arch=gfx950 tools/experiments/shm_register.hip -o /tmp/shm_register && /tmp/shm_register
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <unistd.h>
#include <chrono>

__global__ void k_echo(volatile unsigned long long *door, unsigned int *done, double *out, int rounds)
{
    unsigned int last = 0;
    for (int r = 0; r < rounds;) {
        const unsigned long long d = __hip_atomic_load((const unsigned long long *)door, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const unsigned int tok = (unsigned int)d;
        if (tok == last) { __builtin_amdgcn_s_sleep(8); continue; }
        out[0] = (double)(d >> 32) * 0.5;
        __threadfence_system();
        __hip_atomic_store(done, tok, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        last = tok;
        r++;
    }
}

int main()
{
    const size_t bytes = 1 << 20;
    char name[64];
    snprintf(name, sizeof(name), "/shm_register_%d", (int)getpid());
    const int fd = shm_open(name, O_RDWR | O_CREAT | O_EXCL, 0600);
    if (fd < 0 || ftruncate(fd, bytes) != 0) { perror("shm"); return 1; }
    char *p = (char *)mmap(nullptr, bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
    shm_unlink(name);
    if (p == MAP_FAILED) { perror("mmap"); return 1; }
    memset(p, 0, bytes);
    hipError_t e = hipHostRegister(p, bytes, hipHostRegisterMapped);
    printf("hipHostRegister(shm mapping): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 2;
    void *d = nullptr;
    e = hipHostGetDevicePointer(&d, p, 0);
    printf("hipHostGetDevicePointer: %s (%p -> %p)\n", hipGetErrorString(e), (void *)p, d);
    if (e != hipSuccess) return 3;
    volatile unsigned long long *door = (volatile unsigned long long *)p;
    volatile unsigned int *done = (volatile unsigned int *)(p + 64);
    volatile double *out = (volatile double *)(p + 128);
    const int rounds = 20000;
    hipLaunchKernelGGL(k_echo, dim3(1), dim3(64), 0, 0, (volatile unsigned long long *)d, (unsigned int *)((char *)d + 64), (double *)((char *)d + 128), rounds);
    const auto t0 = std::chrono::steady_clock::now();
    int bad = 0;
    for (unsigned int t = 1; t <= (unsigned)rounds; t++) {
        __atomic_store_n((unsigned long long *)door, ((unsigned long long)t << 32) | t, __ATOMIC_RELEASE);
        while (__atomic_load_n((unsigned int *)done, __ATOMIC_ACQUIRE) != t) __builtin_ia32_pause();
        if (*out != t * 0.5) bad++;
    }
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / rounds;
    e = hipDeviceSynchronize();
    printf("round trips through the registered shm mapping: %.2f us each, %d wrong, sync %s\n", us, bad, hipGetErrorString(e));
    hipHostUnregister(p);
    return bad != 0;
}
