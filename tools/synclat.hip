// synclat.hip -- cost of a device->host round trip on the engine's stream (development microbenchmark)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void k_work(int *p) { if (threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
__global__ void k_signal(const int *p, volatile int *host_flag, int seq) {
    if (threadIdx.x == 0) { host_flag[1] = p[0]; __threadfence_system(); host_flag[0] = seq; }
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    int *d; CK(hipMalloc(&d, 256)); CK(hipMemset(d, 0, 256));
    int *h; CK(hipHostMalloc((void **)&h, 256, hipHostMallocDefault));
    volatile int *m; CK(hipHostMalloc((void **)&m, 256, hipHostMallocMapped | hipHostMallocCoherent)); m[0] = 0;
    int *md; CK(hipHostGetDevicePointer((void **)&md, (void *)m, 0));
    const int N = 2000;
    for (int mode = 0; mode < 3; mode++) {
        auto t0 = std::chrono::steady_clock::now();
        for (int i = 1; i <= N; i++) {
            hipLaunchKernelGGL(k_work, dim3(64), dim3(256), 0, s, d);
            if (mode == 0) { hipMemcpyAsync(h, d, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
            else if (mode == 1) { hipLaunchKernelGGL(k_signal, dim3(1), dim3(64), 0, s, d, (volatile int *)md, i); while (m[0] != i) {} }
            else { hipStreamSynchronize(s); }
        }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / N;
        printf("%-44s %.2f us per round trip\n", mode == 0 ? "kernel + memcpyAsync(4B) + streamSynchronize" : mode == 1 ? "kernel + signal kernel + spin on mapped flag" : "kernel + streamSynchronize only", us);
    }
    return 0;
}
