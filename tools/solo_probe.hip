// solo_probe.hip -- feasibility probe: how long does ONE workgroup need for P dependent phases over N agents,
// each phase = 3 dependent gathers + a store + __syncthreads?  (development microbenchmark)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void __launch_bounds__(1024) k_solo(int *a, const int *idx, int *b, int n, int phases) {
    for (int p = 0; p < phases; p++) {
        for (int i = threadIdx.x; i < n; i += 1024) {
            int j = idx[i];
            int v = a[j];
            int w = b[(v + j) % n];
            a[i] = w + p;
        }
        __syncthreads();
    }
}
__global__ void __launch_bounds__(256) k_multi(int *a, const int *idx, int *b, int n, int p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    int j = idx[i]; int v = a[j]; int w = b[(v + j) % n]; a[i] = w + p;
}
int main() {
    for (int n : {1250, 4000, 16000}) {
        int *a, *b, *idx;
        CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&idx, n * 4));
        std::vector<int> h(n); for (int i = 0; i < n; i++) h[i] = (i * 7919) % n;
        CK(hipMemcpy(idx, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(a, h.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b, h.data(), n * 4, hipMemcpyHostToDevice));
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        const int P = 30;
        float best1 = 1e9, best2 = 1e9;
        for (int r = 0; r < 5; r++) {
            hipEventRecord(e0); hipLaunchKernelGGL(k_solo, dim3(1), dim3(1024), 0, 0, a, idx, b, n, P); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best1) best1 = ms;
            hipEventRecord(e0); for (int p = 0; p < P; p++) hipLaunchKernelGGL(k_multi, dim3((n + 255) / 256), dim3(256), 0, 0, a, idx, b, n, p); hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1); if (ms < best2) best2 = ms;
        }
        printf("n=%5d  30 phases: one workgroup %.1f us   30 launches %.1f us\n", n, best1 * 1e3, best2 * 1e3);
    }
    return 0;
}
