// membw.hip -- write/copy bandwidth ceilings on the box (development microbenchmark, not part of the product)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>  // 0 plain, 1 nontemporal
__global__ void __launch_bounds__(256) k_fill(v4f *out, size_t n4, float v) {
    v4f x = {v, v, v, v};
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        if (MODE == 1) __builtin_nontemporal_store(x, out + i); else out[i] = x;
    }
}
// contiguous chunk per block (like the render kernel: each block writes its own 75 KB)
template <int MODE>
__global__ void __launch_bounds__(256) k_fill_chunk(v4f *out, size_t n4, int chunk4, float v) {
    v4f x = {v, v, v, v};
    size_t base = (size_t)blockIdx.x * chunk4;
    for (int i = threadIdx.x; i < chunk4 && base + i < n4; i += 256) {
        if (MODE == 1) __builtin_nontemporal_store(x, out + base + i); else out[base + i] = x;
    }
}
__global__ void __launch_bounds__(256) k_copy(const v4f *in, v4f *out, size_t n4) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i];
}
int main() {
    size_t bytes = 2ull << 30, n4 = bytes / 16;
    v4f *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto fn, double traffic) {
        fn(); hipDeviceSynchronize();
        float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-34s %8.3f ms  %8.1f GB/s\n", name, best, traffic / best / 1e6);
    };
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(a, 0, bytes, 0); }, (double)bytes);
    for (int blocks : {2048, 8192, 65536}) {
        char nm[64];
        snprintf(nm, 64, "fill plain grid=%d", blocks); timeit(nm, [&] { hipLaunchKernelGGL(k_fill<0>, dim3(blocks), dim3(256), 0, 0, a, n4, 1.f); }, (double)bytes);
        snprintf(nm, 64, "fill nt    grid=%d", blocks); timeit(nm, [&] { hipLaunchKernelGGL(k_fill<1>, dim3(blocks), dim3(256), 0, 0, a, n4, 1.f); }, (double)bytes);
    }
    for (int chunk4 : {1183 * 4, 4096, 16384}) {   // 1183*16/4 float4 = one 16-agent render tile
        int blocks = (int)((n4 + chunk4 - 1) / chunk4);
        char nm[64];
        snprintf(nm, 64, "chunk plain %dB/blk", chunk4 * 16); timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunk<0>, dim3(blocks), dim3(256), 0, 0, a, n4, chunk4, 1.f); }, (double)bytes);
        snprintf(nm, 64, "chunk nt    %dB/blk", chunk4 * 16); timeit(nm, [&] { hipLaunchKernelGGL(k_fill_chunk<1>, dim3(blocks), dim3(256), 0, 0, a, n4, chunk4, 1.f); }, (double)bytes);
    }
    timeit("copy float4 (r+w bytes)", [&] { hipLaunchKernelGGL(k_copy, dim3(8192), dim3(256), 0, 0, a, b, n4); }, 2.0 * bytes);
    return 0;
}
