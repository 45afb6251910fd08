// Development probe (GPU box), part 2: more store patterns against hipMemsetAsync's 6.6 TB/s.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
// E: grid-stride, U stores in flight per thread
template <int U> __global__ void kE(v4f *out, size_t n4) { v4f x = {1, 2, 3, 4}; const size_t T = (size_t)gridDim.x * blockDim.x; size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i + (U - 1) * T < n4; i += U * T) { _Pragma("unroll") for (int u = 0; u < U; u++) out[i + u * T] = x; }
    for (; i < n4; i += T) out[i] = x; }
// G: raw buffer stores with cache-policy bits
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(void *p) { return __builtin_amdgcn_make_buffer_rsrc(p, 0, 0x7FFFFFFF, 0x00020000); }
template <int AUX> __global__ void kG(v4f *out, size_t n4) { v4i x = {1, 2, 3, 4};
    for (size_t base = (size_t)blockIdx.x * 65536; base < n4; base += (size_t)gridDim.x * 65536) {          // 1 MiB windows (32-bit buffer offsets)
        __amdgpu_buffer_rsrc_t rs = make_rsrc(out + base);
        for (unsigned i = threadIdx.x; i < 65536u; i += blockDim.x) __builtin_amdgcn_raw_buffer_store_b128(x, rs, i * 16u, 0, AUX); } }
int main() {
    size_t bytes = 2ull << 30, n4 = bytes / 16;
    v4f *a; if (hipMalloc(&a, bytes) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto timeit = [&](const char *name, auto fn) {
        fn(); hipDeviceSynchronize(); float best = 1e9;
        for (int r = 0; r < 5; r++) { hipEventRecord(e0); fn(); hipEventRecord(e1); hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms; }
        printf("%-44s %7.3f ms %7.0f GB/s\n", name, best, bytes / best / 1e6);
    };
    char nm[80];
    timeit("hipMemsetAsync", [&] { hipMemsetAsync(a, 0, bytes, 0); });
    timeit("hipMemsetD32Async", [&] { hipMemsetD32Async((hipDeviceptr_t)a, 7, bytes / 4, 0); });
    for (int blk : {256, 512, 1024}) for (int grid : {256, 512, 768, 1024}) {
        snprintf(nm, 80, "E U=1 blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kE<1>, dim3(grid), dim3(blk), 0, 0, a, n4); });
        snprintf(nm, 80, "E U=2 blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kE<2>, dim3(grid), dim3(blk), 0, 0, a, n4); });
        snprintf(nm, 80, "E U=4 blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kE<4>, dim3(grid), dim3(blk), 0, 0, a, n4); });
        snprintf(nm, 80, "E U=8 blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kE<8>, dim3(grid), dim3(blk), 0, 0, a, n4); });
    }
    for (int grid : {256, 512, 2048}) {
        snprintf(nm, 80, "G buffer aux=0  grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<0>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=1  grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<1>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=2  grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<2>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=3  grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<3>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=16 grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<16>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=17 grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<17>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=18 grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<18>, dim3(grid), dim3(512), 0, 0, a, n4); });
        snprintf(nm, 80, "G buffer aux=19 grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL(kG<19>, dim3(grid), dim3(512), 0, 0, a, n4); });
    }
    return 0;
}
