// Development probe (GPU box): which store pattern writes HBM fastest?  (hipMemsetAsync reaches 6.7 TB/s on 2 GiB, the float4
// grid-stride fill of tools/membw.hip 6.0, k_render 5.6-5.7.)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v4f __attribute__((ext_vector_type(4)));
template <int NT> __device__ __forceinline__ void st(v4f *p, v4f x) { if (NT) __builtin_nontemporal_store(x, p); else *p = x; }
// A: grid-stride
template <int NT> __global__ void kA(v4f *out, size_t n4) { v4f x = {1, 2, 3, 4}; for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) st<NT>(out + i, x); }
// B: thread-contiguous U float4
template <int NT, int U> __global__ void kB(v4f *out, size_t n4) { v4f x = {1, 2, 3, 4}; for (size_t i = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) * U; i < n4; i += (size_t)gridDim.x * blockDim.x * U) { _Pragma("unroll") for (int u = 0; u < U; u++) st<NT>(out + i + u, x); } }
// C: wave-contiguous U KB (lane-linear, U stores 1 KB apart), grid-stride over waves
template <int NT, int U> __global__ void kC(v4f *out, size_t n4) { v4f x = {1, 2, 3, 4}; const size_t wave = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6; const int l = threadIdx.x & 63;
    for (size_t b = wave * 64 * U; b < n4; b += nw * 64 * U) { _Pragma("unroll") for (int u = 0; u < U; u++) st<NT>(out + b + u * 64 + l, x); } }
// D: block-contiguous chunk
template <int NT> __global__ void kD(v4f *out, size_t n4, size_t chunk4) { v4f x = {1, 2, 3, 4}; for (size_t c = blockIdx.x; c * chunk4 < n4; c += gridDim.x) { size_t base = c * chunk4; for (size_t i = threadIdx.x; i < chunk4 && base + i < n4; i += blockDim.x) st<NT>(out + base + i, x); } }
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
    for (int blk : {256, 512, 1024}) for (int per_cu : {2, 4, 8, 16, 64}) {
        int grid = 256 * per_cu * 256 / blk; if (grid < 256) continue;
        snprintf(nm, 80, "A grid-stride plain blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kA<0>, dim3(grid), dim3(blk), 0, 0, a, n4); });
        snprintf(nm, 80, "A grid-stride nt    blk=%d grid=%d", blk, grid); timeit(nm, [&] { hipLaunchKernelGGL(kA<1>, dim3(grid), dim3(blk), 0, 0, a, n4); });
    }
    for (int grid : {2048, 8192}) {
        snprintf(nm, 80, "B thread 64B plain grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kB<0, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        snprintf(nm, 80, "B thread 64B nt    grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kB<1, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        snprintf(nm, 80, "C wave 4KB plain grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kC<0, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        snprintf(nm, 80, "C wave 4KB nt    grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kC<1, 4>), dim3(grid), dim3(256), 0, 0, a, n4); });
        snprintf(nm, 80, "C wave 16KB nt   grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kC<1, 16>), dim3(grid), dim3(256), 0, 0, a, n4); });
        snprintf(nm, 80, "C wave 16KB plain grid=%d", grid); timeit(nm, [&] { hipLaunchKernelGGL((kC<0, 16>), dim3(grid), dim3(256), 0, 0, a, n4); });
    }
    for (size_t chunk : {65536ul, 262144ul, 1048576ul}) for (int grid : {1024, 4096}) {
        snprintf(nm, 80, "D block chunk %zuKB nt grid=%d", chunk / 1024, grid); timeit(nm, [&] { hipLaunchKernelGGL(kD<1>, dim3(grid), dim3(256), 0, 0, a, n4, chunk / 16); });
        snprintf(nm, 80, "D block chunk %zuKB plain grid=%d", chunk / 1024, grid); timeit(nm, [&] { hipLaunchKernelGGL(kD<0>, dim3(grid), dim3(256), 0, 0, a, n4, chunk / 16); });
    }
    return 0;
}
