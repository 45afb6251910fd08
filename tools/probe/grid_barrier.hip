// Development probe (GPU box): what does a grid-wide barrier cost on the MI355X against a kernel boundary?  A step of a mid-size world is
// a dozen launches of 4-10 us each; a persistent launch with barriers between the phases only pays if a barrier is much
// cheaper than the boundary.  Persistent grids of 256 .. 2048 workgroups (1 .. 8 per CU), every workgroup touching `work` cache lines
// of a shared array between two barriers (the barrier must make them visible across the XCDs: device-scope release / acquire).
//   build: hipcc --offload-arch=gfx950 -O3 -o /tmp/grid_barrier tools/probe/grid_barrier.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ void grid_barrier(unsigned *ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
}
__global__ void __launch_bounds__(256) k_phases(unsigned *ctr, int *data, int n, int phases, int work) {
    for (int p = 0; p < phases; p++) {
        for (int k = 0; k < work; k++) {
            const int i = ((blockIdx.x * work + k) * 256 + threadIdx.x + p * 977) % n;
            data[i] += 1;
        }
        grid_barrier(ctr, (unsigned)(p + 1) * gridDim.x);
    }
}
__global__ void __launch_bounds__(256) k_one(int *data, int n, int p, int work) {
    for (int k = 0; k < work; k++) {
        const int i = ((blockIdx.x * work + k) * 256 + threadIdx.x + p * 977) % n;
        data[i] += 1;
    }
}
int main() {
    hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const int n = 1 << 22;
    int *data; CK(hipMalloc(&data, sizeof(int) * n)); CK(hipMemset(data, 0, sizeof(int) * n));
    unsigned *ctr; CK(hipMalloc(&ctr, 256));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int phases = 200;
    for (int work : {1, 8}) {
        for (int grid : {256, 512, 1024, 2048}) {
            float best = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipMemsetAsync(ctr, 0, 256, s));
                CK(hipEventRecord(e0, s));
                hipLaunchKernelGGL(k_phases, dim3(grid), dim3(256), 0, s, ctr, data, n, phases, work);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); best = ms < best ? ms : best;
            }
            float bestl = 1e9f;
            for (int rep = 0; rep < 3; rep++) {
                CK(hipEventRecord(e0, s));
                for (int p = 0; p < phases; p++) hipLaunchKernelGGL(k_one, dim3(grid), dim3(256), 0, s, data, n, p, work);
                CK(hipEventRecord(e1, s)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1)); bestl = ms < bestl ? ms : bestl;
            }
            printf("work %d lines/workgroup  grid %4d : %6.2f us per phase behind a grid barrier   %6.2f us per phase as a launch of its own\n", work, grid, best * 1e3f / phases, bestl * 1e3f / phases);
        }
    }
    return 0;
}
