// Development probe (GPU box): what does HBM give when every wave writes one agent-sized contiguous chunk (an observation row:
// 1620 B / 2000 B for test_1m's prey / predators, 4732 B for battle) at a PERMUTED position of the output tensor -- the store
// geometry of a render that walks the agents in spatial order but writes every row at its agent index?  Against the same
// kernel with the identity permutation (chunks in order: the geometry of k_render).
//   build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/scatter_chunks tools/probe/scatter_chunks.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <numeric>
#include <random>
#include <string>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

// wave `w` of the launch writes chunks w, w + W, ...: chunk j lies at floats [perm[j] * S, perm[j] * S + S)
template <bool NT>
__global__ void __launch_bounds__(256) k_chunks(float *out, const int *perm, int n, int S) {
    const int lane = threadIdx.x & 63;
    const int wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, waves = (gridDim.x * blockDim.x) >> 6;
    for (int j = wave; j < n; j += waves) {
        const size_t f0 = (size_t)perm[j] * S;
        const int head = (int)((4 - (f0 & 3)) & 3);                 // floats before the first 16-byte boundary
        const int nq = (S - head) >> 2, tail = S - head - (nq << 2);
        float *base = out + f0;
        if (lane < head) base[lane] = 1.0f;
        v4f x = {1, 2, 3, 4};
        v4f *q = (v4f *)(base + head);
        for (int i = lane; i < nq; i += 64) { if (NT) __builtin_nontemporal_store(x, q + i); else q[i] = x; }
        if (lane < tail) base[head + (nq << 2) + lane] = 2.0f;
    }
}

int main(int argc, char **argv) {
    const size_t bytes = 1ull << 30;
    float *a; if (hipMalloc(&a, bytes + 4096) != hipSuccess) return 1;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    std::mt19937 rng(7);
    // (argv[1] == "padded": test_1m's rows padded to whole 128-byte lines -- 1620 -> 1664 B, 2000 -> 2048 B -- beside the ragged rows: round 6)
    const bool padded = argc > 1 && std::string(argv[1]) == "padded";
    for (int S : padded ? std::vector<int>{405, 416, 500, 512} : std::vector<int>{405, 500, 1183, 1024, 2048}) {
        const int n = (int)(bytes / 4 / S);
        std::vector<int> id(n), rnd(n), tiled(n);
        std::iota(id.begin(), id.end(), 0);
        rnd = id; std::shuffle(rnd.begin(), rnd.end(), rng);
        // "blocked": chunks permuted inside blocks of 4096 consecutive chunks (a partially sorted population)
        tiled = id; for (int b = 0; b + 4096 <= n; b += 4096) std::shuffle(tiled.begin() + b, tiled.begin() + b + 4096, rng);
        int *d; hipMalloc(&d, n * sizeof(int));
        for (auto &pr : {std::make_pair("in order", &id), std::make_pair("random", &rnd), std::make_pair("random within 4096", &tiled)}) {
            hipMemcpy(d, pr.second->data(), n * sizeof(int), hipMemcpyHostToDevice);
            for (int grid : {1280, 2048, 8192}) for (int nt = 0; nt < 2; nt++) {
                float best = 1e9;
                for (int r = 0; r < 4; r++) {
                    hipEventRecord(e0);
                    if (nt) hipLaunchKernelGGL(k_chunks<true>, dim3(grid), dim3(256), 0, 0, a, d, n, S);
                    else hipLaunchKernelGGL(k_chunks<false>, dim3(grid), dim3(256), 0, 0, a, d, n, S);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    float ms; hipEventElapsedTime(&ms, e0, e1); if (r && ms < best) best = ms;
                }
                printf("chunk %5d B  %-20s grid %5d %s  %7.3f ms %7.0f GB/s\n", S * 4, pr.first, grid, nt ? "nt   " : "plain", best, (double)n * S * 4 / best / 1e6);
            }
        }
        hipFree(d);
    }
    return 0;
}
