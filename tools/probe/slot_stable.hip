// slot_stable.hip -- development A/B (not part of the product): what clear_dead would cost with SLOT-STABLE agents, against the compaction
// the engine runs ("Next round" item 4).
//
//   compaction (step.hip: k_clear_compact, as the engine runs it at 2 x 400k agents):
//     every survivor moves to its rank: x, y, id, hp, last_action, next_reward -> last_reward copied to the alternate arrays (7 x 4 B read,
//     8 x 4 B written), its map cell is rewritten with its new reference (4 B at a random address), the in-place state (dead, last_op,
//     op_obj, pend) goes back to rest.  `compact` below is that loop over synthetic arrays of the same shapes.
//   slot-stable:
//     nobody moves; the rewards roll over in place, the in-place state goes back to rest, and a table rank -> slot is written (the
//     "logical index" every getter and the render would read through: one more dependent load there, 4 B per agent per launch).
//     `stable` below.  What the probe cannot show is the price on the other side: every kernel that walks a group by logical index
//     (the two renders: x / y of an agent become gathers through the table; get_reward, get_pos, get_alive, set_action: scatters),
//     and dead slots that stay in every per-agent launch of the step until the group is rebuilt.
// Prints both kernels' times for deaths of 0.3 % and 3 % of the agents per step.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int T = 256, ITEMS = 2, TILE = T * ITEMS;
struct Arrays { int *x, *y, *id, *la, *op_obj, *pend; float *hp, *nr, *lr; unsigned char *dead, *last_op; };

__device__ __forceinline__ int wave_rank(bool p, int &tot) {
    const unsigned long long m = __ballot(p);
    tot = __popcll(m);
    return __popcll(m & ((1ull << (threadIdx.x & 63)) - 1));
}
__device__ __forceinline__ int block_prefix(const int *sums, int b) {
    __shared__ int s_p[4];
    int t = 0;
    for (int k = threadIdx.x; k < b; k += T) t += sums[k];
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if ((threadIdx.x & 63) == 0) s_p[threadIdx.x >> 6] = t;
    __syncthreads();
    const int tot = s_p[0] + s_p[1] + s_p[2] + s_p[3];
    __syncthreads();
    return tot;
}
__global__ void __launch_bounds__(T) k_count(Arrays G, int n, int *sums) {
    __shared__ int s_w[4];
    int cnt = 0;
    for (int k = 0; k < ITEMS; k++) { const int i = blockIdx.x * TILE + k * T + threadIdx.x; cnt += __popcll(__ballot(i < n && !G.dead[i])); }
    if ((threadIdx.x & 63) == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) sums[blockIdx.x] = s_w[0] + s_w[1] + s_w[2] + s_w[3];
}
template <bool STABLE>
__global__ void __launch_bounds__(T) k_clear(Arrays G, Arrays D, int n, const int *sums, int *occ, int w, int *slot_of, float step_reward) {
    __shared__ int s_w[ITEMS][4];
    const int base = blockIdx.x * TILE, wave = threadIdx.x >> 6;
    const int before_blocks = block_prefix(sums, blockIdx.x);
    bool p[ITEMS]; int r[ITEMS];
    for (int k = 0; k < ITEMS; k++) {
        const int i = base + k * T + threadIdx.x;
        bool alive = false;
        if (i < n) {
            const bool d = G.dead[i];
            if (d) G.dead[i] = 0;
            G.last_op[i] = 11; G.op_obj[i] = -1; G.pend[i] = 0;
            alive = !d;
        }
        p[k] = alive;
    }
    for (int k = 0; k < ITEMS; k++) { int tot; r[k] = wave_rank(p[k], tot); if ((threadIdx.x & 63) == 0) s_w[k][wave] = tot; }
    __syncthreads();
    int run = before_blocks;
    for (int k = 0; k < ITEMS; k++) {
        int before = 0, all = 0;
        for (int v = 0; v < 4; v++) { const int t = s_w[k][v]; all += t; if (v < wave) before += t; }
        if (p[k]) {
            const int i = base + k * T + threadIdx.x, rank = run + before + r[k];
            if (STABLE) {
                G.lr[i] = G.nr[i]; G.nr[i] = step_reward;
                slot_of[rank] = i;
            } else {
                const int x = G.x[i], y = G.y[i];
                D.x[rank] = x; D.y[rank] = y; D.id[rank] = G.id[i]; D.hp[rank] = G.hp[i]; D.la[rank] = G.la[i];
                D.lr[rank] = G.nr[i]; D.nr[rank] = step_reward;
                occ[y * w + x] = rank;
            }
        }
        run += all;
    }
}

int main() {
    const int n = 400000, w = 1000, groups = 2;
    Arrays G[2], D[2];
    int *sums[2], *occ, *slot_of[2];
    auto alloc = [&](Arrays &A) {
        CK(hipMalloc(&A.x, n * 4)); CK(hipMalloc(&A.y, n * 4)); CK(hipMalloc(&A.id, n * 4)); CK(hipMalloc(&A.la, n * 4)); CK(hipMalloc(&A.op_obj, n * 4)); CK(hipMalloc(&A.pend, n * 4));
        CK(hipMalloc(&A.hp, n * 4)); CK(hipMalloc(&A.nr, n * 4)); CK(hipMalloc(&A.lr, n * 4)); CK(hipMalloc(&A.dead, n)); CK(hipMalloc(&A.last_op, n));
        return 0;
    };
    for (int g = 0; g < groups; g++) { if (alloc(G[g]) || alloc(D[g])) return 1; CK(hipMalloc(&sums[g], 4096 * 4)); CK(hipMalloc(&slot_of[g], n * 4)); }
    CK(hipMalloc(&occ, w * w * 4));
    std::vector<int> hx(n), hy(n);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (double frac : {0.003, 0.03}) {
        srand(7);
        std::vector<unsigned char> hd(n);
        for (int g = 0; g < groups; g++) {
            for (int i = 0; i < n; i++) { hx[i] = rand() % w; hy[i] = rand() % w; hd[i] = (rand() % 100000) < frac * 100000; }
            CK(hipMemcpy(G[g].x, hx.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(G[g].y, hy.data(), n * 4, hipMemcpyHostToDevice));
        }
        const int nb = (n + TILE - 1) / TILE;
        for (int stable = 0; stable < 2; stable++) {
            float best = 1e9f;
            for (int rep = 0; rep < 12; rep++) {
                for (int g = 0; g < groups; g++) CK(hipMemcpy(G[g].dead, hd.data(), n, hipMemcpyHostToDevice));
                for (int g = 0; g < groups; g++) hipLaunchKernelGGL(k_count, dim3(nb), dim3(T), 0, 0, G[g], n, sums[g]);     // (k_strike leaves these in the engine)
                CK(hipDeviceSynchronize());
                hipEventRecord(e0);
                for (int g = 0; g < groups; g++) {          // (the engine: one launch, blockIdx.y = group; two here -- the same work)
                    if (stable) hipLaunchKernelGGL(k_clear<true>, dim3(nb), dim3(T), 0, 0, G[g], D[g], n, sums[g], occ, w, slot_of[g], -0.005f);
                    else hipLaunchKernelGGL(k_clear<false>, dim3(nb), dim3(T), 0, 0, G[g], D[g], n, sums[g], occ, w, slot_of[g], -0.005f);
                }
                hipEventRecord(e1); CK(hipEventSynchronize(e1));
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (ms < best) best = ms;
            }
            printf("2 x %d agents, %.1f %% dead: %-12s %7.1f us\n", n, frac * 100, stable ? "slot-stable" : "compaction", best * 1e3);
        }
    }
    return 0;
}
