// pipe.hip -- MANY environments per launch through the pipeline of plain games (env_cycle_many on worlds beyond the one-launch step)
//
// A world of a few thousand to a few hundred thousand agents steps through a dozen launches (step.hip: set_action, the shuffle's draws,
// k_plain_rank, rounds of k_plain_eval, k_strike, k_plain_commit, the compaction of clear_dead) of 3-30 us each, and most of that is the
// launch itself and the latency chain inside it, not work: BASELINE config 2 (battle 200 x 200, 2 x 2000) is 16 launches around 20 us of
// work, config 4 (gather 500 x 500, 100k agents) 13 around 70.  Reinforcement-learning callers run MANY such worlds per GPU; here the same
// kernel bodies (plain_dev.h) are launched ONCE per phase for all environments of an env_cycle_many call:
//   grid = (tiles of the largest group, groups, environments), the environment's description read from a device array of PipeItem
// so that n environments cost one chain of launches instead of n -- and the chain has no host round trip in it: which groups compact, and
// to what size, k_pipe_clear / k_pipe_finish read from the death counters the step left (the host learns the same numbers from the
// reports, which the batch's last workgroup sends to pinned memory in one piece behind ONE system-scope release: every environment
// publishing its own report -- a release, i.e. a write-back of the L2's dirty lines, each -- made the first version's k_pipe_commit 43 us
// for 32 environments; it is 11 now).
// GridWorld.cc:292-401 (observations), :403-454 (set_action), :456-631 (step), :694-704 (get_reward), :633-665 (clear_dead) -- bit-identical
// to the same environments stepped one by one (the bodies are the same).
#include "plain_dev.h"
#include "render_sweep_dev.h"

namespace magent_amd {

// (glob(): kernels_dev.h -- every pointer of an item is one that the kernel reads from memory)
__device__ __forceinline__ void globalize(RenderArgs &R) { R.mini = glob(R.mini); R.view = glob(R.view); R.feat = glob(R.feat); }
__device__ __forceinline__ RenderWorld pipe_render_world(const PipeItem &it, int g) {
    RenderWorld V;
    V.w = it.W.w; V.h = it.W.h; V.G = it.W.G; V.viewcell = glob(it.W.viewcell); V.mask = glob(it.W.mask); V.grp = it.W.grp[g]; V.type = it.W.type[g];
    V.grp = glob_group(V.grp);
    return V;
}

// (an item is read through scalar loads: the index is the workgroup's, nothing in these launches writes the array)
#define PIPE_ITEM() const PipeItem &it = items[blockIdx.z]; const int g = blockIdx.y; if (g >= it.W.G) return

// the observations of every environment that did not render them itself: blockIdx.y = environment * slots + slot (as k_render_batch)
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_pipe_render(const PipeItem *__restrict__ items, int slots) {
    const int e = blockIdx.y / slots, k = blockIdx.y - e * slots;
    const PipeItem &it = items[e];
    if (k >= it.M.n || (int)blockIdx.x >= it.M.blocks[k]) return;
    const RenderArgs R = it.M.R[k];
    const RenderPlan P = it.M.P[k];
    // (NOT through glob(): with GLOBAL instructions this kernel compiles to 97 VGPRs -- 5 waves per SIMD -- against 50 -- 7 -- with the FLAT ones its
    // generic pointers give it, and its independent workgroups live on occupancy: measured on one box, FLAT / GLOBAL in turn, 32 worlds of 2 x 2000:
    // 0.232-0.241 / 0.252 ms per round, 128 worlds 0.72-0.74 / 0.80-0.81, profiles/r06_raw/pipe_ab_global_address_sweep.txt)
    RenderWorld V;
    V.w = it.W.w; V.h = it.W.h; V.G = it.W.G; V.viewcell = it.W.viewcell; V.mask = it.W.mask; V.grp = it.W.grp[R.g]; V.type = it.W.type[R.g];
    // (plain games: no turn_mode)
    if (it.W.vc_packed) render_block<true, true, 1, true, false>(V, R, P, blockIdx.x, it.M.blocks[k]);
    else render_block<true, true, 1, false, false>(V, R, P, blockIdx.x, it.M.blocks[k]);
}
// ... or, when every observed group of the batch has the battle shape [wall | has, hp, minimap | has, hp, minimap] (two groups, packed view
// cells): the sweeping kernel (render_sweep_dev.h), `sweep` workgroups per (environment, group) segment -- ~256 over the launch, the
// geometry it has on its own -- + the feature rows' workgroups.  (Round 6 first measured it level with the generic workgroups and left it
// off: its loads and stores were FLAT instructions then -- see glob() above -- and its workgroups sat on half of the XCDs -- below.)
// (a flat grid: the sweeping workgroups of all segments first -- `sweep` consecutive ones per segment, so that consecutive workgroup numbers,
// which the dispatcher deals out over the eight XCDs in turn, are the sweeping ones: with (sweep + feature workgroups) per segment on one
// axis, sweep = 4 and 68 feature workgroups put every sweeping workgroup of every segment on XCDs 0..3 -- then the feature rows' workgroups)
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_pipe_render_sweep(const PipeItem *__restrict__ items, int slots, int sweep, int segs, int feat_blocks) {
    int seg, bx;
    if ((int)blockIdx.x < segs * sweep) { seg = blockIdx.x / sweep; bx = blockIdx.x - seg * sweep; }
    else { const int f = blockIdx.x - segs * sweep; seg = f / feat_blocks; bx = sweep + (f - seg * feat_blocks); }
    const int e = seg / slots, k = seg - e * slots;
    const PipeItem &it = items[e];
    if (k >= it.M.n) return;
    RenderArgs R = it.M.R[k];
    globalize(R);
    const RenderPlan P = it.M.P[k];
    const RenderWorld V = pipe_render_world(it, R.g);
    render_sweep2_body<false, 2, 2, true>(V, R, P, sweep, bx, sweep + feat_blocks);
}
__global__ void __launch_bounds__(SCAN_THREADS) k_pipe_set_action(const PipeItem *__restrict__ items) {
    PIPE_ITEM();
    if (!it.actions[g] || (int)(blockIdx.x * SCAN_TILE) >= it.W.grp[g].n) return;
    set_action_tile_body(it.W, g, it.actions[g], it.call_base[g], it.sums, it.wpre, it.P.off[g]);
}
__global__ void __launch_bounds__(256) k_pipe_draw(const PipeItem *__restrict__ items) {
    const PipeItem &it = items[blockIdx.z];
    if ((int)(blockIdx.x * 256) >= it.n_max) return;
    shuffle_draw_launch_body(it.W.counters, it.B.j, it.B.head, it.B.first, it.B.link, nullptr, 0, it.powtab, 1);
}
__global__ void __launch_bounds__(256) k_pipe_rank(const PipeItem *__restrict__ items) {
    PIPE_ITEM();
    plain_rank_body(it.W, it.PW, it.ptab, it.B, it.sums, it.wpre, it.P);
}
__global__ void __launch_bounds__(256) k_pipe_eval(const PipeItem *__restrict__ items, int round, int flag) {
    PIPE_ITEM();
    plain_eval_body(it.W, it.PW, it.ptab, it.gtab, it.ttab, round, flag, it.B.head, it.B.first);
}
// tests only (MAGENT_TUNE attack_pairs=0): no optimistic round at all -- every environment's attack phase is left open for the host
__global__ void k_pipe_force_open(const PipeItem *__restrict__ items) {
    if (threadIdx.x == 0) items[blockIdx.x].W.counters[CTR_OPEN_ATTACK] = 1;
}
__global__ void __launch_bounds__(256) k_pipe_strike(const PipeItem *__restrict__ items) {
    PIPE_ITEM();
    strike_body(it.W, it.PW, it.ptab, it.gtab, it.ttab, it.R);
}
__global__ void __launch_bounds__(256) k_pipe_commit(const PipeItem *__restrict__ items) {
    PIPE_ITEM();
    plain_commit_body(it.W, it.PW, it.rec, it.seq, REPORT_DEVICE);      // (the report goes to device memory here: k_pipe_finish sends them all)
}
// dead_ct + movers taken in of group g, as the step left them (Env::step_end adds up the same counters from the report)
__device__ __forceinline__ int pipe_gone(const int *counters, int g) {
    int gone = counters[CTR_TAKEN + g];
#pragma unroll
    for (int k = 0; k < DEAD_SLOTS; k++) gone += counters[dead_slot(g, k)];
    return gone;
}
// get_reward + clear_dead's compaction (Agent::init_reward alone for groups without deaths), the next minimap's histogram
__global__ void __launch_bounds__(SCAN_THREADS) k_pipe_clear(const PipeItem *__restrict__ items) {
    PIPE_ITEM();
    if (attack_open(it.W)) return;             // the host finishes this environment's step, then clears it by launches of its own
    const int n = it.W.grp[g].n;
    if (n == 0) { if (blockIdx.x == 0 && threadIdx.x == 0) it.newn[g] = 0; return; }
    const int gone = pipe_gone(it.W.counters, g);
    if (blockIdx.x == 0 && threadIdx.x == 0) it.newn[g] = n - gone;
    clear_compact_body(it.W, it.A, gone > 0 ? 2 : 1, it.alive_sums, it.Mi, it.counts, it.rewards[g], it.group_reward[g]);
}
// ... then the death counters, the device copies of the group / type tables as the compaction leaves them (the double-buffered arrays of a
// compacted group have changed places: ClearArgs::dst are the current ones), the division of the next minimap -- and, from the LAST
// environment to get there, every environment's report to the host in one piece and the word the host waits for
__global__ void __launch_bounds__(256) k_pipe_finish(const PipeItem *__restrict__ items, PipeCtl C) {
    const PipeItem &it = items[blockIdx.z];
    const int NG = it.W.G;
    const bool open = attack_open(it.W);
    if (blockIdx.x == 0) {
        if (!open) {
            if (threadIdx.x < MAXG) {
                const int q = threadIdx.x;
                GroupDev N = it.W.grp[q];
                if (q < NG) {
                    const int n_new = it.newn[q];
                    if (n_new != N.n) {            // compacted: mode 2
                        const ClearArgs::Alt D = it.A.dst[q];
                        N.x = D.x; N.y = D.y; N.id = D.id; N.last_action = D.last_action; N.hp = D.hp; N.next_reward = D.next_reward;
                        N.last_reward = D.last_reward; N.absorbed = D.absorbed; N.dir = D.dir;
                    }
                    N.n = n_new;
                }
                it.gtab_out[q] = N; it.ttab_out[q] = it.W.type[q];
            }
            for (int q = 0; q < NG; q++) {         // (zero where the group did not compact: nothing to reset, nothing lost)
                if (threadIdx.x < DEAD_SLOTS) it.W.counters[dead_slot(q, threadIdx.x)] = 0;
                if (threadIdx.x == 0) it.W.counters[CTR_TAKEN + q] = 0;
            }
        }
        // every output of this environment's cycle -- rewards, compacted arrays (the launch before), tables -- is written
        __threadfence();
        __syncthreads();
        __shared__ int s_last;
        if (threadIdx.x == 0) s_last = atomicAdd(C.ticket, 1) == C.n_env - 1;
        __syncthreads();
        if (s_last) {
            __threadfence();
            const int words = PIPE_REPORT_BYTES / 16;                      // 16-byte pieces of one report
            for (int k = threadIdx.x; k < C.n_env * words; k += blockDim.x) {
                const int e = k / words, q = k - e * words;
                ((uint4 *)&C.reports_h[e])[q] = ((const uint4 *)&C.reports_d[e])[q];
            }
            if (threadIdx.x == 0) *C.ticket = 0;
            __threadfence_system();        // (the one system-scope release of the whole batch)
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(C.flag_h, C.flag_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
    if (!open && it.Mi.vh > 0) {         // mini_norm_body with the sizes behind the compaction (k_pipe_clear left them)
        const int VHW = it.Mi.vh * it.Mi.vw, k = blockIdx.x * blockDim.x + threadIdx.x;
        if (k < NG * VHW) {
            const int tot = it.newn[k / VHW];
            int cnt = 0;
            for (int c = 0; c < MINI_COPIES; c++) { cnt += it.counts[c * NG * VHW + k]; it.counts[c * NG * VHW + k] = 0; }
            it.Mi.out[k] = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(cnt, 1 << 24), (float)(unsigned)tot);
        }
    }
}

// the items from pinned host memory to the device in ONE launch (hipMemcpyAsync of the 150 KB of a 32-environment batch ran as four to five
// blit launches of ~5 us each, ahead of everything else of the cycle: profiles/r06_summary.md)
__global__ void __launch_bounds__(256) k_pipe_upload(const uint4 *__restrict__ src_host, uint4 *__restrict__ dst, int n16) {
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n16; k += gridDim.x * blockDim.x) dst[k] = src_host[k];
}
void launch_pipe_upload(hipStream_t s, const PipeItem *h_items, PipeItem *d_items, int n_env) {
    static_assert(sizeof(PipeItem) % 16 == 0, "PipeItem is moved in 16-byte pieces");
    const int n16 = (int)(sizeof(PipeItem) / 16) * n_env;
    hipLaunchKernelGGL(k_pipe_upload, dim3(std::min(256, (n16 + 255) / 256)), dim3(256), 0, s, (const uint4 *)h_items, (uint4 *)d_items, n16);
}
static size_t pipe_eval_lds(int kmax) { return (size_t)kmax * 256 * 8; }
size_t render_sweep_lds(int VHW, int C) { return (size_t)RENDER_WAVES * 2 * 64 * C * sizeof(float) + (size_t)VHW * sizeof(RenderFastPos); }
void launch_pipe_cycle(hipStream_t s, const PipeItem *d_items, const PipeDims &D, const PipeCtl &C) {
    const dim3 by_agent((D.max_n + 255) / 256, D.G, D.n_env), by_tile((D.max_n + SCAN_TILE - 1) / SCAN_TILE, D.G, D.n_env);
    if (D.slots > 0 && D.sweep > 0)
        hipLaunchKernelGGL(k_pipe_render_sweep, dim3((D.sweep + D.render_blocks) * D.n_env * D.slots), dim3(64 * RENDER_WAVES), D.render_lds, s, d_items, D.slots, D.sweep, D.n_env * D.slots, std::max(1, D.render_blocks));
    else if (D.slots > 0 && D.render_blocks > 0)
        hipLaunchKernelGGL(k_pipe_render, dim3(D.render_blocks, D.n_env * D.slots), dim3(64 * RENDER_WAVES), D.render_lds, s, d_items, D.slots);
    hipLaunchKernelGGL(k_pipe_set_action, by_tile, dim3(SCAN_THREADS), 0, s, d_items);
    hipLaunchKernelGGL(k_pipe_draw, dim3((D.max_total + 255) / 256, 1, D.n_env), dim3(256), 0, s, d_items);
    hipLaunchKernelGGL(k_pipe_rank, by_agent, dim3(256), 0, s, d_items);
    if (pipe_eval_lds(D.kmax) > (48u << 10))
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_pipe_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)pipe_eval_lds(D.kmax));
    // (the last round raises CTR_OPEN_ATTACK of the environments in which something still changed: those continue on the host)
    for (int r = 1; r <= D.rounds; r++)
        hipLaunchKernelGGL(k_pipe_eval, by_agent, dim3(256), pipe_eval_lds(D.kmax), s, d_items, r, r == D.rounds ? CTR_OPEN_ATTACK : -1);
    if (D.rounds == 0) hipLaunchKernelGGL(k_pipe_force_open, dim3(D.n_env), dim3(64), 0, s, d_items);
    hipLaunchKernelGGL(k_pipe_strike, by_agent, dim3(256), 0, s, d_items);
    hipLaunchKernelGGL(k_pipe_commit, by_agent, dim3(256), 0, s, d_items);
    hipLaunchKernelGGL(k_pipe_clear, by_tile, dim3(SCAN_THREADS), sizeof(int) * (size_t)D.hist_cells, s, d_items);
    hipLaunchKernelGGL(k_pipe_finish, dim3(std::max(1, (D.hist_cells * D.G + 255) / 256), 1, D.n_env), dim3(256), 0, s, d_items, C);
}

}  // namespace magent_amd
