// cycle.hip -- small worlds: the whole step (k_step_solo) and a whole environment cycle of many environments (k_render_batch + k_step_solo_batch) in one launch each
// (device bodies shared with the other kernel translation units: kernels_dev.h)
#include "kernels_dev.h"

namespace magent_amd {
// ================================================================================================ one-launch step
// Small worlds are bound by the CHAIN of dependent launches, not by bandwidth: a 4000-agent step was 29 launches of
// 2-5 us each.  k_step_solo runs the whole of GridWorld::step (GridWorld.cc:456-631) as ONE workgroup of 1024 threads on
// one CU: the same phase bodies as the multi-launch driver, separated by workgroup barriers instead of kernel
// boundaries, with the fixed-point loops of the attack and move phases iterated to convergence inside the kernel (no
// optimistic rounds, no continuation on the host).  Per-cell scratch is never swept: the hit words / `wanted` counters
// (`S.hit`) and the claim words are zero / CLAIM_NONE between phases because whoever set a word resets it (O(agents),
// not O(cells)); with `live_paint` the painted map follows the step (vacated cells at once, live bodies at the end), so
// the next observation needs no k_paint.  The result goes straight to a pinned host record the host spins on.
constexpr int SOLO_STEP_THREADS = 1024;

#define SOLO_EACH(g_, i_) for (int g_ = 0; g_ < NG; g_++) for (int i_ = tid, n_##i_ = W.grp[g_].n; i_ < n_##i_; i_ += SOLO_STEP_THREADS)
// (uniform trip count: bodies that end in a wave ballot take every lane, i >= n included)
#define SOLO_EACH_UNIFORM(g_, i_) \
    for (int g_ = 0; g_ < NG; g_++) for (int i0_ = 0, n_##i_ = W.grp[g_].n, i_ = tid; i0_ < n_##i_; i0_ += SOLO_STEP_THREADS, i_ += SOLO_STEP_THREADS)

__device__ __forceinline__ void solo_step_main(WorldView &s_W, const SoloStep &S) {
    extern __shared__ unsigned s_dyn[];               // hit lists of the attack evaluation: [kmax][nt_eval] ranks, then refs
    // "something changed in round r" lives in s_flags[r % 3]: while round r runs, thread 0 re-arms the flag of round r + 1,
    // which was last read after the closing barrier of round r - 2 -- and everybody has passed the barrier of round r - 1 since
    __shared__ int s_flags[3];
    __shared__ unsigned long long s_marks[40];
    int n_marks = 0;
#define SOLO_MARK() do { if (tid == 0 && n_marks < 40) s_marks[n_marks] = wall_clock64(); n_marks++; } while (0)
    const int tid = threadIdx.x;
    const WorldView &W = s_W;
    const int NG = W.G;
    const GroupDev *gtab = W.grp;
    const TypeDev *ttab = W.type;
    SOLO_MARK();
    // ---- (cycle) set_action of the groups that act, in handle order (GridWorld::set_action: the order of the calls is the
    // order of the lists)
    for (int g = 0; g < NG; g++)
        if (S.actions[g] && W.grp[g].n > 0) {
            set_action_solo_body(W.grp[g], W.type[g], W.counters, W.large_map, W.bandwidth, S.actions[g], S.call_base[g]);
            __syncthreads();
        }
    const int A = W.counters[CTR_ATTACK];
    const unsigned x0 = (unsigned)W.counters[CTR_RNG];
    int rounds_attack = 0, rounds_move = 0, error = 0;
    __syncthreads();
    SOLO_MARK();

    if (A > 0) {
        // ---- shuffle (exact replay of the reference's Fisher-Yates, see k_shuffle_*)
        for (int i = tid; i < A; i += SOLO_STEP_THREADS) shuffle_draw_body(x0, i, S.sj, S.shead, S.sfirst, S.slink, S.powtab);
        __syncthreads();
        if (tid == 0) W.counters[CTR_RNG] = (int)rng_skip(x0, (unsigned)A);
        SOLO_MARK();   // 1: draw
        for (int i = tid; i < A; i += SOLO_STEP_THREADS) shuffle_chase_body(i, A, S.sj, S.shead, S.sfirst, S.slink, S.rank);
        __syncthreads();
        SOLO_MARK();   // 2: chase
        // ---- ranks, hit bits; the shuffle's list heads go back to zero
        for (int k = tid; k < A; k += SOLO_STEP_THREADS) { S.shead[k] = 0; S.sfirst[k] = 0; }
        // (one-cell bodies: the targets are listed as they are hit -- in the shuffle's link array, free by now)
        __shared__ int s_ntgt;
        const bool listed = !W.any_multicell;
        if (tid == 0) s_ntgt = 0;
        __syncthreads();
        SOLO_EACH(g, i) attack_rank_body(W, gtab, g, i, S.rank, S.hit, listed ? S.slink : nullptr, &s_ntgt);
        __syncthreads();
        SOLO_MARK();   // 5: rank
        // ---- death ranks: in-place fixed point, one round per barrier pair
        unsigned *s_rank = s_dyn;
        int *s_ref = (int *)(s_dyn + S.kmax * S.nt_eval);
        if (tid < 3) s_flags[tid] = 0;
        __syncthreads();
        while (true) {
            rounds_attack++;
            int *flag = &s_flags[rounds_attack % 3];
            if (tid == 0) s_flags[(rounds_attack + 1) % 3] = 0;
            if (listed) {
                if (tid < S.nt_eval)
                    for (int j = tid, n = s_ntgt; j < n; j += S.nt_eval) {
                        const int o = S.slink[j];
                        attack_eval_body(W, gtab, ttab, ref_group(o), ref_index(o), rounds_attack, S.hit, s_rank, s_ref, S.nt_eval, tid, flag, S.kmax);
                    }
            } else if (tid < S.nt_eval)
                for (int g = 0; g < NG; g++)
                    for (int i = tid, n = W.grp[g].n; i < n; i += S.nt_eval)
                        attack_eval_body(W, gtab, ttab, g, i, rounds_attack, S.hit, s_rank, s_ref, S.nt_eval, tid, flag, S.kmax);
            __syncthreads();
            const int changed = *flag;
            if (!changed) break;
            if (rounds_attack > S.max_rounds) { error = 1; break; }
        }
        SOLO_MARK();   // 6: eval rounds
        SOLO_EACH(g, i) attack_apply_body(W, gtab, ttab, g, i, S.hit);
        __syncthreads();
        SOLO_MARK();   // 7: apply
        // ---- the hit words back to zero: every attacker resets the word it may have set (one-cell bodies: in the next phase's loop;
        // generic bodies use the same array for their `wanted` counters there, so they need it clean first)
        if (W.any_multicell) SOLO_EACH(g, i) {
            const int pend = W.grp[g].pend[i];
            if ((pend & ~PEND_ARG) == PEND_ATTACK) {
                const int2 tc = attack_target(W, W.grp[g], W.type[g], i, pend & PEND_ARG);
                const int tx = tc.x, ty = tc.y;
                if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) S.hit[ty * W.w + tx] = 0u;
            }
        }
        if (W.any_multicell) __syncthreads();
    }
    if (tid == 0) W.counters[CTR_LAST_A] = A;
    SOLO_MARK();       // 8: hit words reset

    // ---- starve / recover, then the moves
    if (!W.any_multicell) {
        SOLO_EACH_UNIFORM(g, i) {
            if (A > 0 && i < W.grp[g].n) {      // (the attack phase's hit word of this agent's own attack)
                const int pend = W.grp[g].pend[i];
                if ((pend & ~PEND_ARG) == PEND_ATTACK) {
                    const int2 tc = attack_target(W, W.grp[g], W.type[g], i, pend & PEND_ARG);
                    if (tc.x >= 0 && tc.x < W.w && tc.y >= 0 && tc.y < W.h) S.hit[tc.y * W.w + tc.x] = 0u;
                }
            }
            move_prep_body(W, g, i, 0);
        }
        __syncthreads();
        SOLO_MARK();   // 9: starve + move candidates
        SOLO_EACH(g, i) move_claim_body(W, gtab, g, i);
        __syncthreads();
        SOLO_MARK();   // 10: claim
        SOLO_EACH(g, i) move_init_body(W, g, i);
        __syncthreads();
        SOLO_MARK();   // 11: init
        rounds_move = 1;                    // (chains are walked inside the commit: move_resolve)
        SOLO_MARK();   // 12: jump rounds
        SOLO_EACH(g, i) move_commit_body(W, gtab, g, i);
        __syncthreads();
        SOLO_MARK();   // 13: commit
    } else {
        if (W.turn_mode) {   // starvation, then the turns of this step (bodies re-lay their footprints), then the moves
            SOLO_EACH_UNIFORM(g, i) turn_prep_body(W, g, i, S.hit, 0);
            if (tid < 3) s_flags[tid] = 0;
            __syncthreads();
            int rounds_turn = 0;
            while (true) {
                rounds_turn++;
                int *flag = &s_flags[rounds_turn % 3];
                if (tid == 0) s_flags[(rounds_turn + 1) % 3] = 0;
                SOLO_EACH(g, i) turn_sweep_body(W, gtab, g, i, S.hit, flag);
                __syncthreads();
                const int open = *flag;
                if (!open) break;
                if (rounds_turn > S.max_rounds) { error = 3; break; }
            }
            SOLO_EACH(g, i) turn_vacate_body(W, g, i);
            __syncthreads();
            SOLO_EACH(g, i) turn_enter_body(W, g, i, S.hit);
            __syncthreads();
        }
        SOLO_EACH_UNIFORM(g, i) movg_prep_body(W, g, i, S.hit, 0, !W.turn_mode);
        if (tid < 3) s_flags[tid] = 0;
        __syncthreads();
        while (true) {
            rounds_move++;
            int *flag = &s_flags[rounds_move % 3];
            if (tid == 0) s_flags[(rounds_move + 1) % 3] = 0;
            SOLO_EACH(g, i) movg_sweep_body(W, gtab, g, i, S.hit, flag);
            __syncthreads();
            const int open = *flag;
            if (!open) break;
            if (rounds_move > S.max_rounds) { error = 2; break; }
        }
        SOLO_EACH(g, i) movg_collide_body(W, gtab, g, i, S.hit);
        __syncthreads();
        SOLO_EACH(g, i) movg_vacate_body(W, g, i);
        __syncthreads();
        SOLO_EACH(g, i) movg_enter_body(W, g, i);
        __syncthreads();
    }

    // ---- reward rules, in order (GridWorld::calc_reward)
    for (int r = 0; r < S.n_rules; r++) {
        const RuleArgs &R = S.rules[r];
        if (R.prog >= 0) {
            const RuleProg &P = S.progs[R.prog];
            for (int i0 = 0, n = W.grp[P.ga].n; i0 < n; i0 += SOLO_STEP_THREADS) rule_prog_body(W, gtab, P, i0 + tid);
            __syncthreads();
            if (P.n_obj) { for (int i = tid, n = W.grp[P.gb].n; i < n; i += SOLO_STEP_THREADS) rule_obj_body(W, R, i); __syncthreads(); }
        } else if (R.pair) {
            const int parts = R.ga == R.gy ? 1 : 2;
            if (W.grp[R.ga].n > 0 && W.grp[R.gy].n > 0 && W.grp[R.gb].n > 0) {
                for (int q = 0; q < parts; q++) { const int g = q ? R.gy : R.ga; for (int i = tid, n = W.grp[g].n; i < n; i += SOLO_STEP_THREADS) pair_link_body(W, R, g, i); }
                __syncthreads();
                for (int q = 0; q < parts; q++) { const int g = q ? R.gy : R.ga; for (int i0 = 0, n = W.grp[g].n; i0 < n; i0 += SOLO_STEP_THREADS) pair_pay_body(W, R, g, i0 + tid); }
                __syncthreads();
                for (int i = tid, n = W.grp[R.gb].n; i < n; i += SOLO_STEP_THREADS) pair_obj_body(W, R, i);
                __syncthreads();
            }
        } else {
            for (int i0 = 0, n = W.grp[R.ga].n; i0 < n; i0 += SOLO_STEP_THREADS) rule_body(W, R, i0 + tid);
            __syncthreads();
            if (R.n_obj) { for (int i = tid, n = W.grp[R.gb].n; i < n; i += SOLO_STEP_THREADS) rule_obj_body(W, R, i); __syncthreads(); }
        }
    }

    SOLO_MARK();       // rules (generic moves: + the move phase)
    // ---- end of step: pending actions are consumed; the claim words / wanted counters this step touched go back to
    // their rest state; every live agent paints its body (the vacated cells were emptied when they were left)
    SOLO_EACH(g, i) {
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        G.pend[i] = PEND_NONE;
        const int t = G.drank_a[i];            // the move candidate's target cell (both move paths), -1 otherwise
        if (t >= 0) {
            if (!W.any_multicell) W.claim[t] = CLAIM_NONE;
            else {
                const int ny = t / W.w, nx = t - ny * W.w;
                const int2 fp = body_dims(W, G, T, i);
                for (int by = 0; by < fp.y; by++)
                    for (int bx = 0; bx < fp.x; bx++) S.hit[(ny + by) * W.w + nx + bx] = 0u;
            }
        }
        if (W.live_paint) repaint_body(W, G, T, g, i);
    }
    __syncthreads();
    SOLO_MARK();       // finish + repaint

    // ---- (cycle) get_reward
    for (int g = 0; g < NG; g++)
        if (S.rewards[g]) {
            const GroupDev &G = W.grp[g];
            for (int i = tid; i < G.n; i += SOLO_STEP_THREADS) S.rewards[g][i] = G.next_reward[i] + S.group_reward[g];
        }
    // ---- (cycle) clear_dead: as k_clear_solo_all, the mode of a group decided here from its death counters
    int dead_ct = 0, taken_ct = 0;                    // thread g < NG: this step's report for group g
    if (tid < NG) {
        for (int k = 0; k < DEAD_SLOTS; k++) dead_ct += __hip_atomic_load(&W.counters[dead_slot(tid, k)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        taken_ct = __hip_atomic_load(&W.counters[CTR_TAKEN + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (S.do_clear) {
        __shared__ int s_gone[MAXG], s_alive[MAXG];
        if (tid < NG) s_gone[tid] = dead_ct + taken_ct;
        __syncthreads();                              // (also: the rewards above have read next_reward)
        for (int g = 0; g < NG; g++) {
            const GroupDev &G = W.grp[g];
            const float step_reward = W.type[g].step_reward;
            if (G.n == 0) { if (tid == 0) s_alive[g] = 0; continue; }
            if (s_gone[g] == 0) {
                for (int i = tid; i < G.n; i += SOLO_STEP_THREADS) { G.last_reward[i] = G.next_reward[i]; G.next_reward[i] = step_reward; G.last_op[i] = OP_NULL; G.op_obj[i] = -1; }
                if (tid == 0) s_alive[g] = G.n;
                continue;
            }
            const AltArrays D = S.dst[g];
            const int bw = W.type[g].bw, bl = W.type[g].bl;
            const int alive = solo_rank([&](int i) { return !G.dead[i]; },
                                        [&](int i, int r) {
                                            int x = G.x[i], y = G.y[i];
                                            D.x[r] = x; D.y[r] = y; D.id[r] = G.id[i]; D.hp[r] = G.hp[i]; D.last_action[r] = G.last_action[i];
                                            D.absorbed[r] = G.absorbed[i];
                   if (G.dir) D.dir[r] = G.dir[i];
                                            D.last_reward[r] = G.next_reward[i];
                                            D.next_reward[r] = step_reward;
                                            { const int2 fp = W.turn_mode ? dims_for_dir(W.type[g], G.dir[i]) : make_int2(bw, bl); body_fill(W, x, y, fp.x, fp.y, ref_pack(g, r)); }
                                        },
                                        G.n, 0);
            for (int r = tid; r < alive; r += SOLO_STEP_THREADS) { G.dead[r] = 0; G.last_op[r] = OP_NULL; G.op_obj[r] = -1; G.pend[r] = PEND_NONE; }
            if (tid < DEAD_SLOTS) W.counters[dead_slot(g, tid)] = 0;
            if (tid == 0) { W.counters[CTR_TAKEN + g] = 0; s_alive[g] = alive; }
        }
        __syncthreads();
        if (tid < NG && s_gone[tid] > 0 && s_W.grp[tid].n > 0) {   // the double-buffered arrays change places
            GroupDev &N = s_W.grp[tid];
            const AltArrays D = S.dst[tid];
            N.x = D.x; N.y = D.y; N.id = D.id; N.last_action = D.last_action; N.hp = D.hp; N.next_reward = D.next_reward;
            N.last_reward = D.last_reward; N.absorbed = D.absorbed; N.dir = D.dir;
            N.n = s_alive[tid];
        }
        __syncthreads();
        if (tid < MAXG) { S.gtab_out[tid] = s_W.grp[tid]; S.ttab_out[tid] = s_W.type[tid]; }
        // ---- (cycle) the minimap of the next observations: LDS histogram of every group, then count / total exactly as the
        // reference divides (k_minimap)
        if (S.mini.vh > 0) minimap_one_workgroup(W.grp, NG, S.mini, (int *)s_dyn, SOLO_STEP_THREADS);   // (the hit lists are done with)
    }
    SOLO_MARK();       // (cycle) rewards, clear_dead, minimap

    // ---- the step's report, straight into pinned host memory; per-step counters back to zero.
    // Every wave's device-memory writes (rewards, compacted arrays, tables, minimap) are released and the workgroup has met
    // before wave 0 publishes the sequence number: a host that has seen it may enqueue readers of those outputs
    __threadfence();
    __syncthreads();
    if (tid < 64) {
        const bool trig = tid < CTR_TRIGGER_END - CTR_TRIGGER && __hip_atomic_load(&W.counters[CTR_TRIGGER + tid], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0;
        const unsigned long long mask = __ballot(trig);
        if (tid < CTR_TRIGGER_END - CTR_TRIGGER) W.counters[CTR_TRIGGER + tid] = 0;
        if (tid < NG) { S.rec->dead[tid] = dead_ct; S.rec->taken[tid] = taken_ct; }
        if (tid == 0) {
            S.rec->triggers = mask; S.rec->rounds_mask = 0u;
            S.rec->rng = (unsigned)W.counters[CTR_RNG];
            S.rec->last_a = A;
            S.rec->unsupported = __hip_atomic_load(&W.counters[CTR_UNSUPPORTED], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.rec->pack_overflow = __hip_atomic_load(&W.counters[CTR_PACK_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.rec->bad_action = __hip_atomic_load(&W.counters[CTR_BAD_ACTION], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.rec->hit_overflow = __hip_atomic_load(&W.counters[CTR_HIT_OVERFLOW], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            S.rec->error = error; S.rec->open_attack = 0; S.rec->open_move = 0;
            S.rec->rounds_attack = rounds_attack; S.rec->rounds_move = rounds_move;
            S.rec->n_marks = n_marks < 40 ? n_marks : 40;
            for (int k = 0; k < S.rec->n_marks; k++) S.rec->marks[k] = s_marks[k];
            W.counters[CTR_ATTACK] = 0;
        }
        __threadfence_system();                       // (wave 0 only: lanes 1..NG-1 wrote their part above)
        if (tid == 0) __hip_atomic_store((int *)&S.rec->seq, S.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
#undef SOLO_EACH
#undef SOLO_EACH_UNIFORM
#undef SOLO_MARK

// The world description is indexed by a loop variable in the step (group g): as a by-value kernel argument that would make
// the compiler keep a private copy of it in scratch memory, per lane.  It is copied to LDS once instead and every phase
// body reads it there -- word by word from the kernarg segment (one environment per launch: W is the first argument, S
// follows it), or from a device array of items (many environments per launch, one workgroup each: env_cycle_many).
static_assert(sizeof(WorldView) % 8 == 0 && sizeof(SoloStep) % 4 == 0, "kernarg layout of k_step_solo");
__global__ void __launch_bounds__(SOLO_STEP_THREADS) k_step_solo(WorldView W_kernarg, SoloStep S_kernarg) {
    __shared__ WorldView s_W;
    __shared__ SoloStep s_S;
    typedef const __attribute__((address_space(4))) unsigned *kernarg_words;
    kernarg_words ka = (kernarg_words)__builtin_amdgcn_kernarg_segment_ptr();
    for (int k = threadIdx.x; k < (int)(sizeof(WorldView) / 4); k += SOLO_STEP_THREADS) ((unsigned *)&s_W)[k] = ka[k];
    for (int k = threadIdx.x; k < (int)(sizeof(SoloStep) / 4); k += SOLO_STEP_THREADS) ((unsigned *)&s_S)[k] = ka[sizeof(WorldView) / 4 + k];
    __syncthreads();
    solo_step_main(s_W, s_S);
}
__global__ void __launch_bounds__(SOLO_STEP_THREADS) k_step_solo_batch(const BatchItem *items) {
    __shared__ WorldView s_W;
    __shared__ SoloStep s_S;
    if (items[blockIdx.x].S.rec == nullptr) return;   // an environment of the round that does not take this path
    const unsigned *src = (const unsigned *)&items[blockIdx.x];
    static_assert(offsetof(BatchItem, W) == 0 && offsetof(BatchItem, S) == sizeof(WorldView), "BatchItem layout");
    for (int k = threadIdx.x; k < (int)(sizeof(WorldView) / 4); k += SOLO_STEP_THREADS) ((unsigned *)&s_W)[k] = src[k];
    for (int k = threadIdx.x; k < (int)(sizeof(SoloStep) / 4); k += SOLO_STEP_THREADS) ((unsigned *)&s_S)[k] = src[sizeof(WorldView) / 4 + k];
    __syncthreads();
    solo_step_main(s_W, s_S);
}
// the observations of many small environments in one launch: blockIdx.y = environment * slots + slot
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_batch(const BatchItem *items, int slots) {
    const int e = blockIdx.y / slots, k = blockIdx.y - e * slots;
    const BatchItem &it = items[e];
    if (k >= it.M.n || (int)blockIdx.x >= it.M.blocks[k]) return;
    const RenderArgs R = it.M.R[k];
    const RenderPlan P = it.M.P[k];
    RenderWorld V;
    V.w = it.W.w; V.h = it.W.h; V.G = it.W.G; V.viewcell = it.W.viewcell; V.mask = it.W.mask; V.grp = it.W.grp[R.g]; V.type = it.W.type[R.g];
    if (it.W.turn_mode) {
        if (it.W.vc_packed) render_block<true, true, 1, true, true>(V, R, P, blockIdx.x, it.M.blocks[k]);
        else render_block<true, true, 1, false, true>(V, R, P, blockIdx.x, it.M.blocks[k]);
    } else if (it.W.vc_packed) render_block<true, true, 1, true, false>(V, R, P, blockIdx.x, it.M.blocks[k]);
    else render_block<true, true, 1, false, false>(V, R, P, blockIdx.x, it.M.blocks[k]);
}

// clear_dead for every group of a small world in ONE launch of one workgroup (GridWorld::clear_dead, GridWorld.cc:633-665):
// stable compaction of the survivors into the alternate buffers + Agent::init_reward + re-indexing of the map (groups with
// deaths), Agent::init_reward alone (groups without); then the death counters and the device copy of the group table
// (the double-buffered arrays have changed places: ClearArgs::dst become the current ones).
__global__ void __launch_bounds__(SOLO_THREADS) k_clear_solo_all(WorldView W, ClearArgs A, GroupDev *gtab, TypeDev *ttab, MiniArgs M) {
    extern __shared__ int s_hist[];
    __shared__ int s_alive[MAXG];
    __shared__ GroupDev s_new[MAXG];
    for (int g = 0; g < W.G; g++) {
        const GroupDev &G = W.grp[g];
        const float step_reward = W.type[g].step_reward;
        if (A.mode[g] == 1) {
            for (int i = threadIdx.x; i < G.n; i += SOLO_THREADS) { G.last_reward[i] = G.next_reward[i]; G.next_reward[i] = step_reward; G.last_op[i] = OP_NULL; G.op_obj[i] = -1; }
            if (threadIdx.x == 0) s_alive[g] = G.n;
        } else if (A.mode[g] == 2) {
            const ClearArgs::Alt D = A.dst[g];
            const int bw = W.type[g].bw, bl = W.type[g].bl;
            const int alive = solo_rank([&](int i) { return !G.dead[i]; },
                                        [&](int i, int r) {
                                            int x = G.x[i], y = G.y[i];
                                            D.x[r] = x; D.y[r] = y; D.id[r] = G.id[i]; D.hp[r] = G.hp[i]; D.last_action[r] = G.last_action[i];
                                            D.absorbed[r] = G.absorbed[i];
                   if (G.dir) D.dir[r] = G.dir[i];
                                            D.last_reward[r] = G.next_reward[i];
                                            D.next_reward[r] = step_reward;
                                            { const int2 fp = W.turn_mode ? dims_for_dir(W.type[g], G.dir[i]) : make_int2(bw, bl); body_fill(W, x, y, fp.x, fp.y, ref_pack(g, r)); }
                                        },
                                        G.n, 0);
            // every read of the in-place arrays is done (solo_rank ends with a barrier): reset them for the survivors
            for (int r = threadIdx.x; r < alive; r += SOLO_THREADS) { G.dead[r] = 0; G.last_op[r] = OP_NULL; G.op_obj[r] = -1; G.pend[r] = PEND_NONE; }
            if (threadIdx.x < DEAD_SLOTS) W.counters[dead_slot(g, threadIdx.x)] = 0;
            if (threadIdx.x == 0) { W.counters[CTR_TAKEN + g] = 0; s_alive[g] = alive; }
        } else if (threadIdx.x == 0) s_alive[g] = G.n;
    }
    __syncthreads();
    if (threadIdx.x < MAXG) {
        const int g = threadIdx.x;
        GroupDev N = W.grp[g];
        if (g < W.G && A.mode[g] == 2) {
            const ClearArgs::Alt D = A.dst[g];
            N.x = D.x; N.y = D.y; N.id = D.id; N.last_action = D.last_action; N.hp = D.hp; N.next_reward = D.next_reward;
            N.last_reward = D.last_reward; N.absorbed = D.absorbed; N.dir = D.dir;
            N.n = s_alive[g];
        }
        gtab[g] = N; ttab[g] = W.type[g];
        s_new[g] = N;
    }
    if (M.vh > 0) {      // the minimap of the next observations (they will find it made)
        __syncthreads();
        minimap_one_workgroup(s_new, W.G, M, s_hist, SOLO_THREADS);
    }
}

void launch_step_solo(hipStream_t s, const WorldView &W, const SoloStep &S) {
    hipLaunchKernelGGL(k_step_solo, dim3(1), dim3(SOLO_STEP_THREADS), solo_step_lds(W, S), s, W, S);
}
void launch_clear_solo_all(hipStream_t s, const WorldView &W, const ClearArgs &A, GroupDev *gtab, TypeDev *ttab, const MiniArgs &M) {
    const size_t lds = M.vh > 0 ? sizeof(int) * ((size_t)W.G * M.vh * M.vw + W.G) : 0;
    hipLaunchKernelGGL(k_clear_solo_all, dim3(1), dim3(SOLO_THREADS), lds, s, W, A, gtab, ttab, M);
}
void launch_cycle_batch(hipStream_t s, const BatchItem *d_items, int n_env, int slots, int max_blocks, size_t render_lds, size_t step_lds) {
    if (slots > 0 && max_blocks > 0)
        hipLaunchKernelGGL(k_render_batch, dim3(max_blocks, n_env * slots), dim3(64 * RENDER_WAVES), render_lds, s, d_items, slots);
    hipLaunchKernelGGL(k_step_solo_batch, dim3(n_env), dim3(SOLO_STEP_THREADS), step_lds, s, d_items);
}
size_t solo_step_lds(const WorldView &W, const SoloStep &S) {
    size_t lds = (size_t)S.kmax * S.nt_eval * 8;
    if (S.mini.vh > 0) lds = std::max(lds, sizeof(int) * ((size_t)W.G * S.mini.vh * S.mini.vw + W.G));
    return lds;
}
int solo_step_static_lds() {   // static LDS of k_step_solo (tables, scan scratch): taken off the budget of the hit lists
    hipFuncAttributes a{};
    if (hipFuncGetAttributes(&a, reinterpret_cast<const void *>(k_step_solo)) != hipSuccess) return 8192;
    return (int)a.sharedSizeBytes;
}
bool solo_step_allow_lds(size_t bytes) {   // dynamic LDS above the default limit has to be asked for
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_solo), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void *>(k_step_solo_batch), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes) == hipSuccess;
}

}  // namespace magent_amd
