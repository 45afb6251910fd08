// step.hip -- the multi-launch step of the grid-world engine for gfx950: set_action, attack shuffle, the generic attack / move / turn phases, the pipeline of plain games, reward rules, clear_dead, small gathers, their launchers
// (device bodies shared with the other kernel translation units: kernels_dev.h)
#include "plain_dev.h"

namespace magent_amd {

// tests only (MAGENT_TUNE attack_pairs=0 / move_batches=0): leave a phase open without running a round
__global__ void k_set_counter(int *counters, int index, int value, int unless_index) {
    if (threadIdx.x == 0 && !(unless_index >= 0 && counters[unless_index])) counters[index] = value;
}
// per-step counters back to zero after the end-of-step readback (dead_ct lives until clear_dead)
__global__ void k_step_reset(int *counters) {
    if (threadIdx.x == 0) counters[CTR_ATTACK] = 0;
    for (int k = CTR_TRIGGER + threadIdx.x; k < CTR_TRIGGER_END; k += blockDim.x) counters[k] = 0;
    for (int k = threadIdx.x; k < ROUND_SLOTS; k += blockDim.x) counters[CTR_ROUND_CHANGED + k] = 0;
    for (int k = threadIdx.x; k < ATT_SLOTS; k += blockDim.x) counters[att_slot(k)] = 0;
}
__global__ void __launch_bounds__(64) k_step_report(int *counters, StepRecord *rec, int seq, int NG, int mode) { step_report_body(counters, rec, seq, NG, mode); }
static int report_mode() { return REPORT_HOST; }
__global__ void k_set_rng(int *counters, unsigned x) { if (threadIdx.x == 0) counters[CTR_RNG] = (int)x; }

// memset that respects the gate: the claim array still holds the attack phase's hit bits when the host has to continue
// the attack rounds
__global__ void __launch_bounds__(256) k_fill32_gated(WorldView W, unsigned *p, unsigned v, size_t n) {
    if (attack_open(W)) return;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

__global__ void __launch_bounds__(256) k_fill32(unsigned *p, unsigned v, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}

// ------------------------------------------------------------------------------------------------ device tables
// copies the by-value group/type tables into device memory for kernels that index them per lane
__global__ void k_set_tables(WorldView W, GroupDev *gtab, TypeDev *ttab) {
    int i = threadIdx.x;
    if (i < MAXG) { gtab[i] = W.grp[i]; ttab[i] = W.type[i]; }
}
// clear_dead compacts groups of up to this many agents in one workgroup (MAGENT_TUNE scan_solo_max; tests lower it further so that
// small worlds run the multi-block scans of the large ones).  1024 since round 5, measured (profiles/r05_summary.md): one workgroup
// walking 20000 agents takes 0.108 ms where the block scans take 0.016 (battle 600 x 600, 2 x 20000: 0.202 -> 0.121 ms per cycle);
// at 2 x 2000 the block scans are level or ahead (0.0100 against 0.0131)
static int scan_solo_max() {
    static const int v = std::max(0, std::min(SOLO_MAX, tune("scan_solo_max", 1024)));
    return v;
}
__global__ void __launch_bounds__(SCAN_THREADS) k_set_action_a(WorldView W, int g, const int *actions, int call_base, int *sums, int *wpre, int tile_off) {
    set_action_tile_body(W, g, actions, call_base, sums, wpre, tile_off);
}
// the sequence numbers written out (a step that was given its actions in tiles but runs as ONE launch after all: k_step_solo reads them
// from `key`; happens when the world shrank below the one-launch limit between set_action and step)
__global__ void __launch_bounds__(256) k_seq_assign(WorldView W, int g, const int *sums, const int *wpre, int tile_off, int write_total) {
    if (write_total && blockIdx.x == 0 && threadIdx.x < 64) {     // (the list's length where the one-launch step looks for it; the spread counters back to zero)
        int v = threadIdx.x < ATT_SLOTS ? W.counters[att_slot(threadIdx.x)] : 0;
        if (threadIdx.x < ATT_SLOTS) W.counters[att_slot(threadIdx.x)] = 0;
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
        if (threadIdx.x == 0) W.counters[CTR_ATTACK] = v;
    }
    const GroupDev G = W.grp[g];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool att = i < G.n && (G.pend[i] & ~PEND_ARG) == PEND_ATTACK;
    const int seq = attack_seq(sums, wpre, tile_off, i, att);
    if (att) G.key[i] = (unsigned)seq;
}
__global__ void __launch_bounds__(SOLO_THREADS) k_set_action_solo(WorldView W, int g, const int *actions, int call_base) {
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    set_action_solo_body(G, T, W.counters, W.large_map, W.bandwidth, actions, call_base);
}
// last_action of a group whose actions are set but not stepped yet (an observation between set_action and step)
__global__ void __launch_bounds__(256) k_commit_action(GroupDev G, TypeDev T) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    const int pend = G.pend[i];
    if (pend != PEND_NONE) G.last_action[i] = pend_action(pend, T);
}

__global__ void __launch_bounds__(256) k_shuffle_draw(int *counters, int *j, int *head, int *first, int *link, unsigned *hitbits, size_t ncell,
                                                     const unsigned *powtab, int tiled) {
    shuffle_draw_launch_body(counters, j, head, first, link, hitbits, ncell, powtab, tiled);
}
__global__ void __launch_bounds__(256) k_shuffle_chase(int *counters, const int *j, const int *head, const int *first, const int *link, int *rank) {
    const int A = counters[CTR_ATTACK];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) {   // (every draw has read the old state: k_shuffle_draw ran before)
        counters[CTR_LAST_A] = A;
        counters[CTR_RNG] = (int)rng_skip((unsigned)counters[CTR_RNG], (unsigned)A);   // the host mirror is refreshed by the end-of-step report
    }
    if (i >= A) return;
    shuffle_chase_body(i, A, j, head, first, link, rank);
}
__global__ void __launch_bounds__(256) k_attack_rank(WorldView W, const GroupDev *gtab, const int *rank, unsigned *hitbits, int *shuf_head, int *shuf_first,
                                                     const int *sums, const int *wpre, SeqPlan P) {
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // attack rounds start
    const int A = W.counters[CTR_ATTACK];
    // the shuffle's list heads and first-hit words have been read for the last time (k_shuffle_chase): back to zero for their next use
    for (int k = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < A; k += gridDim.x * gridDim.y * blockDim.x) {
        shuf_head[k] = 0; shuf_first[k] = 0;
    }
    if (A == 0) return;
    const int g = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = W.grp[g].n;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    int seq = -1;
    if (P.off[g] >= 0) seq = attack_seq(sums, wpre, P.off[g], i, i < n && (W.grp[g].pend[i] & ~PEND_ARG) == PEND_ATTACK);   // (every thread of the workgroup)
    if (i >= n) return;
    attack_rank_body(W, gtab, g, i, rank, hitbits, nullptr, nullptr, seq);
}
// workgroup size: as large as the hit lists (kmax x threads x 8 B of LDS) allow, see att_threads()
__global__ void __launch_bounds__(256) k_attack_eval(WorldView W, const GroupDev *gtab, const TypeDev *ttab, int round,
                                                     const unsigned *hitbits, int kmax, int flag) {
    if (W.counters[CTR_ATTACK] == 0) return;
    extern __shared__ unsigned s_hit[];               // [kmax][ATT_THREADS] ranks, then [kmax][ATT_THREADS] refs
    const int ATT_THREADS = blockDim.x;
    const int g = blockIdx.y, tid = threadIdx.x;
    const int i = blockIdx.x * blockDim.x + tid;
    if (i >= W.grp[g].n) return;
    attack_eval_body(W, gtab, ttab, g, i, round, hitbits, s_hit, (int *)(s_hit + kmax * ATT_THREADS), ATT_THREADS, tid,
                     flag >= 0 ? &W.counters[flag] : nullptr, kmax);
}

// food_mode: the food that lay on the map before this step.  One thread per cell: the hits on a food cell eat from it
// in rank order (Map.cc:292-303); evaluated in every round (an eater that turns out to be dead does not eat).
__global__ void __launch_bounds__(256) k_food_eval(WorldView W, const GroupDev *gtab, const TypeDev *ttab, int round,
                                                  const unsigned *hitbits, int kmax, int flag) {
    if (W.counters[CTR_ATTACK] == 0) return;
    extern __shared__ unsigned s_hit[];
    const int NT = blockDim.x, tid = threadIdx.x;
    unsigned *s_rank = s_hit;
    int *s_ref = (int *)(s_hit + kmax * NT);
    const int c = blockIdx.x * blockDim.x + tid;
    if (c >= W.w * W.h || W.occ[c] != OCC_FOOD) return;
    const unsigned bits = hitbits[c];
    if (!bits) return;
    const int cy = c / W.w, cx = c - cy * W.w;
    const int nh = gather_hits(W, bits, cx, cy, s_rank, s_ref, NT, tid, 0, kmax);
    sort_hits(s_rank, s_ref, NT, tid, nh);
    float food = W.food[c];
    bool present = true;
    for (int k = 0; k < nh; k++) {
        const int a = s_ref[k * NT + tid];
        float e = -1.0f;
        if (present && (unsigned)gtab[ref_group(a)].drank_a[ref_index(a)] >= s_rank[k * NT + tid]) {
            e = fminf(ttab[ref_group(a)].eat_ability, food);
            food -= e;
            if ((double)food < 0.1) present = false;
        }
        set_eat(W, gtab, a, e, round, flag >= 0 ? &W.counters[flag] : nullptr);
    }
    W.food_next[c] = present ? food : -1.0f;
}

__global__ void __launch_bounds__(256) k_food_apply(WorldView W, const unsigned *hitbits) {
    if (W.counters[CTR_ATTACK] == 0 || attack_open(W)) return;
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= W.w * W.h || W.occ[c] != OCC_FOOD || !hitbits[c]) return;
    const float left = W.food_next[c];
    if (left < 0.0f) { W.occ[c] = OCC_EMPTY; if (W.live_paint) vc_store(W, c, OCC_EMPTY, 0u); } else W.food[c] = left;
}
__global__ void __launch_bounds__(256) k_attack_apply(WorldView W, const GroupDev *gtab, const TypeDev *ttab, const unsigned *hitbits) {
    if (W.counters[CTR_ATTACK] == 0 || attack_open(W)) return;
    const int g = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W.grp[g].n) return;
    attack_apply_body(W, gtab, ttab, g, i, hitbits);
}

// render support: ev[rank] = {attacker id, target x, target y, 1} for every attack that was executed (attacker alive at
// its turn), in the order the reference appends them (GridWorld.cc:483-485: before the blank-target test, so blank and
// out-of-board targets are recorded too); {.,.,.,0} for list entries whose attacker was already dead
__global__ void __launch_bounds__(256) k_attack_events(WorldView W, int4 *ev) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    const int pend = G.pend[i];
    if ((pend & ~PEND_ARG) != PEND_ATTACK) return;
    const unsigned my_rank = G.key[i];
    const int dr = G.drank_a[i];
    const int2 tc = attack_target(W, G, W.type[g], i, pend & PEND_ARG);
    const bool executed = dr != -1 && (unsigned)dr >= my_rank;
    ev[my_rank] = make_int4(G.id[i], tc.x, tc.y, executed ? 1 : 0);
}
__global__ void __launch_bounds__(256) k_move_prep(WorldView W, unsigned *claim_words, size_t n_words) {
    if (attack_open(W)) return;
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // move rounds start
    // the claim words back to "nobody" (they held the attack phase's hit bits until now)
    for (size_t k = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < n_words;
         k += (size_t)gridDim.x * gridDim.y * blockDim.x) claim_words[k] = 0xFFFFFFFFu;
    move_prep_body(W, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.x % DEAD_SLOTS);
}
__global__ void __launch_bounds__(256) k_move_claim(WorldView W, const GroupDev *gtab) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) move_claim_body(W, gtab, g, i);
}
__global__ void __launch_bounds__(256) k_move_init(WorldView W) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) move_init_body(W, g, i);
}
// (multi-launch driver: the last launch of the move phase also keeps the painted map current -- every live agent paints its
// cell; the cells that were left were emptied where they were left.  No cell has two writers: a cell somebody enters is not
// emptied by the one who left it, see above)
__global__ void __launch_bounds__(256) k_move_commit(WorldView W, const GroupDev *gtab) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= W.grp[g].n) return;
    move_commit_body(W, gtab, g, i);
    if (W.live_paint) repaint_body(W, W.grp[g], W.type[g], g, i);
}

__global__ void __launch_bounds__(256) k_plain_rank(WorldView W, PlainWorld PW, const PlainGroup *ptab, ShuffleBufs B, const int *sums, const int *wpre, SeqPlan P) {
    plain_rank_body(W, PW, ptab, B, sums, wpre, P);
}
__global__ void __launch_bounds__(256) k_plain_eval(WorldView W, PlainWorld PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag,
                                                   int *shuf_head, int *shuf_first) {
    plain_eval_body(W, PW, ptab, gtab, ttab, round, flag, shuf_head, shuf_first);
}
__global__ void __launch_bounds__(256) k_strike(WorldView W, PlainWorld PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, StrikeRules R) {
    strike_body(W, PW, ptab, gtab, ttab, R);
}
__global__ void __launch_bounds__(256) k_plain_commit(WorldView W, PlainWorld PW, StepRecord *rec, int seq, int report_mode) {
    plain_commit_body(W, PW, rec, seq, report_mode);
}
__global__ void __launch_bounds__(256) k_movg_prep(WorldView W, unsigned *wanted, int starve) {
    if (attack_open(W)) return;
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // move rounds start
    movg_prep_body(W, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, wanted, blockIdx.x % DEAD_SLOTS, starve != 0);
}
__global__ void __launch_bounds__(256) k_movg_sweep(WorldView W, const GroupDev *gtab, const unsigned *wanted, int flag) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_sweep_body(W, gtab, g, i, wanted, flag >= 0 ? &W.counters[flag] : nullptr);
}
__global__ void __launch_bounds__(256) k_movg_collide(WorldView W, const GroupDev *gtab, const unsigned *wanted) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_collide_body(W, gtab, g, i, wanted);
}
__global__ void __launch_bounds__(256) k_movg_vacate(WorldView W) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_vacate_body(W, g, i);
}
__global__ void __launch_bounds__(256) k_movg_enter(WorldView W) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_enter_body(W, g, i);
}
__global__ void __launch_bounds__(256) k_turn_prep(WorldView W, unsigned *wanted) {
    if (attack_open(W)) return;
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;
    turn_prep_body(W, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, wanted, blockIdx.x % DEAD_SLOTS);
}
__global__ void __launch_bounds__(256) k_turn_sweep(WorldView W, const GroupDev *gtab, const unsigned *wanted, int flag) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) turn_sweep_body(W, gtab, g, i, wanted, flag >= 0 ? &W.counters[flag] : nullptr);
}
__global__ void __launch_bounds__(256) k_turn_vacate(WorldView W) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) turn_vacate_body(W, g, i);
}
__global__ void __launch_bounds__(256) k_turn_enter(WorldView W) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) turn_enter_body(W, g, i, nullptr);
}
__global__ void __launch_bounds__(256) k_rule(WorldView W, RuleBatch B) {
    if (step_open(W)) return;
    rule_body(W, B.r[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_rule_obj(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    rule_obj_body(W, A, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_rule_prog(WorldView W, const GroupDev *gtab, RuleProg P) {
    if (step_open(W)) return;
    rule_prog_body(W, gtab, P, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_pair_link(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    pair_link_body(W, A, blockIdx.y ? A.gy : A.ga, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_pair_pay(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    pair_pay_body(W, A, blockIdx.y ? A.gy : A.ga, blockIdx.x * blockDim.x + threadIdx.x);
}
__global__ void __launch_bounds__(256) k_pair_obj(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    pair_obj_body(W, A, blockIdx.x * blockDim.x + threadIdx.x);
}

// end of step: pending actions are consumed
__global__ void __launch_bounds__(256) k_finish(WorldView W) {
    if (step_open(W)) return;
    const GroupDev G = W.grp[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    G.pend[i] = PEND_NONE;
    if (W.live_paint) repaint_body(W, G, W.type[blockIdx.y], blockIdx.y, i);   // (generic bodies: after every enter / absorb of the step)
}

// ------------------------------------------------------------------------------------------------ small gathers
__global__ void __launch_bounds__(256) k_get_reward(GroupDev G, float group_reward, float *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) out[i] = G.next_reward[i] + group_reward;
}
__global__ void __launch_bounds__(256) k_get_pos(GroupDev G, int *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) ((int2 *)out)[i] = make_int2(G.x[i], G.y[i]);
}
__global__ void __launch_bounds__(256) k_get_alive(GroupDev G, unsigned char *out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) out[i] = G.dead[i] ? 0 : 1;
}

// ------------------------------------------------------------------------------------------------ clear_dead
// Agent::init_reward for a group without deaths (no compaction needed)
__global__ void __launch_bounds__(256) k_init_reward(WorldView W, int g) {
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    G.last_reward[i] = G.next_reward[i];
    G.next_reward[i] = W.type[g].step_reward;
    G.last_op[i] = OP_NULL;
    G.op_obj[i] = -1;
}

// all groups at once (blockIdx.y = group): block totals of the survivors ...
__global__ void __launch_bounds__(SCAN_THREADS) k_clear_count(WorldView W, ClearArgs A, int *sums) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    if (A.mode[g] != 2 || (int)(blockIdx.x * SCAN_TILE) >= G.n) return;
    int tot = block_count([&](int i) { return !G.dead[i]; }, G.n);
    if (threadIdx.x == 0) sums[A.sums_off[g] + blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_clear_compact(WorldView W, ClearArgs A, const int *sums, MiniArgs M, int *counts) {
    clear_compact_body(W, A, A.mode[blockIdx.y], sums, M, counts);
}
__global__ void __launch_bounds__(256) k_mini_norm(WorldView Wn, MiniArgs M, int *counts) {
    mini_norm_body(Wn, M, counts, blockIdx.x * blockDim.x + threadIdx.x);
}

// ... and, with the pointers swapped (Wn = the view after clear_dead): the death counters, the device copies of the group / type
// tables and the division of the next minimap -- a handful of workgroups (the per-agent resets ride in k_clear_compact since round 4)
__global__ void __launch_bounds__(256) k_clear_finish(WorldView Wn, ClearArgs A, GroupDev *gtab, TypeDev *ttab, MiniArgs M, int *counts) {
    if (blockIdx.x == 0) {
        if (threadIdx.x < MAXG) { gtab[threadIdx.x] = Wn.grp[threadIdx.x]; ttab[threadIdx.x] = Wn.type[threadIdx.x]; }
        for (int g = 0; g < Wn.G; g++) {
            if (A.mode[g] != 2) continue;
            if (threadIdx.x < DEAD_SLOTS) Wn.counters[dead_slot(g, threadIdx.x)] = 0;
            if (threadIdx.x == 0) Wn.counters[CTR_TAKEN + g] = 0;
        }
    }
    if (M.vh > 0) mini_norm_body(Wn, M, counts, blockIdx.x * blockDim.x + threadIdx.x);
}

__global__ void __launch_bounds__(SOLO_THREADS) k_compact_solo(WorldView W, int g, GroupDev D) {
    const GroupDev G = W.grp[g];
    const float step_reward = W.type[g].step_reward;
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
    for (int r = threadIdx.x; r < alive; r += SOLO_THREADS) { D.dead[r] = 0; D.last_op[r] = OP_NULL; D.op_obj[r] = -1; D.pend[r] = PEND_NONE; }
    if (threadIdx.x < DEAD_SLOTS) W.counters[dead_slot(g, threadIdx.x)] = 0;
    if (threadIdx.x == 0) W.counters[CTR_TAKEN + g] = 0;
}

void launch_set_tables(hipStream_t s, const WorldView &W, GroupDev *gtab, TypeDev *ttab) {
    hipLaunchKernelGGL(k_set_tables, dim3(1), dim3(64), 0, s, W, gtab, ttab);
}

void launch_commit_action(hipStream_t s, const GroupDev &G, const TypeDev &T) {
    if (G.n > 0) hipLaunchKernelGGL(k_commit_action, dim3((G.n + 255) / 256), dim3(256), 0, s, G, T);
}

// tile_off < 0: the one-workgroup form (worlds that step in one launch: it assigns the sequence numbers itself); else the tiled form,
// whose counts go to sums[tile_off ...] / wpre (launch.h: SeqPlan)
void launch_set_action(hipStream_t s, const WorldView &W, int g, const int *actions, int call_base, int *sums, int *wpre, int tile_off) {
    int n = W.grp[g].n;
    if (n <= 0) return;
    if (tile_off < 0) { hipLaunchKernelGGL(k_set_action_solo, dim3(1), dim3(SOLO_THREADS), 0, s, W, g, actions, call_base); return; }
    int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_set_action_a, dim3(nb), dim3(SCAN_THREADS), 0, s, W, g, actions, call_base, sums, wpre, tile_off);
}
void launch_seq_assign(hipStream_t s, const WorldView &W, int g, const int *sums, const int *wpre, int tile_off, bool write_total) {
    int n = W.grp[g].n;
    if (n > 0) hipLaunchKernelGGL(k_seq_assign, dim3((n + 255) / 256), dim3(256), 0, s, W, g, sums, wpre, tile_off, write_total ? 1 : 0);
}

// n_max = upper bound of the attack-list length (the number of agents); the actual length is read on the device
void launch_shuffle(hipStream_t s, int n_max, int *counters, const ShuffleBufs &B, int *rank, unsigned *hitbits, size_t ncell, const unsigned *powtab, bool tiled) {
    // head / first are zero here: zeroed when allocated, and again by k_attack_rank after every use
    dim3 g((n_max + 255) / 256), b(256);
    hipLaunchKernelGGL(k_shuffle_draw, g, b, 0, s, counters, B.j, B.head, B.first, B.link, hitbits, ncell, powtab, tiled ? 1 : 0);
    hipLaunchKernelGGL(k_shuffle_chase, g, b, 0, s, counters, B.j, B.head, B.first, B.link, rank);
}
// ================================================================================================ the literal loop: repeated set_action, goals that move
// GridWorld::set_action APPENDS to the step's action lists (GridWorld.cc:403-454): a group that is given actions twice before a step
// has every agent act twice -- two entries in the shuffled attack list, two turns and two moves in list order, the second from wherever
// the first one ended.  The parallel phases above rest on "one pending action per agent" (and the generic move resolution on "goals stand
// still"); no caller of the reference does either, so these steps are served by the reference's own sequential loops on ONE lane of the
// device: exact by construction, every game (bodies of any size, turn_mode, food_mode, goals, kill_supply), and slow -- about a
// microsecond per list entry.
//   attack loop GridWorld.cc:464-507 (Map::get_attack_obj Map.cc:209-252, Map::do_attack Map.cc:255-310, Agent::be_attack
//   GridWorld.h:203-209), starve GridWorld.cc:519-542, turns GridWorld.cc:544-572 (Map::do_turn Map.cc:361-406), moves
//   GridWorld.cc:574-613 (Map::do_move Map.cc:313-358, is_blank_area :454-470, get_collide :486-501)
// events: {attacker id, target x, target y, 1} per executed attack in list order (GridWorld.cc:484), {., ., ., 0} for the dead's entries
__device__ __forceinline__ bool serial_blank_area(const WorldView &W, int x, int y, int bw, int bl, int self) {
    if (x < 0 || y < 0 || x + bw >= W.w || y + bl >= W.h) return false;
    for (int a = 0; a < bw; a++)
        for (int b = 0; b < bl; b++) {
            const int o = W.occ[(y + b) * W.w + x + a];
            if (o != OCC_EMPTY && o != self) return false;        // walls, food and other bodies occupy
        }
    return true;
}
__global__ void __launch_bounds__(64) k_step_serial(WorldView W, const SerialCall *calls, int n_calls, int2 *alist, int4 *mlist, int4 *msorted, int n_sep,
                                                   int4 *events) {
    if (threadIdx.x != 0) return;
    const int bandwidth = W.bandwidth;
    int A = 0, M = 0;
    // ---- the lists, in call order (Agent::set_action stores last_action at once: the last call wins).  Turns and moves share one
    // array: list number 0 .. n_sep (stripes, then the boundary list) for a turn, n_sep + 1 + the same for a move -- every turn runs
    // before the first move
    for (int c = 0; c < n_calls; c++) {
        const int g = calls[c].g;
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        const int *act = calls[c].actions;
        for (int i = 0; i < G.n; i++) {
            const int a = act[i];
            if (a < 0 || a >= T.n_move + T.n_turn + T.n_attack) { W.counters[CTR_BAD_ACTION] = 1; continue; }
            G.last_action[i] = a;
            if (a < T.n_move + T.n_turn) {
                int list = n_sep;                                            // the boundary list runs last
                if (W.large_map) { const int x_ = G.x[i] % bandwidth; if (!(x_ < 4 || x_ > bandwidth - 4)) list = G.x[i] / bandwidth; }
                mlist[M++] = make_int4(ref_pack(g, i), a, a < T.n_move ? n_sep + 1 + list : list, 0);
            } else alist[A++] = make_int2(ref_pack(g, i), a - T.n_move - T.n_turn);
        }
    }
    // ---- shuffle (GridWorld.cc:464-468): minstd_rand0, (int)rng() % (i + 1)
    unsigned long long x = (unsigned)W.counters[CTR_RNG];
    for (int i = 0; i < A; i++) {
        x = x * 16807ull % 2147483647ull;
        const int j = (int)x % (i + 1);
        const int2 t = alist[i]; alist[i] = alist[j]; alist[j] = t;
    }
    W.counters[CTR_RNG] = (int)x;
    W.counters[CTR_LAST_A] = A;
    // ---- attacks, in that order
    for (int e = 0; e < A; e++) {
        const int g = ref_group(alist[e].x), i = ref_index(alist[e].x), k = alist[e].y;
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        if (G.dead[i]) { events[e] = make_int4(0, 0, 0, 0); continue; }
        const int2 tc = attack_target(W, G, T, i, k);
        events[e] = make_int4(G.id[i], tc.x, tc.y, 1);
        int o = OCC_EMPTY, cell = -1;
        if (tc.x >= 0 && tc.x < W.w && tc.y >= 0 && tc.y < W.h) { cell = tc.y * W.w + tc.x; o = W.occ[cell]; }
        if (o == OCC_FOOD) {      // Map.cc:292-303: eat; the attack counts as one on an object (reward 0.0 + attack_penalty)
            const float add = fminf(T.eat_ability, W.food[cell]);
            G.hp[i] = fminf(T.hp, G.hp[i] + add);
            W.food[cell] -= add;
            if (W.food[cell] < 0.1f) W.occ[cell] = OCC_EMPTY;
            G.next_reward[i] += 0.0f + T.attack_penalty;
            continue;
        }
        if (o < 0 || (!T.attack_in_group && ref_group(o) == g)) { G.next_reward[i] += T.attack_penalty; continue; }
        const int tg = ref_group(o), ti = ref_index(o);
        const GroupDev &V = W.grp[tg];
        const TypeDev &TV = W.type[tg];
        float reward = 0.0f;
        V.hp[ti] -= T.damage;
        if (V.hp[ti] < 0.0f) { V.dead[ti] = 1; V.next_reward[ti] = TV.dead_penalty; }
        if (V.dead[ti]) {
            G.last_op[i] = OP_KILL; G.op_obj[i] = o;
            const int2 fp = body_dims(W, V, TV, ti);
            body_fill(W, V.x[ti], V.y[ti], fp.x, fp.y, OCC_EMPTY);            // Map::remove_agent
            W.counters[dead_slot(tg, 0)] += 1;
            G.hp[i] = fminf(T.hp, G.hp[i] + TV.kill_supply);
            if (W.food_mode) { W.occ[cell] = OCC_FOOD; W.food[cell] = TV.food_supply; }   // on the attacked cell only (Map.cc:277-284)
            reward = TV.kill_reward;
        } else { G.last_op[i] = OP_ATTACK; G.op_obj[i] = o; }
        G.next_reward[i] += reward + T.attack_penalty;
    }
    // ---- starve / recover
    for (int g = 0; g < W.G; g++) {
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        for (int i = 0; i < G.n; i++) {
            if (G.dead[i]) continue;
            if (T.step_recover > 0) G.hp[i] = fminf(T.hp, G.hp[i] + T.step_recover);
            else {
                G.hp[i] -= -T.step_recover;
                if (G.hp[i] < 0.0f) {
                    G.dead[i] = 1; G.next_reward[i] = T.dead_penalty;
                    const int2 fp = body_dims(W, G, T, i);
                    body_fill(W, G.x[i], G.y[i], fp.x, fp.y, OCC_EMPTY);
                    W.counters[dead_slot(g, 0)] += 1;
                }
            }
        }
    }
    // ---- turns, then moves: lists in number order, each in insertion order (a stable counting sort by list)
    int start[80];
    const int n_lists = 2 * (n_sep + 1);
    for (int l = 0; l < n_lists; l++) start[l] = 0;
    for (int e = 0; e < M; e++) start[mlist[e].z]++;
    for (int l = 0, run = 0; l < n_lists; l++) { const int c = start[l]; start[l] = run; run += c; }
    for (int e = 0; e < M; e++) msorted[start[mlist[e].z]++] = mlist[e];
    for (int e = 0; e < M; e++) {
        const int self = msorted[e].x, g = ref_group(self), i = ref_index(self), a = msorted[e].y;
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        if (G.dead[i]) continue;
        const int2 fp = body_dims(W, G, T, i);
        if (a >= T.n_move) {
            // Map::do_turn about the body's reference corner (turn offsets are 0, AgentType.cc:108): the new direction from the action
            // number as the reference computes it (turned_dir), the footprint transposed
            const int dir = G.dir[i], ndir = turned_dir(dir, a);
            int rx, ry, sx, sy;
            saved_to_real(dir, T.bw, T.bl, G.x[i], G.y[i], rx, ry);
            real_to_saved(ndir, T.bw, T.bl, rx, ry, sx, sy);
            const int2 nd = dims_for_dir(T, ndir);
            if (serial_blank_area(W, sx, sy, nd.x, nd.y, self)) {
                body_fill(W, G.x[i], G.y[i], fp.x, fp.y, OCC_EMPTY);
                G.dir[i] = ndir;
                body_fill(W, sx, sy, nd.x, nd.y, self);
                G.x[i] = sx; G.y[i] = sy;
            }
            continue;
        }
        if (G.absorbed[i]) continue;                                              // a goal that has taken a mover in stands still (GridWorld.cc:580)
        const int2 d = agent_delta(W, G, i, T.move_off, a);
        const int nx = G.x[i] + d.x, ny = G.y[i] + d.y;
        if (serial_blank_area(W, nx, ny, fp.x, fp.y, self)) {
            body_fill(W, G.x[i], G.y[i], fp.x, fp.y, OCC_EMPTY);
            body_fill(W, nx, ny, fp.x, fp.y, self);
            G.x[i] = nx; G.y[i] = ny;
        } else if (!(nx < 0 || ny < 0 || nx + fp.x >= W.w || ny + fp.y >= W.h)) {
            // Map::get_collide: the first agent met, x outer, y inner (walls and food are no objects)
            int o = -1;
            for (int bx = 0; bx < fp.x && o < 0; bx++)
                for (int by = 0; by < fp.y; by++) {
                    const int v = W.occ[(ny + by) * W.w + nx + bx];
                    if (v >= 0 && v != self) { o = v; break; }
                }
            if (o < 0) continue;
            const GroupDev &O = W.grp[ref_group(o)];
            const int oi = ref_index(o);
            if (W.type[ref_group(o)].can_absorb) {                                // Map.cc:341-350: the first mover to bump into a goal is taken in
                if (O.absorbed[oi]) continue;                                     // a goal that is already taken: nothing happens, not even a collide
                O.absorbed[oi] = 1; O.hp[oi] = O.hp[oi] * 2;
                G.dead[i] = 1;                                                    // dead without counting in dead_ct (Map.cc:345-346)
                body_fill(W, G.x[i], G.y[i], fp.x, fp.y, OCC_EMPTY);
                W.counters[CTR_TAKEN + g] += 1;
            }
            G.last_op[i] = OP_COLLIDE; G.op_obj[i] = o;
        }
    }
    // ---- the step's pending actions are consumed
    for (int g = 0; g < W.G; g++) for (int i = 0; i < W.grp[g].n; i++) W.grp[g].pend[i] = PEND_NONE;
}
// the actions a group's pending actions came from (the first call of a step, when a second one follows)
__global__ void __launch_bounds__(256) k_pend_to_actions(GroupDev G, TypeDev T, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) out[i] = G.pend[i] == PEND_NONE ? T.n_move + T.n_turn + T.n_attack : pend_action(G.pend[i], T);   // (an action outside the space stays one)
}
void launch_step_serial(hipStream_t s, const WorldView &W, const SerialCall *calls, int n_calls, int2 *alist, int4 *mlist, int4 *msorted, int n_sep, int4 *events) {
    hipLaunchKernelGGL(k_step_serial, dim3(1), dim3(64), 0, s, W, calls, n_calls, alist, mlist, msorted, n_sep, events);
}
// does any of these actions do more than stand still?  (a group of goals that is only ever told to stay where it is steps through the
// parallel phases like a group that was given no actions; anything else -- a displacement, a turn, an attack, an action outside the space --
// sends the step through the literal loop: Env::set_action_device)
__global__ void __launch_bounds__(256) k_any_real_action(const int *actions, int n, TypeDev T, const int2 *delta, int *flag) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool real = false;
    if (i < n) {
        const int a = actions[i];
        real = true;
        if (a >= 0 && a < T.n_move) { const int2 d = delta[T.move_off + a]; real = (d.x | d.y) != 0; }
    }
    if (__ballot(real) && lane_id() == 0) *flag = 1;
}
void launch_any_real_action(hipStream_t s, const int *actions, int n, const TypeDev &T, const int2 *delta, int *flag) {
    if (n > 0) hipLaunchKernelGGL(k_any_real_action, dim3((n + 255) / 256), dim3(256), 0, s, actions, n, T, delta, flag);
}
void launch_pend_to_actions(hipStream_t s, const GroupDev &G, const TypeDev &T, int *out) {
    if (G.n > 0) hipLaunchKernelGGL(k_pend_to_actions, dim3((G.n + 255) / 256), dim3(256), 0, s, G, T, out);
}

void launch_step_report(hipStream_t s, int *counters, StepRecord *rec, int seq, int NG) {
    hipLaunchKernelGGL(k_step_report, dim3(1), dim3(64), 0, s, counters, rec, seq, NG, report_mode());
}
void launch_step_reset(hipStream_t s, int *counters) { hipLaunchKernelGGL(k_step_reset, dim3(1), dim3(64), 0, s, counters); }
void launch_set_rng(hipStream_t s, int *counters, unsigned x) { hipLaunchKernelGGL(k_set_rng, dim3(1), dim3(64), 0, s, counters, x); }
void launch_set_counter(hipStream_t s, int *counters, int index, int value, int unless_index) {
    hipLaunchKernelGGL(k_set_counter, dim3(1), dim3(64), 0, s, counters, index, value, unless_index);
}

// (the hit bits have an array of their own, WorldView::hitbits -- until round 4 they shared the move phase's claim words)
void launch_attack_rank(hipStream_t s, const WorldView &W, const GroupDev *gtab, const int *rank, const ShuffleBufs &B, bool clear_hitbits, const int *sums, const int *wpre,
                        const SeqPlan &P) {
    if (clear_hitbits) (void)hipMemsetAsync(W.hitbits, 0, sizeof(unsigned) * (size_t)W.w * W.h, s);   // (else k_shuffle_draw did it, or the fused step keeps them zero)
    hipLaunchKernelGGL(k_attack_rank, grid_all(W, 256), dim3(256), 0, s, W, gtab, rank, W.hitbits, B.head, B.first, sums, wpre, P);
}
static int att_threads(int kmax) {
    static const int forced = tune("att_threads", 0);
    if (forced == 64 || forced == 128 || forced == 256) return forced;
    return kmax <= 16 ? 256 : kmax <= 32 ? 128 : 64;     // <= 32 KB of hit lists per workgroup
}
// hit lists above the default dynamic-LDS limit have to be asked for (checked once, at reset)
bool attack_lds_ok(int kmax) {
    const size_t lds = (size_t)kmax * att_threads(kmax) * 8;
    if (lds <= (48u << 10)) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_attack_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess &&
           hipFuncSetAttribute(reinterpret_cast<const void *>(k_food_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess;
}
void launch_attack_iter(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int round, int kmax, int flag) {
    const int ATT_THREADS = att_threads(kmax);
    size_t lds = (size_t)kmax * ATT_THREADS * 8;
    hipLaunchKernelGGL(k_attack_eval, grid_all(W, ATT_THREADS), dim3(ATT_THREADS), lds, s, W, gtab, ttab, round, (const unsigned *)W.hitbits, kmax, flag);
    if (W.food_mode) launch_food_iter(s, W, gtab, ttab, round, kmax, flag);   // the food cells are part of the same fixed point
}
void launch_food_iter(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int round, int kmax, int flag) {
    const int NT = att_threads(kmax);
    hipLaunchKernelGGL(k_food_eval, dim3((W.w * W.h + NT - 1) / NT), dim3(NT), (size_t)kmax * NT * 8, s, W, gtab, ttab, round,
                       (const unsigned *)W.hitbits, kmax, flag);
}
void launch_attack_events(hipStream_t s, const WorldView &W, int4 *ev) {
    hipLaunchKernelGGL(k_attack_events, grid_all(W, 256), dim3(256), 0, s, W, ev);
}
void launch_attack_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int kmax) {
    (void)kmax;
    hipLaunchKernelGGL(k_attack_apply, grid_all(W, 256), dim3(256), 0, s, W, gtab, ttab, (const unsigned *)W.hitbits);
    if (W.food_mode) hipLaunchKernelGGL(k_food_apply, dim3((W.w * W.h + 255) / 256), dim3(256), 0, s, W, (const unsigned *)W.hitbits);
}

void launch_move_prep(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    const size_t words = 2 * (size_t)W.w * W.h;
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_move_prep, g, dim3(256), 0, s, W, (unsigned *)W.claim, words);   // starve + claim reset + candidates
    hipLaunchKernelGGL(k_move_claim, g, dim3(256), 0, s, W, gtab);
    hipLaunchKernelGGL(k_move_init, g, dim3(256), 0, s, W);
}
// the per-cell "wanted" counters live in the claim array (unused by the generic path otherwise)
void launch_movg_prep(hipStream_t s, const WorldView &W, bool starve) {
    const size_t words = (size_t)W.w * W.h;
    hipLaunchKernelGGL(k_fill32_gated, dim3((unsigned)std::min<size_t>((words + 255) / 256, 2048)), dim3(256), 0, s, W, (unsigned *)W.claim, 0u, words);
    hipLaunchKernelGGL(k_movg_prep, grid_all(W, 256), dim3(256), 0, s, W, (unsigned *)W.claim, starve ? 1 : 0);
}
// turn_mode with generic bodies: starvation + turn candidates, sweeps (host-checked), commit
void launch_turn_prep(hipStream_t s, const WorldView &W) {
    const size_t words = (size_t)W.w * W.h;
    hipLaunchKernelGGL(k_fill32_gated, dim3((unsigned)std::min<size_t>((words + 255) / 256, 2048)), dim3(256), 0, s, W, (unsigned *)W.claim, 0u, words);
    hipLaunchKernelGGL(k_turn_prep, grid_all(W, 256), dim3(256), 0, s, W, (unsigned *)W.claim);
}
void launch_turn_sweep(hipStream_t s, const WorldView &W, const GroupDev *gtab, int flag) {
    hipLaunchKernelGGL(k_turn_sweep, grid_all(W, 256), dim3(256), 0, s, W, gtab, (const unsigned *)W.claim, flag);
}
void launch_turn_apply(hipStream_t s, const WorldView &W) {
    hipLaunchKernelGGL(k_turn_vacate, grid_all(W, 256), dim3(256), 0, s, W);
    hipLaunchKernelGGL(k_turn_enter, grid_all(W, 256), dim3(256), 0, s, W);
}
void launch_movg_sweep(hipStream_t s, const WorldView &W, const GroupDev *gtab, int flag) {
    hipLaunchKernelGGL(k_movg_sweep, grid_all(W, 256), dim3(256), 0, s, W, gtab, (const unsigned *)W.claim, flag);
}
void launch_movg_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_movg_collide, g, dim3(256), 0, s, W, gtab, (const unsigned *)W.claim);
    hipLaunchKernelGGL(k_movg_vacate, g, dim3(256), 0, s, W);
    hipLaunchKernelGGL(k_movg_enter, g, dim3(256), 0, s, W);
}
// the step of plain games behind the shuffle: k_plain_rank, rounds of k_plain_eval, then k_strike, k_plain_commit
// (launch_plain_tail).  `rules`: the compiled rules, if every one of them pays the attacker of one event only (fused_rules); else
// null, and launch_rules runs behind the commit as usual
bool fused_rules(const RuleArgs *rules, int n) {
    if (n > 4) return false;
    // (attack-phase events only: `collide` is decided by the move phase, behind k_strike)
    for (int k = 0; k < n; k++) if (rules[k].pair || rules[k].prog >= 0 || rules[k].n_obj || (rules[k].op != OP_ATTACK && rules[k].op != OP_KILL)) return false;
    return true;
}
StrikeRules strike_rules(const RuleArgs *rules, int n_rules) {
    StrikeRules R{};
    if (rules) {
        R.n = n_rules;
        for (int k = 0; k < n_rules; k++) {
            R.r[k].ga = rules[k].ga; R.r[k].gb = rules[k].gb; R.r[k].op = rules[k].op; R.r[k].rule_no = rules[k].rule_no; R.r[k].n_subj = rules[k].n_subj;
            for (int q = 0; q < 4; q++) R.r[k].v[q] = rules[k].v_subj[q];
        }
    }
    return R;
}
void launch_plain_rank(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const ShuffleBufs &B, const int *sums,
                       const int *wpre, const SeqPlan &P) {
    hipLaunchKernelGGL(k_plain_rank, grid_all(W, 256), dim3(256), 0, s, W, PW, ptab, B, sums, wpre, P);
}
size_t plain_eval_lds(int kmax) { return (size_t)kmax * 256 * 8; }
bool plain_eval_lds_ok(int kmax) {
    if (plain_eval_lds(kmax) <= (48u << 10)) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_plain_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plain_eval_lds(kmax)) == hipSuccess;
}
void launch_plain_eval(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag,
                       const ShuffleBufs &B) {
    hipLaunchKernelGGL(k_plain_eval, grid_all(W, 256), dim3(256), plain_eval_lds(PW.kmax), s, W, PW, ptab, gtab, ttab, round, flag, B.head, B.first);
}
// the draws of the attack shuffle alone (the plain pipeline: every attacker chases its own list entry in k_plain_rank)
void launch_shuffle_draw(hipStream_t s, int n_max, int *counters, const ShuffleBufs &B, const unsigned *powtab, bool tiled) {
    hipLaunchKernelGGL(k_shuffle_draw, dim3((n_max + 255) / 256), dim3(256), 0, s, counters, B.j, B.head, B.first, B.link, (unsigned *)nullptr, (size_t)0, powtab, tiled ? 1 : 0);
}
void launch_plain_tail(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab,
                       const RuleArgs *rules, int n_rules, StepRecord *rec, int seq) {
    const StrikeRules R = strike_rules(rules, n_rules);
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_strike, g, dim3(256), 0, s, W, PW, ptab, gtab, ttab, R);
    // (rec != null: the step's report goes out from the first wave of the commit's launch -- see k_plain_commit)
    hipLaunchKernelGGL(k_plain_commit, g, dim3(256), 0, s, W, PW, rec, seq, report_mode());
}

void launch_move_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_move_commit, g, dim3(256), 0, s, W, gtab);
}

void launch_rule(hipStream_t s, const WorldView &W, const RuleArgs &A) {
    int na = W.grp[A.ga].n, nb = W.grp[A.gb].n;
    RuleBatch one{};
    one.r[0] = A;
    if (A.pair) {
        int ny = W.grp[A.gy].n;
        if (na <= 0 || ny <= 0 || nb <= 0) return;
        dim3 grid((std::max(na, ny) + 255) / 256, A.ga == A.gy ? 1 : 2);
        hipLaunchKernelGGL(k_pair_link, grid, dim3(256), 0, s, W, A);
        hipLaunchKernelGGL(k_pair_pay, grid, dim3(256), 0, s, W, A);
        hipLaunchKernelGGL(k_pair_obj, dim3((nb + 255) / 256), dim3(256), 0, s, W, A);
        return;
    }
    if (na > 0) hipLaunchKernelGGL(k_rule, dim3((na + 255) / 256), dim3(256), 0, s, W, one);
    if (A.n_obj && na > 0 && nb > 0) hipLaunchKernelGGL(k_rule_obj, dim3((nb + 255) / 256), dim3(256), 0, s, W, A);
}
// all rules of a step, in order.  Consecutive subject-only rules that pay different groups touch disjoint rewards: their
// order among each other is not observable and they share one launch.
void launch_rules(hipStream_t s, const WorldView &W, const RuleArgs *rules, int n, const RuleProg *progs, const GroupDev *gtab) {
    for (int k = 0; k < n;) {
        if (rules[k].prog >= 0) {
            const RuleProg &P = progs[rules[k].prog];
            const int na = W.grp[P.ga].n;
            if (na > 0) hipLaunchKernelGGL(k_rule_prog, dim3((na + 255) / 256), dim3(256), 0, s, W, gtab, P);
            if (P.n_obj && na > 0 && W.grp[P.gb].n > 0)
                hipLaunchKernelGGL(k_rule_obj, dim3((W.grp[P.gb].n + 255) / 256), dim3(256), 0, s, W, rules[k]);
            k++;
            continue;
        }
        RuleBatch B{};
        int m = 0, mx = 0;
        unsigned paid = 0;
        while (k + m < n && m < 4) {
            const RuleArgs &a = rules[k + m];
            if (a.pair || a.prog >= 0 || a.n_obj || (paid >> a.ga & 1u)) break;
            paid |= 1u << a.ga;
            B.r[m++] = a;
            mx = std::max(mx, W.grp[a.ga].n);
        }
        if (m >= 2) {
            if (mx > 0) hipLaunchKernelGGL(k_rule, dim3((mx + 255) / 256, m), dim3(256), 0, s, W, B);
            k += m;
        } else launch_rule(s, W, rules[k++]);
    }
}
void launch_finish(hipStream_t s, const WorldView &W) { hipLaunchKernelGGL(k_finish, grid_all(W, 256), dim3(256), 0, s, W); }

void launch_get_reward(hipStream_t s, const GroupDev &G, float group_reward, float *out) {
    if (G.n > 0) hipLaunchKernelGGL(k_get_reward, dim3((G.n + 255) / 256), dim3(256), 0, s, G, group_reward, out);
}
void launch_get_pos(hipStream_t s, const GroupDev &G, int *out) {
    if (G.n > 0) hipLaunchKernelGGL(k_get_pos, dim3((G.n + 255) / 256), dim3(256), 0, s, G, out);
}
void launch_get_alive(hipStream_t s, const GroupDev &G, unsigned char *out) {
    if (G.n > 0) hipLaunchKernelGGL(k_get_alive, dim3((G.n + 255) / 256), dim3(256), 0, s, G, out);
}

void launch_init_reward(hipStream_t s, const WorldView &W, int g) {
    int n = W.grp[g].n;
    if (n > 0) hipLaunchKernelGGL(k_init_reward, dim3((n + 255) / 256), dim3(256), 0, s, W, g);
}
bool compact_is_solo(int n) { return n <= scan_solo_max(); }
void launch_compact(hipStream_t s, const WorldView &W, int g, const GroupDev &D, int new_n, int *sums) {   // small groups: one workgroup
    (void)new_n; (void)sums;
    if (W.grp[g].n > 0) hipLaunchKernelGGL(k_compact_solo, dim3(1), dim3(SOLO_THREADS), 0, s, W, g, D);
}
void launch_clear_compact(hipStream_t s, const WorldView &W, const ClearArgs &A_in, int *sums, const MiniArgs &M, int *counts) {
    ClearArgs A = A_in;
    if (A.sums_per_tile <= 0) A.sums_per_tile = 1;
    int mx = 1;
    bool any = false;
    for (int g = 0; g < W.G; g++) { mx = std::max(mx, W.grp[g].n); any |= A.mode[g] == 2; }
    dim3 grid((mx + SCAN_TILE - 1) / SCAN_TILE, W.G);
    if (any && A.sums_per_tile == 1) hipLaunchKernelGGL(k_clear_count, grid, dim3(SCAN_THREADS), 0, s, W, A, sums);    // (else: k_strike counted)
    hipLaunchKernelGGL(k_clear_compact, grid, dim3(SCAN_THREADS), sizeof(int) * (size_t)M.vh * M.vw, s, W, A, sums, M, counts);
}
void launch_mini_norm(hipStream_t s, const WorldView &Wn, const MiniArgs &M, int *counts) {
    hipLaunchKernelGGL(k_mini_norm, dim3((Wn.G * M.vh * M.vw + 255) / 256), dim3(256), 0, s, Wn, M, counts);
}
void launch_clear_finish(hipStream_t s, const WorldView &Wn, const ClearArgs &A, GroupDev *gtab, TypeDev *ttab, const MiniArgs &M, int *counts) {
    const int blocks = std::max(1, (Wn.G * M.vh * M.vw + 255) / 256);
    hipLaunchKernelGGL(k_clear_finish, dim3(blocks), dim3(256), 0, s, Wn, A, gtab, ttab, M, counts);
}


}  // namespace magent_amd
