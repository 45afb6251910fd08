// plain_dev.h -- device bodies of the multi-launch pipeline of plain games (one-cell bodies, no turn_mode / food_mode / goals / kill_supply:
// battle, gather) and of the launches around it that a whole environment cycle needs: the tiled set_action, the shuffle's draws, k_plain_rank,
// k_plain_eval, k_strike, k_plain_commit, get_reward, clear_dead's compaction.  Two sets of kernels run them:
//   step.hip : one environment per launch, the world description a by-value kernel argument (Env::step_begin ...)
//   pipe.hip : MANY environments per launch (env_cycle_many), blockIdx.z = environment, the descriptions in a device array of PipeItem
// The bodies take the grid position from blockIdx.x (tile / 256 agents of a group) and blockIdx.y (group) in both.
#pragma once
#include "kernels_dev.h"

namespace magent_amd {

// GridWorld::set_action (GridWorld.cc:403-454) for one tile of SCAN_TILE agents of group g: pending actions, move keys, the tile's attack counts
__device__ __forceinline__ void set_action_tile_body(const WorldView &W, int g, const int *actions, int call_base, int *sums, int *wpre, int tile_off) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int tile0 = blockIdx.x * SCAN_TILE;
    // (every load of the tile first -- SCAN_ITEMS independent requests per thread -- then the classification: one round trip
    // per launch instead of eight)
    int act[SCAN_ITEMS], xs[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int i = tile0 + k * SCAN_THREADS + threadIdx.x;
        act[k] = i < G.n ? actions[i] : 0;
        xs[k] = (i < G.n && W.large_map) ? G.x[i] : 0;
    }
    __shared__ int s_w[SCAN_WAVES];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int i = tile0 + k * SCAN_THREADS + threadIdx.x;
        const bool attack = i < G.n && act[k] >= T.n_move + T.n_turn && act[k] < T.n_move + T.n_turn + T.n_attack;
        const int cnt = __popcll(__ballot(attack));
        if (lane_id() == 0) s_w[k * (SCAN_THREADS / 64) + (threadIdx.x >> 6)] = cnt;     // wave (k, w) holds agents tile0 + 64 (4 k + w) ...
        if (i < G.n) {
            const int a = act[k];
            if (a < 0 || a >= T.n_move + T.n_turn + T.n_attack) {   // outside the action space: no action, reported at the end of the step
                W.counters[CTR_BAD_ACTION] = 1;
                G.pend[i] = PEND_NONE;
            } else if (a < T.n_move + T.n_turn) {   // moves and (turn_mode) turns: ordered by stripe class, then insertion
                unsigned bound = 0;
                if (W.large_map) { int x_ = xs[k] % W.bandwidth; bound = (x_ < 4 || x_ > W.bandwidth - 4) ? 1u : 0u; }
                G.pend[i] = (a < T.n_move ? PEND_MOVE : PEND_TURN) | a;
                G.key[i] = (bound << 31) | (unsigned)(call_base + i);
            } else {
                G.pend[i] = PEND_ATTACK | (a - T.n_move - T.n_turn);
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < SCAN_WAVES) {
        int before = 0;
        for (int v = 0; v < (int)threadIdx.x; v++) before += s_w[v];
        wpre[(size_t)(tile_off + blockIdx.x) * SCAN_WAVES + threadIdx.x] = before;
        if (threadIdx.x == SCAN_WAVES - 1) {
            const int tot = before + s_w[SCAN_WAVES - 1];
            sums[tile_off + blockIdx.x] = tot;
            // (one atomic per tile, on ATT_SLOTS different cache lines: 782 of them on ONE word serialise at ~15 ns apiece -- measured:
            // the launch went from 4.6 to 11.9 us; k_shuffle_draw adds the slots up into CTR_ATTACK)
            if (tot) atomicAdd(&W.counters[att_slot((tile_off + blockIdx.x) % ATT_SLOTS)], tot);
        }
    }
}
// (the draws of this step's list and the hit words' zero-fill.  The list's length: counters[CTR_ATTACK] when the one-workgroup
// set_action left it there, else -- `tiled` -- the sum of the spread counters of k_set_action_a, which workgroup 0 then leaves in
// CTR_ATTACK for every later launch of the step)
__device__ __forceinline__ void shuffle_draw_launch_body(int *counters, int *j, int *head, int *first, int *link, unsigned *hitbits, size_t ncell,
                                                         const unsigned *powtab, int tiled) {
    int A;
    if (tiled) {
        __shared__ int s_a;
        if (threadIdx.x < 64) {
            int v = threadIdx.x < ATT_SLOTS ? counters[att_slot(threadIdx.x)] : 0;
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) v += __shfl_down(v, d);
            if (threadIdx.x == 0) s_a = v;
        }
        __syncthreads();
        A = s_a;
        if (blockIdx.x == 0 && threadIdx.x == 0) counters[CTR_ATTACK] = A;
    } else A = counters[CTR_ATTACK];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    // the per-cell hit words of the coming attack phase start from zero (they share the move phase's claim array)
    if (hitbits && A > 0) for (size_t c = i; c < ncell; c += (size_t)gridDim.x * blockDim.x) hitbits[c] = 0u;
    if (i >= A) return;
    shuffle_draw_body((unsigned)counters[CTR_RNG], i, j, head, first, link, powtab);
}

// The end-of-step report of the multi-launch step, straight into pinned host memory (the host spins on `seq`: a stream
// synchronisation behind a device-to-host copy costs several times the PCIe write it waits for), and the per-step counters
// back to zero -- unless a phase was left open: then the host continues from exactly this state and resets afterwards.
// `mode`: where `rec` lives.  REPORT_HOST: pinned host memory, `seq` behind a system-scope release (which writes the L2's dirty lines back
// first: ~10 us when a step's kernels have just been through it).  REPORT_DEVICE: device memory, read by a later launch (pipe.hip:
// k_pipe_finish sends every environment's report to the host in one piece, behind ONE such release).
// (Tried in round 6 and withdrawn: "wait until the wave's stores are acknowledged, then send `seq` with a relaxed store" instead of the
// release -- the acknowledgement comes from the L2, not from host memory: the host saw `seq` ahead of the record, profiles/r06_summary.md.)
constexpr int REPORT_HOST = 1, REPORT_DEVICE = 0;
__device__ __forceinline__ void step_report_body(int *counters, StepRecord *rec, int seq, int NG, int mode = REPORT_HOST) {      // (one wave: threads 0..63)
    const int tid = threadIdx.x;
    const int oa = counters[CTR_OPEN_ATTACK], om = counters[CTR_OPEN_MOVE];
    const bool open = (oa | om) != 0;
    const bool trig = tid < CTR_TRIGGER_END - CTR_TRIGGER && counters[CTR_TRIGGER + tid] != 0;
    const unsigned long long mask = __ballot(trig);
    const unsigned long long rounds = __ballot(tid < ROUND_SLOTS && counters[CTR_ROUND_CHANGED + tid] != 0);
    if (tid < NG) {
        int d = 0;
        for (int k = 0; k < DEAD_SLOTS; k++) d += counters[dead_slot(tid, k)];
        rec->dead[tid] = d; rec->taken[tid] = counters[CTR_TAKEN + tid];
    }
    if (tid == 0) {
        rec->triggers = mask; rec->rounds_mask = (unsigned)rounds;
        rec->rng = (unsigned)counters[CTR_RNG];
        rec->last_a = counters[CTR_ATTACK];
        rec->unsupported = counters[CTR_UNSUPPORTED]; rec->pack_overflow = counters[CTR_PACK_OVERFLOW];
        rec->bad_action = counters[CTR_BAD_ACTION]; rec->hit_overflow = counters[CTR_HIT_OVERFLOW];
        rec->error = 0; rec->rounds_attack = 0; rec->rounds_move = 0; rec->n_marks = 0;
        rec->open_attack = oa; rec->open_move = om;
    }
    if (!open) {
        if (tid < CTR_TRIGGER_END - CTR_TRIGGER) counters[CTR_TRIGGER + tid] = 0;
        if (tid < ROUND_SLOTS) counters[CTR_ROUND_CHANGED + tid] = 0;
        if (tid < ATT_SLOTS) counters[att_slot(tid)] = 0;
        if (tid == 0) counters[CTR_ATTACK] = 0;
    }
    if (mode == REPORT_DEVICE) { if (tid == 0) rec->seq = seq; return; }
    __threadfence_system();
    if (tid == 0) __hip_atomic_store((int *)&rec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void plain_rank_body(const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const ShuffleBufs &B, const int *sums, const int *wpre, const SeqPlan &P) {
    const int A = W.counters[CTR_ATTACK];
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) {
        W.counters[CTR_CHANGED] = 0;   // attack rounds start
        // (every draw has read the old engine state: k_shuffle_draw ran before; nobody reads it in this launch)
        W.counters[CTR_LAST_A] = A;
        W.counters[CTR_RNG] = (int)rng_skip((unsigned)W.counters[CTR_RNG], (unsigned)A);
    }
    const int g = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int n = G.n;
    if ((int)(blockIdx.x * blockDim.x) >= n) return;
    const int pend = i < n ? G.pend[i] : PEND_NONE;
    const bool att = (pend & ~PEND_ARG) == PEND_ATTACK;
    int seq = -1;
    if (A != 0 && P.off[g] >= 0) seq = attack_seq(sums, wpre, P.off[g], i, att);   // (every thread of the workgroup)
    if (i >= n) return;
    const bool dead = G.dead[i];
    const int x = G.x[i], y = G.y[i];
    unsigned key = G.key[i];          // a move's order key -- or, from the one-workgroup set_action, the attack's sequence number
    int tgt = -1, t = -1;
    if (!dead && att) {
        // the attack's rank in the shuffled list: the chase of its own list entry (k_shuffle_chase's walk, by the agent itself -- no rank
        // array, one launch less); the lists are read-only in this launch and go back to zero in round 1 of k_plain_eval
        key = (unsigned)shuffle_chase_pos(seq >= 0 ? seq : (int)key, A, B.j, B.head, B.first, B.link);
        const int k = pend & PEND_ARG;
        const int2 d = W.delta[T.attack_off + k];
        const int tx = x + d.x, ty = y + d.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            const int o = W.occ[ty * W.w + tx];
            if (o >= 0 && (T.attack_in_group || ref_group(o) != g)) {       // Map::get_attack_obj (Map.cc:229-247)
                tgt = o;
                const PlainGroup TG = ptab[ref_group(o)];
                const int slot = T.attack_bit + k;
                glob(TG.hlist)[(size_t)ref_index(o) * PW.S + slot] = make_uint2(key, (unsigned)ref_pack(g, i));
                atomicOr(&glob(TG.hmask)[ref_index(o)], 1u << slot);
            }
        }
    } else if (!dead && (pend & ~PEND_ARG) == PEND_MOVE) {      // (an attacker that was dead before the step: its list entry exists, and does nothing)
        const int2 d = W.delta[T.move_off + (pend & PEND_ARG)];
        const int nx = x + d.x, ny = y + d.y;
        // is_blank_area bounds (Map.cc:455) for a 1x1 body; a zero move "succeeds" in place and never vacates.  Whether the cell is a wall
        // is looked up by k_strike, which reads the cell anyway: until then a mover into a wall counts as one that "may leave" -- whoever
        // claims its cell on that ground depends on its move, which fails: the same outcome as "occupied by somebody who stays"
        if ((d.x | d.y) != 0 && nx >= 0 && ny >= 0 && nx + 1 < W.w && ny + 1 < W.h) t = ny * W.w + nx;
    }
    PW.g[g].rec[i] = make_int4((int)key, dead ? -1 : RANK_INF, t, (int)MV_FAIL_SAME);    // (status: "no move, hp as it was" until k_strike knows better)
    PW.g[g].atk[i] = tgt;
    // hp as the attack phase leaves it unless somebody hits me (k_plain_eval overwrites it then): only claimants that must know whether
    // their occupant is about to starve read it of an agent that was not hit -- types that recover never starve
    if (!(T.step_recover > 0)) G.mv[i] = __float_as_uint(G.hp[i]);
}

// (s_rank / s_ref: the thread's hit list, stride NT, slot tid -- sort_hits)
__device__ __forceinline__ void plain_eval_body(const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag,
                                                int *shuf_head, int *shuf_first) {
    // (both counters requested before either is looked at: one trip to memory instead of two before a converged round returns)
    const int A = W.counters[CTR_ATTACK];
    const int prev_changed = W.counters[CTR_ROUND_CHANGED + ((round - 1) & (ROUND_SLOTS - 1))];
    if (A == 0) return;
    // the shuffle's list heads and first-hit words have been read for the last time (k_plain_rank): back to zero for their next use
    if (round == 1)
        for (int k = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < A; k += gridDim.x * gridDim.y * blockDim.x) {
            shuf_head[k] = 0; shuf_first[k] = 0;
        }
    // nobody's death rank changed in the round before: nobody is stamped for this one
    if (round > 1 && prev_changed == 0) return;
    extern __shared__ unsigned s_hit[];
    const int NT = blockDim.x, tid = threadIdx.x;
    unsigned *s_rank = s_hit;
    int *s_ref = (int *)(s_hit + PW.kmax * NT);
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + tid;
    const GroupDev &G = W.grp[g];
    if (i >= G.n) return;
    // no input of mine has changed since my last evaluation (stamps are PW.round_base + round: they count on from step to step, nothing
    // resets them; in round 1 everybody who is hit is evaluated)
    if (round > 1 && (int)((unsigned)G.drank_b[i] - (unsigned)(PW.round_base + round - 1)) < 0) return;
    unsigned mask = PW.g[g].hmask[i];
    if (!mask) return;                                   // nobody hits me: I stay alive (RANK_INF, the initial value)
    const int dr_cur = PW.g[g].rec[i].y;
    if (dr_cur == -1) return;                            // dead before the phase (never a target: it is off the map)
    int nh = 0;
    const uint2 *mine = PW.g[g].hlist + (size_t)i * PW.S;
    while (mask) {
        const int slot = __ffs(mask) - 1;
        mask &= mask - 1;
        const uint2 e = mine[slot];
        s_rank[nh * NT + tid] = e.x; s_ref[nh * NT + tid] = (int)e.y;
        nh++;
    }
    sort_hits(s_rank, s_ref, NT, tid, nh);
    // replay in rank order: a hit counts iff its attacker did not die at an EARLIER rank
    float hp = G.hp[i];
    int dr = RANK_INF;
    for (int k = 0; k < nh; k++) {
        const unsigned r = s_rank[k * NT + tid];
        const int a = s_ref[k * NT + tid];
        const int adr = glob(ptab[ref_group(a)].rec)[ref_index(a)].y;
        if ((unsigned)adr >= r) {
            hp -= ttab[ref_group(a)].damage;
            if (hp < 0.0f) { dr = (int)r; break; }       // death iff hp < 0 strictly (GridWorld.h:205)
        }
    }
    G.mv[i] = __float_as_uint(hp);                       // final once the death ranks are: k_strike takes it from here
    if (dr != dr_cur) {
        PW.g[g].rec[i].y = dr;
        const int reader = PW.g[g].atk[i];               // who reads my death rank: my target (is its attacker alive at that rank?)
        if (reader >= 0) glob(gtab[ref_group(reader)].drank_b)[ref_index(reader)] = PW.round_base + round;
        if (flag >= 0) W.counters[flag] = 1;             // (only the last round of a batch reports)
        W.counters[CTR_ROUND_CHANGED + (round & (ROUND_SLOTS - 1))] = 1;
    }
}

__device__ __forceinline__ void strike_body(const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, const StrikeRules &R) {
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    if ((int)(blockIdx.x * blockDim.x) >= G.n) return;
    // (the agent's own fields are requested before the counters are looked at: they travel together)
    const int il = i < G.n ? i : G.n - 1;
    const int pend = G.pend[il];
    const int4 me = PW.g[g].rec[il];                     // {key | rank, death rank, move target, -}
    const float hp0 = G.hp[il], nr0 = G.next_reward[il];
    const bool dead0 = G.dead[il];
    if (attack_open(W)) return;
    const bool attacked = W.counters[CTR_ATTACK] != 0;
    bool died = false, alive = false;
    unsigned trig = 0;
    if (i < G.n) {
        if (pend != PEND_NONE) G.last_action[i] = pend_action(pend, T);     // Agent::set_action's `last_action = act` (see k_set_action_a)
        bool dead = dead0;
        float hp = hp0;
        const unsigned hp_before = __float_as_uint(hp);
        float nr = nr0;
        const unsigned nr_before = __float_as_uint(nr);
        int last_op = OP_NULL, op_obj = -1;              // (what clear_dead left: with rules fused, the host has seen it run since the last step)
        // ---- the attack phase applied from the converged death ranks (attack_apply_body, one-cell bodies, no supply)
        if (attacked && !dead) {
            if (PW.g[g].hmask[i]) {                       // somebody hit me: my last evaluation left my hp; the mask was read for the last time
                PW.g[g].hmask[i] = 0u;
                hp = __uint_as_float(G.mv[i]);
            }
            if ((pend & ~PEND_ARG) == PEND_ATTACK) {
                const unsigned my_rank = (unsigned)me.x;
                const int tgt = PW.g[g].atk[i];
                int tgt_dr = RANK_INF;
                if (tgt >= 0) tgt_dr = glob(ptab[ref_group(tgt)].rec)[ref_index(tgt)].y;
                if ((unsigned)me.y >= my_rank) {             // alive at my turn (GridWorld.cc:479-480)
                    float own;
                    if (tgt < 0 || (unsigned)tgt_dr < my_rank) own = T.attack_penalty;   // blank, or the target died before my turn (Map.cc:229-231)
                    else {
                        float reward = 0.0f;
                        if ((unsigned)tgt_dr == my_rank) { last_op = OP_KILL; reward = ttab[ref_group(tgt)].kill_reward; }
                        else last_op = OP_ATTACK;
                        op_obj = tgt;
                        G.last_op[i] = (unsigned char)last_op; G.op_obj[i] = tgt;
                        own = reward + T.attack_penalty;     // add_reward(reward + attack_penalty) (GridWorld.cc:505)
                    }
                    nr += own;
                }
            }
            if (me.y != RANK_INF) { dead = died = true; nr = T.dead_penalty; }   // dead_penalty overwrites what was accumulated (GridWorld.h:207)
        }
        // ---- starve / recover (GridWorld.cc:519-542)
        if (!dead) {
            if (T.step_recover > 0) hp = fminf(T.hp, hp + T.step_recover);
            else {
                hp -= -T.step_recover;
                if (hp < 0.0f) { dead = died = true; nr = T.dead_penalty; }
            }
        }
        if (__float_as_uint(hp) != hp_before) G.hp[i] = hp;      // (most agents of a battle stand at full hp: stores only where something changed)
        if (died) G.dead[i] = 1;
        alive = !dead;
        // ---- calc_reward for the rules that pay their subject (rule_body; the reference visits the dead too, GridWorld.cc:681-692)
        for (int k = 0; k < R.n; k++) {
            if (R.r[k].ga != g) continue;
            if (op_obj >= 0 && ref_group(op_obj) == R.r[k].gb && last_op == R.r[k].op) {
                trig |= 1u << k;
                for (int q = 0; q < R.r[k].n_subj; q++) nr += R.r[k].v[q];
            }
        }
        if (__float_as_uint(nr) != nr_before) G.next_reward[i] = nr;
        // ---- my move: the claim on its target cell (move_prep_body + move_claim_body)
        if (!dead && me.z >= 0) {
            const int c = me.z;
            const unsigned key = (unsigned)me.x;
            int o = W.occ[c];
            bool ok = o == OCC_EMPTY;
            if (o == OCC_WALL) PW.g[g].rec[i].z = -1;    // no move at all (Map::is_blank_area): k_plain_commit sees a non-mover
            if (o >= 0) {
                const int4 oc = glob(ptab[ref_group(o)].rec)[ref_index(o)];
                const float orec = ttab[ref_group(o)].step_recover;
                bool gone = attacked && oc.y != RANK_INF;                                       // killed in this step's attack phase
                if (!gone && !(orec > 0)) gone = __uint_as_float(glob(gtab[ref_group(o)].mv)[ref_index(o)]) - (-orec) < 0.0f;   // ... or about to starve
                if (gone) { ok = true; o = OCC_EMPTY; }                                          // the cell is empty when the moves begin
                else ok = oc.z >= 0 && (unsigned)oc.x < key;                                     // the occupant may leave, and before my turn
            }
            PW.g[g].atk[i] = o;              // what my target cell holds when the moves begin, for k_plain_commit (a mover has no attack target)
            if (ok) atomicMin(&W.claim[c], claim_word(PW.epoch, key, ref_pack(g, i)));
        }
        // (the status k_plain_rank left says "hp as it was": corrected here where it is not)
        if (died) PW.g[g].rec[i].w = (int)MV_DIED;
        else if (__float_as_uint(hp) != hp_before) PW.g[g].rec[i].w = (int)MV_FAIL;
    }
    int wtot;
    wave_rank(died, wtot);
    if (wtot && lane_id() == 0) atomicAdd(&W.counters[dead_slot(g, blockIdx.x % DEAD_SLOTS)], wtot);
    for (int k = 0; k < R.n; k++)
        if (__ballot((trig >> k) & 1u) && lane_id() == 0) W.counters[CTR_TRIGGER + R.r[k].rule_no] = 1;
    // the survivors of this workgroup's 256 agents, for clear_dead's compaction (k_clear_count's pass: nobody dies after this launch)
    __shared__ int s_alive[4];
    const int alive_w = __popcll(__ballot(alive));
    if (lane_id() == 0) s_alive[threadIdx.x >> 6] = alive_w;
    __syncthreads();
    if (threadIdx.x == 0 && (int)(blockIdx.x * blockDim.x) < G.n) PW.alive[PW.alive_off[g] + blockIdx.x] = s_alive[0] + s_alive[1] + s_alive[2] + s_alive[3];
}

// The end of the step of plain games: who moves (Map::do_move, Map.cc:313-358, in key order), the map and the painted map brought up to date.
// One launch since round 4 (k_plain_init used to store every claim's static winner first): a mover decides from k_strike's records
//   * it is the winner of its target cell -- the live claim word there names it --
//   * and the cell is empty when the moves begin (atk = OCC_EMPTY), or its occupant is a winner that leaves in turn (the same question one
//     record further: the occupant's target, that cell's claim word, its occupant ...; chains are short, most end at the first record),
// reading only what k_strike left (rec, atk, claim): nothing this launch writes is read by another agent of it.  The killed and the starved
// leave the map here (Map::remove_agent, Map.cc:272, GridWorld.cc:536) unless a mover has claimed their cell -- that mover found the cell
// empty (k_strike's `gone`), so it succeeds and writes the cell itself; the same rule as for the cell a mover leaves behind.
__device__ __forceinline__ const int4 *plain_rec(const PlainWorld &PW, int NG, int gg) {
    const int4 *p = glob(PW.g[0].rec);
#pragma unroll
    for (int k = 1; k < MAXG; k++) if (k < NG) p = gg == k ? glob(PW.g[k].rec) : p;      // (NG is wave-uniform; the table sits in scalar registers)
    return p;
}
__device__ __forceinline__ const int *plain_atk(const PlainWorld &PW, int NG, int gg) {
    const int *p = glob(PW.g[0].atk);
#pragma unroll
    for (int k = 1; k < MAXG; k++) if (k < NG) p = gg == k ? glob(PW.g[k].atk) : p;
    return p;
}
// does mover `a` (a packed ref; its record `ra` already read) leave its cell?  MV_OK or MV_FAIL
__device__ __forceinline__ unsigned plain_leaves(const WorldView &W, const PlainWorld &PW, int a, int4 ra) {
    for (int hops = 0;; hops++) {
        if (ra.z < 0) return MV_FAIL;                                              // no move (or into a wall): it stays
        const unsigned long long cl = glob(W.claim)[ra.z];
        if (!claim_live(cl, PW.epoch) || claim_ref(cl) != a) return MV_FAIL;       // not the winner of its target
        const int o = plain_atk(PW, W.G, ref_group(a))[ref_index(a)];
        if (o == OCC_EMPTY) return MV_OK;
        a = o;                                                                     // succeeds iff its own occupant leaves (a lower key: the chain ends)
        ra = plain_rec(PW, W.G, ref_group(a))[ref_index(a)];
    }
}
__device__ __forceinline__ void plain_commit_body(const WorldView &W, const PlainWorld &PW, StepRecord *rec, int seq, int report_mode = REPORT_HOST) {
    // The step's report rides in the first wave of this launch when nothing it carries is decided by the moves (rec != null: the rules are
    // fused or there are none): deaths, rule triggers and the generator are final since k_strike -- a launch boundary ago -- and nothing
    // the report resets is read by this kernel.  The host has `done` while the moves run; its next launches queue up behind them.
    if (rec && (blockIdx.x | blockIdx.y) == 0 && threadIdx.x < 64) step_report_body(glob(W.counters), glob(rec), seq, W.G, report_mode);
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev G = glob_group(W.grp[g]);
    if (i >= G.n) return;
    // (my own record first, the question whether the attack rounds ran out behind it: the loads are in flight while the counter arrives)
    const int4 me = glob(PW.g[g].rec)[i];                          // {key, death rank, move target, k_strike's status}
    const int o = glob(PW.g[g].atk)[i];
    const int px = G.x[i], py = G.y[i];
    if (attack_open(W)) return;
    const int self = ref_pack(g, i), old = py * W.w + px, c = me.z;
    const bool died = (unsigned)me.w == MV_DIED;
    const bool mover = c >= 0 && !died;                      // (a move target implies "alive when the step began": k_plain_rank)
    const bool alive = me.y != -1 && !died;
    const bool chain = mover && o >= 0;
    // one level of independent loads: the claim word of my target, the record of my target's occupant, and the claim word of my own cell
    // where it is likely to be needed (the dead; movers into an empty cell -- most of them win it).  A mover behind an occupant asks for it
    // only once it knows that it moves: most of those do not, and every such word is a request of its own (PMC: 14 MB per step)
    const unsigned long long cl_c = glob(W.claim)[mover ? c : 0];
    const bool old_now = died || (mover && !chain);
    unsigned long long cl_old = glob(W.claim)[old_now ? old : 0];
    const int4 ro = plain_rec(PW, W.G, chain ? ref_group(o) : g)[chain ? ref_index(o) : i];
    const unsigned s_o = chain ? plain_leaves(W, PW, o, ro) : MV_OK;
    const bool winner = mover && claim_live(cl_c, PW.epoch) && claim_ref(cl_c) == self;
    const bool ok = winner && s_o == MV_OK;
    if (ok && !old_now) cl_old = glob(W.claim)[old];
    int cell = old;
    if (ok) {
        if (!claim_live(cl_old, PW.epoch)) { glob(W.occ)[old] = OCC_EMPTY; if (W.live_paint) vc_store(W, old, OCC_EMPTY, 0u); }   // nobody claimed my cell
        glob(W.occ)[c] = self;
        const int ny = c / W.w;
        G.x[i] = c - ny * W.w; G.y[i] = ny;
        cell = c;
    } else if (mover) {
        // Map::get_collide: what I ran into -- the occupant, or whoever took the cell before my turn (the lowest key: the claim's winner)
        int blocker;
        if (o == OCC_EMPTY) blocker = claim_ref(cl_c);
        else blocker = (s_o == MV_OK && (unsigned)ro.x < (unsigned)me.x) ? claim_ref(cl_c) : o;
        G.last_op[i] = OP_COLLIDE;
        G.op_obj[i] = blocker;
    } else if (died && !claim_live(cl_old, PW.epoch)) { glob(W.occ)[old] = OCC_EMPTY; if (W.live_paint) vc_store(W, old, OCC_EMPTY, 0u); }
    G.pend[i] = PEND_NONE;   // end of step: pending actions are consumed
    // live paint: every agent that moved or whose hp changed paints its cell (most agents of a battle stand at full hp: 4 of 5 stores saved)
    if (W.live_paint && alive && (ok || (unsigned)me.w == MV_FAIL))
        vc_store(W, cell, g, __float_as_uint(__fdiv_rn(G.hp[i], W.type[g].hp)));      // repaint_body for a 1 x 1 body
}
// ... then stable compaction into the alternate buffers + init_reward + re-index the map (groups with deaths), or
// Agent::init_reward alone (groups without)
// (M.vh > 0: the minimap of the NEXT observations rides along -- every block adds the survivors it handles to an LDS histogram of
// their minimap cells and flushes it with one global atomic per non-empty bin; k_clear_finish / k_mini_norm divide.  That is
// k_minimap + k_minimap_norm, two launches per cycle, gone: the positions pass through this kernel anyway)
// (reward_out != null: GridWorld::get_reward (GridWorld.cc:694-704) rides along -- every agent's next_reward + the group's, the dead included,
// read by the thread that resets it)
__device__ __forceinline__ void clear_compact_body(const WorldView &W, const ClearArgs &A, int mode, const int *sums, const MiniArgs &M, int *counts,
                                                   float *reward_out = nullptr, float group_reward = 0.0f) {
    extern __shared__ int s_hist[];
    const int g = blockIdx.y;
    const GroupDev &G = W.grp[g];
    if ((int)(blockIdx.x * SCAN_TILE) >= G.n) return;
    const float step_reward = W.type[g].step_reward;
    const int VHW = M.vh * M.vw;
    if (VHW > 0) {
        for (int k = threadIdx.x; k < VHW; k += SCAN_THREADS) s_hist[k] = 0;
        __syncthreads();
    }
    if (mode == 1) {
        for (int k = 0; k < SCAN_ITEMS; k++) {
            const int i = blockIdx.x * SCAN_TILE + k * SCAN_THREADS + threadIdx.x;
            if (i < G.n) {
                const float nr = G.next_reward[i];
                if (reward_out) reward_out[i] = nr + group_reward;
                G.last_reward[i] = nr; G.next_reward[i] = step_reward; G.last_op[i] = OP_NULL; G.op_obj[i] = -1;
                if (VHW > 0) atomicAdd(&s_hist[(G.y[i] / M.scale_h) * M.vw + G.x[i] / M.scale_w], 1);
            }
        }
    } else if (mode == 2) {
        const ClearArgs::Alt D = A.dst[g];
        const int bw = W.type[g].bw, bl = W.type[g].bl;
        // (the single-buffered state goes back to its rest values at every agent's OWN index -- all that matters are the positions below
        // the new size, and each is some thread's own; `dead` is read by that thread alone in this launch: no second pass for it)
        block_rank([&](int i) {
                       if (reward_out) reward_out[i] = G.next_reward[i] + group_reward;
                       const bool d = G.dead[i];
                       if (d) G.dead[i] = 0;
                       G.last_op[i] = OP_NULL; G.op_obj[i] = -1; G.pend[i] = PEND_NONE;
                       return !d;
                   },
                   [&](int i, int r) {
                       int x = G.x[i], y = G.y[i];
                       D.x[r] = x; D.y[r] = y; D.id[r] = G.id[i]; D.hp[r] = G.hp[i]; D.last_action[r] = G.last_action[i];
                       D.absorbed[r] = G.absorbed[i];
                       if (G.dir) D.dir[r] = G.dir[i];
                       D.last_reward[r] = G.next_reward[i];
                       D.next_reward[r] = step_reward;
                       { const int2 fp = W.turn_mode ? dims_for_dir(W.type[g], G.dir[i]) : make_int2(bw, bl); body_fill(W, x, y, fp.x, fp.y, ref_pack(g, r)); }
                       if (VHW > 0) atomicAdd(&s_hist[(y / M.scale_h) * M.vw + x / M.scale_w], 1);
                   },
                   G.n, block_prefix(sums + A.sums_off[g], blockIdx.x * A.sums_per_tile));
    }
    if (VHW > 0) {
        __syncthreads();
        for (int k = threadIdx.x; k < VHW; k += SCAN_THREADS)
            if (s_hist[k]) atomicAdd(&counts[((blockIdx.x % MINI_COPIES) * W.G + g) * VHW + k], s_hist[k]);   // MINI_COPIES histograms: same-address atomics serialise
    }
}

}  // namespace magent_amd
