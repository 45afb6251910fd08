// kernels.hip -- hand-written HIP kernels of the grid-world step engine for gfx950 (CDNA4, wave64).
//
// No dense contraction on this path: MFMA is unused on purpose.  The rules that matter are coalesced SoA access,
// LDS staging of the map windows, wide streaming stores, wave ballots for in-wave ranking and 64-bit global
// atomics (umin) for move arbitration.  Compiled with -ffp-contract=off; float ops keep the reference's order.
//
// Reference semantics each kernel restates (file:line into /root/reference/src/gridworld):
//   k_paint / k_minimap / k_render   GridWorld::get_observation GridWorld.cc:292-401, Map::extract_view Map.cc:129-207
//   k_set_action + scan trio         GridWorld::set_action GridWorld.cc:403-454
//   k_attack_*                       GridWorld::step attack loop GridWorld.cc:475-506, Map.cc:209-310, GridWorld.h:203-209
//   k_starve                         GridWorld.cc:519-542, GridWorld.h:194-201
//   k_move_*                         GridWorld.cc:574-613, Map::do_move Map.cc:313-358
//   k_rule*                          GridWorld::calc_reward GridWorld.cc:681-692, RewardEngine.cc:216-443
//   k_compact_* / k_init_reward      GridWorld::clear_dead GridWorld.cc:633-665, Agent::init_reward GridWorld.h:168-174
#include "engine.h"
#include "launch.h"

namespace magent_amd {

// ------------------------------------------------------------------------------------------------ small helpers
// division of a 32-bit unsigned by a runtime-constant divisor: round-up multiply-shift, exact for all 32-bit n
__device__ __forceinline__ unsigned fdiv_u32(unsigned n, FastDiv d) {
    unsigned t = __umulhi(n, d.mul);
    unsigned q = (t + ((n - t) >> 1)) >> d.shift;
    return d.one ? n : q;
}

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }

// number of set predicate bits in lanes below this one, and in the whole wave (wave64 ballot + mbcnt)
__device__ __forceinline__ int wave_rank(bool pred, int &wave_total) {
    unsigned long long m = __ballot(pred);
    wave_total = __popcll(m);
    return __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0));
}

// ------------------------------------------------------------------------------------------------ device tables
// copies the by-value group/type tables into device memory for kernels that index them per lane
__global__ void k_set_tables(WorldView W, GroupDev *gtab, TypeDev *ttab) {
    int i = threadIdx.x;
    if (i < MAXG) { gtab[i] = W.grp[i]; ttab[i] = W.type[i]; }
}

// ------------------------------------------------------------------------------------------------ paint
// viewcell[c] = {group | EMPTY | WALL, bits(hp / type.hp)}: one pass over the map, coalesced 4 B in / 8 B out.
// The division is the reference's `p->get_hp() / p->get_type().hp` (Map.cc:197), IEEE round-to-nearest.
__global__ void __launch_bounds__(256) k_paint(WorldView W, const GroupDev *gtab, const TypeDev *ttab) {
    const int ncell = W.w * W.h;
    for (int c = blockIdx.x * blockDim.x + threadIdx.x; c < ncell; c += gridDim.x * blockDim.x) {
        int o = W.occ[c];
        int2 rec = make_int2(o, 0);
        if (o >= 0) {
            int g = ref_group(o), i = ref_index(o);
            rec.x = g;
            rec.y = __float_as_int(__fdiv_rn(gtab[g].hp[i], ttab[g].hp));
        }
        W.viewcell[c] = rec;
    }
}

// ------------------------------------------------------------------------------------------------ minimap histogram
// counts[j][cell] = number of agents of group j whose (x / scale_w, y / scale_h) is cell (GridWorld.cc:341-352;
// dead-but-not-cleared agents are counted, as in the reference).  LDS int atomics per block, then one global
// atomic per non-empty bin.  blockIdx.y = group.
__global__ void __launch_bounds__(256) k_minimap(WorldView W, RenderArgs R, int *counts) {
    extern __shared__ int s_hist[];
    const int VHW = R.VH * R.VW, j = blockIdx.y;
    const GroupDev G = W.grp[j];
    if ((int)(blockIdx.x * blockDim.x) >= G.n) return;
    for (int k = threadIdx.x; k < VHW; k += blockDim.x) s_hist[k] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < G.n; i += gridDim.x * blockDim.x) {
        int cx = G.x[i] / R.scale_w, cy = G.y[i] / R.scale_h;
        atomicAdd(&s_hist[cy * R.VW + cx], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < VHW; k += blockDim.x)
        if (s_hist[k]) atomicAdd(&counts[j * VHW + k], s_hist[k]);
}

// ------------------------------------------------------------------------------------------------ observation render
// One workgroup renders AG consecutive agents of the observing group.
//   phase 0: agent positions + the G minimaps (count / total, float) into LDS
//   phase 1: each agent's VH x VW window of `viewcell` into LDS (masked by the view range and the map bounds)
//   phase 2: the AG * S output floats as a contiguous float4 stream, every element written exactly once
//            (zeros included: the reference's memset GridWorld.cc:310 is fused into the stores)
//   phase 3: the AG * F feature floats
// The output is the algorithmic traffic (4 * (S + F) bytes per agent); map reads come from L2 / LDS.
// blockIdx -> agent tile mapping is XCD-aware: the 8 XCDs each walk a contiguous eighth of the agent range, so
// spatially ordered groups keep each XCD's map window resident in its own L2.
template <bool VEC4, bool NT>
__global__ void __launch_bounds__(256) k_render(WorldView W, RenderArgs R, RenderPlan P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VHW = R.VH * R.VW, AG = P.AG, G = W.G;
    int2 *s_cell = (int2 *)smem;                     // [AG][VHW]
    float *s_mini = (float *)(s_cell + AG * VHW);    // [G][VHW]
    int *s_ax = (int *)(s_mini + G * VHW);           // [AG]
    int *s_ay = s_ax + AG;                           // [AG]
    int *s_self = s_ay + AG;                         // [AG]  minimap cell of the agent itself
    int *s_desc = s_self + AG;                       // [C]

    // XCD-aware tile index: hardware places block b on XCD b % 8 (speed only, never correctness)
    int tile = blockIdx.x;
    if (P.xcd_chunk > 0 && tile < P.xcd_chunk * 8) tile = (tile & 7) * P.xcd_chunk + (tile >> 3);
    const int a0 = tile * AG;
    const int nA = min(AG, R.n - a0);
    if (nA <= 0) return;
    const GroupDev Gd = W.grp[R.g];
    const TypeDev T = W.type[R.g];
    const int tid = threadIdx.x;

    // ---- phase 0
    if (tid < AG) {
        int x = 0, y = 0;
        if (tid < nA) { x = Gd.x[a0 + tid]; y = Gd.y[a0 + tid]; }
        s_ax[tid] = x; s_ay[tid] = y;
        s_self[tid] = R.minimap ? (y / R.scale_h) * R.VW + (x / R.scale_w) : -1;
    }
    if (tid < R.C) s_desc[tid] = R.chan_desc[tid];
    if (R.minimap) {
        for (int k = tid; k < G * VHW; k += 256) {
            int j = fdiv_u32(k, P.div_vhw);
            int cnt = R.mini_counts[k], tot = R.totals[j];
            // float(count) / float(total) as the reference (GridWorld.cc:350,356); float ++ saturates at 2^24;
            // an empty group divides 0 by 0: the x86 default NaN the reference produces is 0xFFC00000
            float v = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(cnt, 1 << 24), (float)(unsigned)tot);
            s_mini[k] = v;
        }
    }
    __syncthreads();

    // ---- phase 1: windows -> LDS
    const unsigned char *mask = W.mask + T.mask_off;
    for (int k = tid; k < nA * VHW; k += 256) {
        int a = fdiv_u32(k, P.div_vhw);
        int cell = k - a * VHW;
        int vy = fdiv_u32(cell, P.div_vw);
        int vx = cell - vy * R.VW;
        int mx = s_ax[a] + T.view_x1 + vx, my = s_ay[a] + T.view_y1 + vy;
        int2 rec = make_int2(OCC_EMPTY, 0);
        if (mask[cell] && mx >= 0 && mx < W.w && my >= 0 && my < W.h) rec = W.viewcell[my * W.w + mx];
        s_cell[k] = rec;
    }
    __syncthreads();

    // ---- phase 2: stream the view tensor
    const int total = nA * R.S;
    float *out = R.view + (size_t)a0 * R.S;
    auto value_at = [&](int a, int cell, int c) -> float {
        int desc = s_desc[c];
        int kind = desc >> 8, code = desc & 0xff;
        if (kind == 2) {                               // minimap channel of group `code`: unmasked copy + self marker
            float m = s_mini[code * VHW + cell];
            if (cell == s_self[a] && m == m) m += 1.0f;  // NaN stays the same NaN (x86 propagates the operand)
            return m;
        }
        int2 rec = s_cell[a * VHW + cell];
        bool hit = (rec.x & 0xff) == code;
        return hit ? (kind == 0 ? 1.0f : __int_as_float(rec.y)) : 0.0f;
    };
    if (VEC4) {
        const int nq = total >> 2;
        for (int q = tid; q < nq; q += 256) {
            unsigned e = (unsigned)q << 2;
            int a = fdiv_u32(e, P.div_s);
            int rem = e - a * R.S;
            int cell = fdiv_u32(rem, P.div_c);
            int c = rem - cell * R.C;
            float v[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                v[k] = value_at(a, cell, c);
                if (++c == R.C) { c = 0; if (++cell == VHW) { cell = 0; ++a; } }
            }
            v4f f4 = {v[0], v[1], v[2], v[3]};
            if (NT) __builtin_nontemporal_store(f4, (v4f *)out + q);
            else ((v4f *)out)[q] = f4;
        }
        for (int e = (nq << 2) + tid; e < total; e += 256) {   // < 4 trailing floats of the last tile
            int a = fdiv_u32(e, P.div_s);
            int rem = e - a * R.S;
            int cell = fdiv_u32(rem, P.div_c);
            out[e] = value_at(a, cell, rem - cell * R.C);
        }
    } else {
        for (int e = tid; e < total; e += 256) {
            int a = fdiv_u32(e, P.div_s);
            int rem = e - a * R.S;
            int cell = fdiv_u32(rem, P.div_c);
            out[e] = value_at(a, cell, rem - cell * R.C);
        }
    }

    // ---- phase 3: features [id bits x E | one-hot last_action x NA | last_reward | x / w | y / h]
    float *fo = R.feat + (size_t)a0 * R.F;
    for (int k = tid; k < nA * R.F; k += 256) {
        int a = fdiv_u32(k, P.div_f);
        int f = k - a * R.F;
        int i = a0 + a;
        float v = 0.0f;
        if (f < R.E) v = (f < 31) ? (float)((Gd.id[i] >> f) & 1) : 0.0f;
        else if (f < R.E + R.NA) v = (Gd.last_action[i] == f - R.E) ? 1.0f : 0.0f;
        else if (f == R.E + R.NA) v = Gd.last_reward[i];        // a fresh agent's last_action == NA lands here and
        else if (f == R.E + R.NA + 1) v = __fdiv_rn((float)s_ax[a], (float)W.w);  // is overwritten (GridWorld.cc:390-392)
        else if (f == R.E + R.NA + 2) v = __fdiv_rn((float)s_ay[a], (float)W.h);
        fo[k] = v;
    }
}

// ------------------------------------------------------------------------------------------------ block scan trio
// Exclusive prefix sum of a per-agent predicate over one group, SCAN_ITEMS elements per thread:
//   pass A  per-block totals            pass B  one block scans the totals (+ base)      pass C  per-element ranks
// In-wave ranks come from ballots (wave64), cross-wave from LDS.
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = 8, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;

template <class Pred>
__device__ __forceinline__ int block_count(Pred pred, int n) {
    __shared__ int s_w[SCAN_THREADS / 64];
    int base = blockIdx.x * SCAN_TILE, cnt = 0;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int i = base + k * SCAN_THREADS + threadIdx.x;
        bool p = i < n && pred(i);
        cnt += __popcll(__ballot(p));
    }
    if (lane_id() == 0) s_w[threadIdx.x >> 6] = cnt;
    __syncthreads();
    int tot = 0;
    for (int k = 0; k < SCAN_THREADS / 64; k++) tot += s_w[k];
    return tot;
}

// calls emit(i, exclusive_rank) for every i in this block's tile with pred(i)
template <class Pred, class Emit>
__device__ __forceinline__ void block_rank(Pred pred, Emit emit, int n, int block_offset) {
    __shared__ int s_w[SCAN_THREADS / 64];
    int base = blockIdx.x * SCAN_TILE, run = block_offset;
    const int wave = threadIdx.x >> 6;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int i = base + k * SCAN_THREADS + threadIdx.x;
        bool p = i < n && pred(i);
        int wtot, r = wave_rank(p, wtot);
        if (lane_id() == 0) s_w[wave] = wtot;
        __syncthreads();
        int before = 0, all = 0;
        for (int v = 0; v < SCAN_THREADS / 64; v++) { int t = s_w[v]; all += t; if (v < wave) before += t; }
        if (p) emit(i, run + before + r);
        run += all;
        __syncthreads();
    }
}

// pass B: exclusive scan of block totals in place; counter[slot] is the running base and receives the new total
__global__ void __launch_bounds__(1024) k_scan_blocks(int *sums, int nb, int *counter) {
    __shared__ int s_w[16];
    __shared__ int s_carry;
    if (threadIdx.x == 0) s_carry = counter ? *counter : 0;
    __syncthreads();
    for (int start = 0; start < nb; start += 1024) {
        int i = start + threadIdx.x;
        int v = i < nb ? sums[i] : 0;
        int x = v;  // inclusive scan inside the wave by shuffles
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { int y = __shfl_up(x, d); if (lane_id() >= d) x += y; }
        if (lane_id() == 63) s_w[threadIdx.x >> 6] = x;
        __syncthreads();
        int before = 0, all = 0;
        for (int k = 0; k < 16; k++) { int t = s_w[k]; all += t; if (k < (int)(threadIdx.x >> 6)) before += t; }
        if (i < nb) sums[i] = s_carry + before + x - v;
        __syncthreads();
        if (threadIdx.x == 0) s_carry += all;
        __syncthreads();
    }
    if (threadIdx.x == 0 && counter) *counter = s_carry;
}

// ------------------------------------------------------------------------------------------------ set_action
// Stores last_action, classifies the action (move | attack) and computes the agent's order key.
//   move  : key = (boundary << 31) | insertion index.  Reference: moves run stripe lists 0..S-1 then the boundary
//           list, each in insertion order (GridWorld.cc:605-613); interior moves of different stripes cannot
//           interact (margin 4 > max speed 3), so only "boundary after interior" + insertion order is observable.
//   attack: key = running sequence number in the attack list (the shuffle permutes these; pass C below).
__global__ void __launch_bounds__(SCAN_THREADS) k_set_action_a(WorldView W, int g, const int *actions, int call_base, int *sums) {
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    const int tile0 = blockIdx.x * SCAN_TILE;
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int i = tile0 + k * SCAN_THREADS + threadIdx.x;
        if (i < G.n) {
            int act = actions[i];
            G.last_action[i] = act;
            if (act < T.n_move) {
                unsigned bound = 0;
                if (W.large_map) { int x_ = G.x[i] % W.bandwidth; bound = (x_ < 4 || x_ > W.bandwidth - 4) ? 1u : 0u; }
                G.pend[i] = PEND_MOVE | act;
                G.key[i] = (bound << 31) | (unsigned)(call_base + i);
            } else {
                G.pend[i] = PEND_ATTACK | (act - T.n_move);
            }
        }
    }
    int tot = block_count([&](int i) { return actions[i] >= T.n_move; }, G.n);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

__global__ void __launch_bounds__(SCAN_THREADS) k_set_action_c(WorldView W, int g, const int *actions, const int *sums) {
    const GroupDev G = W.grp[g];
    const int n_move = W.type[g].n_move;
    block_rank([&](int i) { return actions[i] >= n_move; }, [&](int i, int r) { G.key[i] = (unsigned)r; }, G.n, sums[blockIdx.x]);
}

// ------------------------------------------------------------------------------------------------ attack phase
// rank[seq] = position of attack-list entry `seq` after the reference's shuffle (GridWorld.cc:464-468)
__global__ void __launch_bounds__(256) k_attack_rank(WorldView W, const int *rank) {
    const GroupDev G = W.grp[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    bool att = (G.pend[i] & ~PEND_ARG) == PEND_ATTACK;
    if (att) G.key[i] = (unsigned)rank[G.key[i]];
    G.drank_a[i] = G.dead[i] ? -1 : RANK_INF;   // agents dead before the phase never act and are not on the map
}

// Exact parallel form of the sequential attack loop.  For a target t the incoming hits are found by PULLING:
// for every attacker group g' and attack offset d of g', the only agent that can hit t with d stands at
// pos(t) - d; it hits iff its pending action is "attack with offset d".  Hits are sorted by rank (LDS) and replayed
// in order: a hit counts iff its attacker is still alive at that rank (death_rank[attacker] > rank).  death_rank
// is iterated to its fixed point; after k rounds every event of dependency depth <= k is final.
constexpr int ATT_THREADS = 128, ATT_KMAX = 32;

struct HitList {
    int n;
};

template <bool APPLY>
__global__ void __launch_bounds__(ATT_THREADS) k_attack_eval(WorldView W, const GroupDev *gtab, const TypeDev *ttab,
                                                             int use_b /* read drank_b, write drank_a */) {
    __shared__ unsigned s_rank[ATT_KMAX][ATT_THREADS];
    __shared__ int s_ref[ATT_KMAX][ATT_THREADS];
    const int g = blockIdx.y, tid = threadIdx.x;
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    const int i = blockIdx.x * blockDim.x + tid;
    if (i >= G.n) return;
    const int *dr_self_cur = use_b ? G.drank_b : G.drank_a;
    int *dr_self_next = use_b ? G.drank_a : G.drank_b;
    const int dr_me_cur = dr_self_cur[i];
    if (dr_me_cur == -1) { if (!APPLY) dr_self_next[i] = -1; return; }   // dead before the phase

    const int x = G.x[i], y = G.y[i];
    // ---- gather incoming hits
    int nh = 0;
    for (int ga = 0; ga < W.G; ga++) {
        const TypeDev TA = W.type[ga];
        if (TA.n_attack == 0 || (!TA.attack_in_group && ga == g)) continue;
        const GroupDev A = W.grp[ga];
        for (int k = 0; k < TA.n_attack; k++) {
            int2 d = W.delta[TA.attack_off + k];
            int ax = x - d.x, ay = y - d.y;
            if (ax < 0 || ax >= W.w || ay < 0 || ay >= W.h) continue;
            int o = W.occ[ay * W.w + ax];
            if (o < 0 || ref_group(o) != ga) continue;
            int ai = ref_index(o);
            if (A.pend[ai] != (PEND_ATTACK | k)) continue;
            if (nh < ATT_KMAX) { s_rank[nh][tid] = A.key[ai]; s_ref[nh][tid] = o; }
            nh++;
        }
    }
    if (nh > ATT_KMAX) nh = ATT_KMAX;   // cannot happen: the host checks sum(n_attack) <= ATT_KMAX at reset
    // ---- insertion sort by rank (ranks are unique)
    for (int a = 1; a < nh; a++) {
        unsigned r = s_rank[a][tid]; int f = s_ref[a][tid];
        int b = a - 1;
        while (b >= 0 && s_rank[b][tid] > r) { s_rank[b + 1][tid] = s_rank[b][tid]; s_ref[b + 1][tid] = s_ref[b][tid]; b--; }
        s_rank[b + 1][tid] = r; s_ref[b + 1][tid] = f;
    }
    // ---- own attack (needed for kill_supply replay and, in APPLY, for the attacker-side results)
    const int pend = G.pend[i];
    const bool attacker = (pend & ~PEND_ARG) == PEND_ATTACK;
    unsigned my_rank = 0xFFFFFFFFu;
    int tgt = -1;          // packed ref of my target at phase start, -1 = blank / wall / out of board / same group
    if (attacker) {
        my_rank = G.key[i];
        int2 d = W.delta[T.attack_off + (pend & PEND_ARG)];
        int tx = x + d.x, ty = y + d.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            int o = W.occ[ty * W.w + tx];
            if (o >= 0 && (T.attack_in_group || ref_group(o) != g)) tgt = o;
        }
    }
    // death rank of my target as of the current iterate
    int tgt_dr = RANK_INF;
    if (tgt >= 0) { const GroupDev TG = gtab[ref_group(tgt)]; tgt_dr = (use_b ? TG.drank_b : TG.drank_a)[ref_index(tgt)]; }
    const bool supply = W.any_kill_supply && tgt >= 0 && (unsigned)tgt_dr == my_rank;

    // ---- replay in rank order
    float hp = G.hp[i];
    int dr = RANK_INF;
    bool supplied = !supply;
    for (int k = 0; k < nh; k++) {
        unsigned r = s_rank[k][tid];
        if (!supplied && my_rank < r) { hp = fminf(T.hp, hp + ttab[ref_group(tgt)].kill_supply); supplied = true; }
        int a = s_ref[k][tid];
        const GroupDev A = gtab[ref_group(a)];
        int adr = (use_b ? A.drank_b : A.drank_a)[ref_index(a)];
        if ((unsigned)adr > r) {                       // attacker alive when its turn comes (RANK_INF > any rank)
            hp -= ttab[ref_group(a)].damage;
            if (hp < 0.0f) { dr = (int)r; break; }     // death iff hp < 0 strictly (GridWorld.h:205)
        }
    }
    if (!supplied && dr == RANK_INF) hp = fminf(T.hp, hp + ttab[ref_group(tgt)].kill_supply);

    if (!APPLY) {
        dr_self_next[i] = dr;
        if (dr != dr_me_cur) W.counters[CTR_CHANGED] = 1;
        return;
    }
    // ---- APPLY (the iterate has converged: dr == dr_me_cur)
    float nr = G.next_reward[i];
    if (attacker && (unsigned)dr > my_rank) {          // alive at my turn (GridWorld.cc:479-480)
        if (tgt < 0 || (unsigned)tgt_dr < my_rank) {   // blank, or the target died before my turn (Map.cc:229-231)
            nr += T.attack_penalty;
        } else {
            float reward = 0.0f;
            if ((unsigned)tgt_dr == my_rank) { G.last_op[i] = OP_KILL; reward = ttab[ref_group(tgt)].kill_reward; }
            else G.last_op[i] = OP_ATTACK;
            G.op_obj[i] = tgt;
            nr += reward + T.attack_penalty;           // add_reward(reward + attack_penalty) (GridWorld.cc:505)
        }
    }
    G.hp[i] = hp;
    if (dr != RANK_INF) {
        G.dead[i] = 1;
        nr = T.dead_penalty;                           // overwrites what was accumulated (GridWorld.h:207)
        W.occ[y * W.w + x] = OCC_EMPTY;
        atomicAdd(&W.counters[CTR_DEAD + g], 1);
    }
    G.next_reward[i] = nr;
}

// ------------------------------------------------------------------------------------------------ starve / recover
__global__ void __launch_bounds__(256) k_starve(WorldView W) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool died = false;
    if (i < G.n && !G.dead[i]) {
        float hp = G.hp[i];
        if (T.step_recover > 0) hp = fminf(T.hp, hp + T.step_recover);
        else {
            hp -= -T.step_recover;
            if (hp < 0.0f) { died = true; G.dead[i] = 1; G.next_reward[i] = T.dead_penalty; W.occ[G.y[i] * W.w + G.x[i]] = OCC_EMPTY; }
        }
        G.hp[i] = hp;
    }
    int wtot; wave_rank(died, wtot);
    if (wtot && lane_id() == 0) atomicAdd(&W.counters[CTR_DEAD + g], wtot);
}

// ------------------------------------------------------------------------------------------------ move phase
// Exact parallel form of "first come in key order, vacate-then-enter chains" (Map.cc:313-333):
//   a cell empty at phase start goes to its lowest-key contender; a cell occupied by O is freed at key(O) iff O's
//   own move succeeds, and then goes to the lowest-key contender with key > key(O).  Which contender that would be
//   is static (64-bit atomic umin of {key, ref} per cell); whether O leaves is a chain of such dependencies that
//   only points to lower keys, resolved by pointer jumping.
// tgt (= drank_a, free after the attack phase): target cell of a move candidate, -1 otherwise.
__global__ void __launch_bounds__(256) k_move_prep(WorldView W) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int t = -1;
    int pend = G.pend[i];
    if (!G.dead[i] && (pend & ~PEND_ARG) == PEND_MOVE) {
        int2 d = W.delta[T.move_off + (pend & PEND_ARG)];
        int nx = G.x[i] + d.x, ny = G.y[i] + d.y;
        // is_blank_area bounds (Map.cc:455) for a 1x1 body; a zero move "succeeds" in place and never vacates
        if ((d.x | d.y) != 0 && nx >= 0 && ny >= 0 && nx + 1 < W.w && ny + 1 < W.h && W.occ[ny * W.w + nx] != OCC_WALL)
            t = ny * W.w + nx;
    }
    G.drank_a[i] = t;
    G.mv[i] = MV_FAIL;
}

__global__ void __launch_bounds__(256) k_move_claim(WorldView W, const GroupDev *gtab) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int c = G.drank_a[i];
    if (c < 0) return;
    unsigned key = G.key[i];
    int o = W.occ[c];
    bool ok = o == OCC_EMPTY;
    if (o >= 0) {
        const GroupDev O = gtab[ref_group(o)];
        int oi = ref_index(o);
        ok = O.drank_a[oi] >= 0 && O.key[oi] < key;   // the occupant may leave, and before my turn
    }
    if (ok) atomicMin(&W.claim[c], ((unsigned long long)key << 32) | (unsigned)ref_pack(g, i));
}

__global__ void __launch_bounds__(256) k_move_init(WorldView W) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int c = G.drank_a[i];
    if (c < 0) return;
    unsigned long long cl = W.claim[c];
    if ((unsigned)cl != (unsigned)ref_pack(g, i)) return;          // not the static winner: stays MV_FAIL
    int o = W.occ[c];
    if (o == OCC_EMPTY) G.mv[i] = MV_OK;
    else { G.mv[i] = (unsigned)o; W.counters[CTR_CHANGED] = 1; }  // succeeds iff the occupant o succeeds
}

__global__ void __launch_bounds__(256) k_move_jump(WorldView W, const GroupDev *gtab) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    unsigned m = G.mv[i];
    if (m >= MV_OK) return;
    unsigned s = gtab[ref_group((int)m)].mv[ref_index((int)m)];
    G.mv[i] = s;                                                   // OK / FAIL resolve me; otherwise jump
    if (s < MV_OK) W.counters[CTR_CHANGED] = 1;
}

// collide bookkeeping for failed moves (Map.cc:334-353) + vacate the old cells of successful ones
__global__ void __launch_bounds__(256) k_move_apply1(WorldView W, const GroupDev *gtab) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int c = G.drank_a[i];
    if (c < 0) return;
    if (G.mv[i] == MV_OK) return;   // old cell is vacated in apply2's first half (after every reader of occ is done)
    int o = W.occ[c];
    int blocker;
    if (o == OCC_EMPTY) blocker = (int)(unsigned)W.claim[c];       // lost an empty cell to the lowest key
    else {
        const GroupDev O = gtab[ref_group(o)];
        int oi = ref_index(o);
        bool left_before = O.mv[oi] == MV_OK && O.key[oi] < G.key[i];
        blocker = left_before ? (int)(unsigned)W.claim[c] : o;
    }
    G.last_op[i] = OP_COLLIDE;
    G.op_obj[i] = blocker;
}

__global__ void __launch_bounds__(256) k_move_vacate(WorldView W) {
    const GroupDev G = W.grp[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n || G.drank_a[i] < 0 || G.mv[i] != MV_OK) return;
    W.occ[G.y[i] * W.w + G.x[i]] = OCC_EMPTY;
}

__global__ void __launch_bounds__(256) k_move_enter(WorldView W) {
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int c = G.drank_a[i];
    if (c < 0 || G.mv[i] != MV_OK) return;
    W.occ[c] = ref_pack(g, i);
    int ny = c / W.w;
    G.x[i] = c - ny * W.w; G.y[i] = ny;
}

// ------------------------------------------------------------------------------------------------ reward rules
// Event(a, op, b) with 'any' symbols: every agent i of group(a), in index order, whose last_op == op and whose
// op_obj is in group(b) triggers the rule once (RewardEngine.cc:373-414).  Receivers that are the subject are added
// by the subject's own thread; receivers that are the object are counted with an int atomic and replayed as `hits`
// sequential float adds of the same value -- order-independent, hence exact.
__global__ void __launch_bounds__(256) k_rule(WorldView W, RuleArgs A) {
    const GroupDev G = W.grp[A.ga];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool trig = false;
    if (i < G.n) {
        int o = G.op_obj[i];
        if (o >= 0 && ref_group(o) == A.gb && G.last_op[i] == A.op) {
            trig = true;
            if (A.n_subj) {
                float nr = G.next_reward[i];
                for (int k = 0; k < A.n_subj; k++) nr += A.v_subj[k];
                G.next_reward[i] = nr;
            }
            if (A.n_obj) atomicAdd(&W.grp[A.gb].hits[ref_index(o)], 1);
        }
    }
    if (__ballot(trig) && lane_id() == 0) W.counters[CTR_TRIGGER + A.rule_no] = 1;
}

__global__ void __launch_bounds__(256) k_rule_obj(WorldView W, RuleArgs A) {
    const GroupDev G = W.grp[A.gb];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    int h = G.hits[i];
    if (!h) return;
    float nr = G.next_reward[i];
    for (; h > 0; h--) for (int k = 0; k < A.n_obj; k++) nr += A.v_obj[k];
    G.next_reward[i] = nr;
    G.hits[i] = 0;
}

// end of step: pending actions are consumed
__global__ void __launch_bounds__(256) k_finish(WorldView W) {
    const GroupDev G = W.grp[blockIdx.y];
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) G.pend[i] = PEND_NONE;
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

__global__ void __launch_bounds__(SCAN_THREADS) k_compact_a(GroupDev G, int *sums) {
    int tot = block_count([&](int i) { return !G.dead[i]; }, G.n);
    if (threadIdx.x == 0) sums[blockIdx.x] = tot;
}

// stable compaction into the alternate buffers `D` + init_reward + re-index the map
__global__ void __launch_bounds__(SCAN_THREADS) k_compact_c(WorldView W, int g, GroupDev D, const int *sums) {
    const GroupDev G = W.grp[g];
    const float step_reward = W.type[g].step_reward;
    block_rank([&](int i) { return !G.dead[i]; },
               [&](int i, int r) {
                   int x = G.x[i], y = G.y[i];
                   D.x[r] = x; D.y[r] = y; D.id[r] = G.id[i]; D.hp[r] = G.hp[i]; D.last_action[r] = G.last_action[i];
                   D.last_reward[r] = G.next_reward[i];
                   D.next_reward[r] = step_reward;
                   W.occ[y * W.w + x] = ref_pack(g, r);
               },
               G.n, sums[blockIdx.x]);
}

// the non-double-buffered per-agent state of the survivors
__global__ void __launch_bounds__(256) k_compact_reset(GroupDev D, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    D.dead[i] = 0; D.last_op[i] = OP_NULL; D.op_obj[i] = -1; D.pend[i] = PEND_NONE;
}

// ================================================================================================ launchers
static inline dim3 grid_all(const WorldView &W, int threads) {
    int mx = 1;
    for (int g = 0; g < W.G; g++) mx = W.grp[g].n > mx ? W.grp[g].n : mx;
    return dim3((mx + threads - 1) / threads, W.G);
}

void launch_set_tables(hipStream_t s, const WorldView &W, GroupDev *gtab, TypeDev *ttab) {
    hipLaunchKernelGGL(k_set_tables, dim3(1), dim3(64), 0, s, W, gtab, ttab);
}

void launch_paint(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab) {
    int ncell = W.w * W.h;
    int blocks = (ncell + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_paint, dim3(blocks), dim3(256), 0, s, W, gtab, ttab);
}

void launch_minimap(hipStream_t s, const WorldView &W, const RenderArgs &R, int *counts) {
    int VHW = R.VH * R.VW;
    (void)hipMemsetAsync(counts, 0, sizeof(int) * W.G * VHW, s);
    int mx = 1;
    for (int g = 0; g < W.G; g++) mx = W.grp[g].n > mx ? W.grp[g].n : mx;
    int bx = (mx + 255) / 256;
    if (bx > 512) bx = 512;
    hipLaunchKernelGGL(k_minimap, dim3(bx, W.G), dim3(256), VHW * sizeof(int), s, W, R, counts);
}

size_t render_lds_bytes(const WorldView &W, const RenderArgs &R, int AG) {
    int VHW = R.VH * R.VW;
    return (size_t)AG * VHW * 8 + (size_t)W.G * VHW * 4 + (size_t)AG * 12 + (size_t)R.C * 4;
}

void launch_render(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4, bool nt) {
    if (R.n <= 0) return;
    int tiles = (R.n + P.AG - 1) / P.AG;
    size_t lds = render_lds_bytes(W, R, P.AG);
    if (vec4 && nt) hipLaunchKernelGGL((k_render<true, true>), dim3(tiles), dim3(256), lds, s, W, R, P);
    else if (vec4) hipLaunchKernelGGL((k_render<true, false>), dim3(tiles), dim3(256), lds, s, W, R, P);
    else hipLaunchKernelGGL((k_render<false, false>), dim3(tiles), dim3(256), lds, s, W, R, P);
}

void launch_set_action(hipStream_t s, const WorldView &W, int g, const int *actions, int call_base, int *sums) {
    int n = W.grp[g].n;
    if (n <= 0) return;
    int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_set_action_a, dim3(nb), dim3(SCAN_THREADS), 0, s, W, g, actions, call_base, sums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, sums, nb, W.counters + CTR_ATTACK);
    hipLaunchKernelGGL(k_set_action_c, dim3(nb), dim3(SCAN_THREADS), 0, s, W, g, actions, sums);
}

void launch_attack_rank(hipStream_t s, const WorldView &W, const int *rank) {
    hipLaunchKernelGGL(k_attack_rank, grid_all(W, 256), dim3(256), 0, s, W, rank);
}
void launch_attack_iter(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int use_b) {
    hipLaunchKernelGGL((k_attack_eval<false>), grid_all(W, ATT_THREADS), dim3(ATT_THREADS), 0, s, W, gtab, ttab, use_b);
}
void launch_attack_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int use_b) {
    hipLaunchKernelGGL((k_attack_eval<true>), grid_all(W, ATT_THREADS), dim3(ATT_THREADS), 0, s, W, gtab, ttab, use_b);
}
void launch_starve(hipStream_t s, const WorldView &W) { hipLaunchKernelGGL(k_starve, grid_all(W, 256), dim3(256), 0, s, W); }

void launch_move_prep(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    (void)hipMemsetAsync(W.claim, 0xFF, sizeof(unsigned long long) * (size_t)W.w * W.h, s);
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_move_prep, g, dim3(256), 0, s, W);
    hipLaunchKernelGGL(k_move_claim, g, dim3(256), 0, s, W, gtab);
    hipLaunchKernelGGL(k_move_init, g, dim3(256), 0, s, W);
}
void launch_move_jump(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    hipLaunchKernelGGL(k_move_jump, grid_all(W, 256), dim3(256), 0, s, W, gtab);
}
void launch_move_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab) {
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_move_apply1, g, dim3(256), 0, s, W, gtab);
    hipLaunchKernelGGL(k_move_vacate, g, dim3(256), 0, s, W);
    hipLaunchKernelGGL(k_move_enter, g, dim3(256), 0, s, W);
}

void launch_rule(hipStream_t s, const WorldView &W, const RuleArgs &A) {
    int na = W.grp[A.ga].n, nb = W.grp[A.gb].n;
    if (na > 0) hipLaunchKernelGGL(k_rule, dim3((na + 255) / 256), dim3(256), 0, s, W, A);
    if (A.n_obj && na > 0 && nb > 0) hipLaunchKernelGGL(k_rule_obj, dim3((nb + 255) / 256), dim3(256), 0, s, W, A);
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
void launch_compact(hipStream_t s, const WorldView &W, int g, const GroupDev &D, int new_n, int *sums) {
    int n = W.grp[g].n;
    if (n <= 0) return;
    int nb = (n + SCAN_TILE - 1) / SCAN_TILE;
    hipLaunchKernelGGL(k_compact_a, dim3(nb), dim3(SCAN_THREADS), 0, s, W.grp[g], sums);
    hipLaunchKernelGGL(k_scan_blocks, dim3(1), dim3(1024), 0, s, sums, nb, (int *)nullptr);
    hipLaunchKernelGGL(k_compact_c, dim3(nb), dim3(SCAN_THREADS), 0, s, W, g, D, sums);
    if (new_n > 0) hipLaunchKernelGGL(k_compact_reset, dim3((new_n + 255) / 256), dim3(256), 0, s, D, new_n);
}

}  // namespace magent_amd
