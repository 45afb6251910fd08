// kernels.hip -- hand-written HIP kernels of the grid-world step engine for gfx950 (CDNA4, wave64).
//
// No dense contraction on this path: MFMA is unused on purpose.  The rules that matter are coalesced SoA access,
// wave-private LDS strips that turn per-cell work into 1 KiB streaming stores, wave ballots for in-wave ranking,
// LDS / global integer atomics (histogram, hit bits, 64-bit umin move arbitration) and as few host round trips as
// possible.  Compiled with -ffp-contract=off; float ops keep the reference's order.
//
// Reference semantics each kernel restates (file:line into /root/reference/src/gridworld):
//   k_paint / k_minimap / k_render   GridWorld::get_observation GridWorld.cc:292-401, Map::extract_view Map.cc:129-207
//   k_set_action_*                   GridWorld::set_action GridWorld.cc:403-454
//   k_attack_*                       GridWorld::step attack loop GridWorld.cc:475-506, Map.cc:209-310, GridWorld.h:203-209
//   starve_body (in k_move*_prep)    GridWorld.cc:519-542, GridWorld.h:194-201
//   k_move_*                         GridWorld.cc:574-613, Map::do_move Map.cc:313-358
//   k_rule*                          GridWorld::calc_reward GridWorld.cc:681-692, RewardEngine.cc:216-443
//   k_clear_* / k_compact_solo       GridWorld::clear_dead GridWorld.cc:633-665, Agent::init_reward GridWorld.h:168-174
#include "engine.h"
#include "launch.h"
#include <algorithm>
#include <cstddef>

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

// writes `v` into every map cell of a bw x bl body whose top-left cell is (x, y) (Map::fill_area / clear_area)
__device__ __forceinline__ void body_fill(const WorldView &W, int x, int y, int bw, int bl, int v) {
    for (int by = 0; by < bl; by++)
        for (int bx = 0; bx < bw; bx++) W.occ[(y + by) * W.w + x + bx] = v;
}

// the offset of action payload `k` of table `off` as the agent (g, i) means it: given in the agent's frame, turned by the way
// it faces when turn_mode is on (Map.cc:209-226, GridWorld.cc:585-598)
__device__ __forceinline__ int2 agent_delta(const WorldView &W, const GroupDev &G, int i, int table_off, int k) {
    int2 d = W.delta[table_off + k];
    if (W.turn_mode) { int ax, ay; dir_rotate(G.dir[i], d.x, d.y, ax, ay); d = make_int2(ax, ay); }
    return d;
}

// the footprint of a body on the map: a body lying east-west is transposed (Map.cc:589-599)
__device__ __forceinline__ int2 dims_for_dir(const TypeDev &T, int dir) {
    return (dir == DIR_NORTH || dir == DIR_SOUTH) ? make_int2(T.bw, T.bl) : make_int2(T.bl, T.bw);
}
__device__ __forceinline__ int2 body_dims(const WorldView &W, const GroupDev &G, const TypeDev &T, int i) {
    return W.turn_mode ? dims_for_dir(T, G.dir[i]) : make_int2(T.bw, T.bl);
}
// the cell attack offset `k` of agent (g, i) points at (possibly outside the map): counted from the body's reference corner,
// in the agent's frame (Map::get_attack_obj, Map.cc:209-226)
__device__ __forceinline__ int2 attack_target(const WorldView &W, const GroupDev &G, const TypeDev &T, int i, int k) {
    const int2 d = W.delta[T.attack_off + k];
    if (!W.turn_mode) return make_int2(G.x[i] + d.x, G.y[i] + d.y);
    const int dir = G.dir[i];
    int rx, ry, ax, ay;
    saved_to_real(dir, T.bw, T.bl, G.x[i], G.y[i], rx, ry);
    dir_rotate(dir, d.x, d.y, ax, ay);
    return make_int2(rx + ax, ry + ay);
}

// Gates of the single-sync step (engine.hip: Env::step).  Fixed-point rounds are launched without waiting for the
// host; rounds after convergence find nothing to do, and everything after a phase whose rounds ran out returns at
// once so that the host can take over from exactly that state.  No gate kernel: the last round of a phase writes the
// phase's flag itself (`flag` = counter index, < 0 = do not report).
__device__ __forceinline__ bool attack_open(const WorldView &W) { return W.counters[CTR_OPEN_ATTACK] != 0; }
__device__ __forceinline__ bool step_open(const WorldView &W) { return (W.counters[CTR_OPEN_ATTACK] | W.counters[CTR_OPEN_MOVE]) != 0; }

// tests only (MAGENT_OPT_ATTACK_PAIRS=0 / MAGENT_OPT_MOVE_BATCHES=0): leave a phase open without running a round
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
// The end-of-step report of the multi-launch step, straight into pinned host memory (the host spins on `seq`: a stream
// synchronisation behind a device-to-host copy costs several times the PCIe write it waits for), and the per-step counters
// back to zero -- unless a phase was left open: then the host continues from exactly this state and resets afterwards.
__global__ void __launch_bounds__(64) k_step_report(int *counters, StepRecord *rec, int seq, int NG) {
    const int tid = threadIdx.x;
    const int oa = counters[CTR_OPEN_ATTACK], om = counters[CTR_OPEN_MOVE];
    const bool open = (oa | om) != 0;
    const bool trig = tid < CTR_TRIGGER_END - CTR_TRIGGER && counters[CTR_TRIGGER + tid] != 0;
    const unsigned long long mask = __ballot(trig);
    if (tid < NG) {
        int d = 0;
        for (int k = 0; k < DEAD_SLOTS; k++) d += counters[dead_slot(tid, k)];
        rec->dead[tid] = d; rec->taken[tid] = counters[CTR_TAKEN + tid];
    }
    if (tid == 0) {
        rec->triggers = mask;
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
    __threadfence_system();
    if (tid == 0) __hip_atomic_store((int *)&rec->seq, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
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

// ------------------------------------------------------------------------------------------------ paint
// viewcell[c] = {group | EMPTY | WALL, bits(hp / type.hp)}: one pass over the map, coalesced 4 B in / 8 B out.
// The division is the reference's `p->get_hp() / p->get_type().hp` (Map.cc:197), IEEE round-to-nearest.
// With at most 3 groups the record packs into ONE 32-bit word: hp / type.hp lies in [0, 1] (hp is capped at type.hp
// and agents with hp < 0 are off the map), so the two top bits of its float pattern are free for the group; EMPTY and
// WALL are the two all-ones-ish sentinels.  Half the footprint: the 1000 x 1000 map is 4 MB and lives in an XCD's L2.
constexpr unsigned VC_EMPTY = 0xFFFFFFFFu, VC_WALL = 0xFFFFFFFEu, VC_FOOD = 0xFFFFFFFDu;

template <bool PACKED>
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
        if (PACKED) {
            unsigned v = o == OCC_EMPTY ? VC_EMPTY : o == OCC_WALL ? VC_WALL : o == OCC_FOOD ? VC_FOOD : (((unsigned)rec.x << 30) | (unsigned)rec.y);
            if (o >= 0 && ((unsigned)rec.y >> 30)) W.counters[CTR_PACK_OVERFLOW] = 1;   // ratio outside [0, 2): never expected
            ((unsigned *)W.viewcell)[c] = v;
        } else {
            W.viewcell[c] = rec;
        }
    }
}

// one cell of the painted copy, in whichever format the game uses (must match k_paint)
__device__ __forceinline__ void vc_store(const WorldView &W, int c, int code, unsigned hpbits) {
    if (W.vc_packed) {
        ((unsigned *)W.viewcell)[c] = code == OCC_EMPTY ? VC_EMPTY : code == OCC_WALL ? VC_WALL : code == OCC_FOOD ? VC_FOOD : (((unsigned)code << 30) | hpbits);
        if (code >= 0 && (hpbits >> 30)) W.counters[CTR_PACK_OVERFLOW] = 1;
    } else W.viewcell[c] = make_int2(code, (int)hpbits);
}
// Map::clear_area; with live_paint the painted copy follows at once (the step keeps it current: cells are emptied where
// they are vacated, and at the end of the step every live agent paints its own body -- repaint_body)
__device__ __forceinline__ void cells_clear(const WorldView &W, int x, int y, int bw, int bl) {
    for (int by = 0; by < bl; by++)
        for (int bx = 0; bx < bw; bx++) {
            const int c = (y + by) * W.w + x + bx;
            W.occ[c] = OCC_EMPTY;
            if (W.live_paint) vc_store(W, c, OCC_EMPTY, 0u);
        }
}
__device__ __forceinline__ void repaint_body(const WorldView &W, const GroupDev &G, const TypeDev &T, int g, int i) {
    if (G.dead[i]) return;
    const unsigned bits = __float_as_uint(__fdiv_rn(G.hp[i], T.hp));   // the reference's `get_hp() / get_type().hp` (Map.cc:197)
    const int x = G.x[i], y = G.y[i];
    const int2 fp = body_dims(W, G, T, i);
    for (int by = 0; by < fp.y; by++)
        for (int bx = 0; bx < fp.x; bx++) vc_store(W, (y + by) * W.w + x + bx, g, bits);
}

// ------------------------------------------------------------------------------------------------ minimap histogram
// counts[j][cell] = number of agents of group j whose (x / scale_w, y / scale_h) is cell (GridWorld.cc:341-352;
// dead-but-not-cleared agents are counted, as in the reference).  LDS int atomics per block, then one global
// atomic per non-empty bin.  blockIdx.y = group.
// `skip`: the observing type is can_absorb -- absorbed agents are left out and counted in left_out[j], which the
// normalisation takes off the divisor (GridWorld.cc:343-347: the OBSERVING group's type decides).
__global__ void __launch_bounds__(256) k_minimap(WorldView W, RenderArgs R, int *counts, int *left_out, int skip) {
    extern __shared__ int s_hist[];
    const int VHW = R.VH * R.VW, j = blockIdx.y;
    const GroupDev G = W.grp[j];
    if ((int)(blockIdx.x * blockDim.x) >= G.n) return;
    for (int k = threadIdx.x; k < VHW; k += blockDim.x) s_hist[k] = 0;
    __syncthreads();
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < G.n; i += gridDim.x * blockDim.x) {
        if (skip && G.absorbed[i]) { atomicAdd(&left_out[j], 1); continue; }
        int cx = G.x[i] / R.scale_w, cy = G.y[i] / R.scale_h;
        atomicAdd(&s_hist[cy * R.VW + cx], 1);
    }
    __syncthreads();
    for (int k = threadIdx.x; k < VHW; k += blockDim.x)
        if (s_hist[k]) atomicAdd(&counts[j * VHW + k], s_hist[k]);
}

// ------------------------------------------------------------------------------------------------ minimap normalise
// mini[j][cell] = float(count) / float(total_j) exactly as the reference (GridWorld.cc:350,356): float ++ saturates
// at 2^24; an empty group divides 0 by 0 and the x86 default NaN the reference then holds is 0xFFC00000.
__global__ void __launch_bounds__(256) k_minimap_norm(RenderArgs R, int G, int *counts, float *mini, const int *left_out, int skip) {
    const int VHW = R.VH * R.VW;
    int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= G * VHW) return;
    int tot = R.totals[k / VHW] - (skip ? left_out[k / VHW] : 0);
    mini[k] = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(counts[k], 1 << 24), (float)(unsigned)tot);
    counts[k] = 0;   // the next histogram starts from zero (the buffer is zeroed when it is allocated)
}

// ------------------------------------------------------------------------------------------------ feature rows
// feature rows [id bits x E | one-hot last_action x NA | last_reward | x / w | y / h] (GridWorld.cc:386-396)
struct AgentFeat { int id, la; float lr, fx, fy; };

// what the observation kernels read of the world: scalars, the painted map, the observing group
struct RenderWorld {
    int w, h, G;
    const int2 *viewcell;
    const unsigned char *mask;
    GroupDev grp;
    TypeDev type;
};
__host__ __device__ __forceinline__ RenderWorld render_world(const WorldView &W, int g) {
    RenderWorld V;
    V.w = W.w; V.h = W.h; V.G = W.G; V.viewcell = W.viewcell; V.mask = W.mask; V.grp = W.grp[g]; V.type = W.type[g];
    return V;
}
__device__ __forceinline__ AgentFeat load_feat(const RenderWorld &W, const GroupDev &Gd, int i) {
    AgentFeat a;
    a.id = Gd.id[i]; a.la = Gd.last_action[i]; a.lr = Gd.last_reward[i];
    a.fx = __fdiv_rn((float)Gd.x[i], (float)W.w);
    a.fy = __fdiv_rn((float)Gd.y[i], (float)W.h);
    return a;
}

// value of feature slot f from registers (no loads, no divergent paths with memory behind them)
__device__ __forceinline__ float feature_value(const RenderArgs &R, const AgentFeat &a, int f) {
    const int rel = f - R.E;
    float v = (f < 31 && ((a.id >> f) & 1)) ? 1.0f : 0.0f;                 // id bits, LSB first
    v = f >= R.E ? (a.la == rel ? 1.0f : 0.0f) : v;                        // one-hot last action
    v = rel == R.NA ? a.lr : v;                                            // a fresh agent's last_action == NA lands here
    v = rel == R.NA + 1 ? a.fx : v;                                        // and is overwritten (GridWorld.cc:390-392)
    v = rel == R.NA + 2 ? a.fy : v;
    return v;
}

// the feature tensor of the group, as float4 where the pointer allows; `block` of `n_blocks` workgroups of 256 threads
template <bool VEC4>
__device__ __forceinline__ void features_body(const RenderWorld &W, const RenderArgs &R, const RenderPlan &P, unsigned block, unsigned n_blocks) {
    const GroupDev Gd = W.grp;
    const unsigned total = (unsigned)R.n * (unsigned)R.F;
    const unsigned nq = VEC4 ? total >> 2 : 0;
    for (unsigned q = block * 256u + threadIdx.x; q < nq; q += n_blocks * 256u) {
        const unsigned k = q << 2;
        const int i = fdiv_u32(k, P.div_f);
        int f = k - i * R.F;
        // four consecutive floats touch at most two agents when F >= 4 (the feature row always holds >= 2 slots, so
        // the general case walks on); both agents' fields are loaded up front so the loads overlap
        AgentFeat a0 = load_feat(W, Gd, i), a1 = load_feat(W, Gd, min(i + 1, R.n - 1));
        int cur = i;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; e++) {
            if (cur > i + 1) { a1 = load_feat(W, Gd, cur); }               // only when F < 3 (never in practice)
            v[e] = feature_value(R, cur == i ? a0 : a1, f);
            if (++f == R.F) { f = 0; ++cur; }
        }
        v4f f4 = {v[0], v[1], v[2], v[3]};
        __builtin_nontemporal_store(f4, (v4f *)R.feat + q);
    }
    for (unsigned k = (nq << 2) + block * 256u + threadIdx.x; k < total; k += n_blocks * 256u) {
        const int i = fdiv_u32(k, P.div_f);
        R.feat[k] = feature_value(R, load_feat(W, Gd, i), k - i * R.F);
    }
}

// stand-alone launch, used when the feature pointer is not 16-byte aligned while the view pointer is (or vice versa)
template <bool VEC4>
__global__ void __launch_bounds__(256) k_features(WorldView W, RenderArgs R, RenderPlan P) {
    features_body<VEC4>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}

// ------------------------------------------------------------------------------------------------ observation render
// The view tensor of a group is one contiguous array of n * VH * VW cells x C floats.  The kernel walks it as a flat
// sequence of window cells, 64 cells (one per lane) per wave step:
//   load   : the lane's cell of `viewcell` (ONE 8-byte load, masked by the view range and the map bounds) and the
//            G minimap floats of its window position; agent x / y are wave-broadcast loads (a wave spans <= 2 agents)
//   expand : a wave-uniform loop over the C channels (descriptor = scalar load) turns the cell into its C floats --
//            no lane diverges on the channel kind -- written to a wave-private LDS strip of 64 * C floats
//   store  : the strip is read back as float4 and streamed out: 256 * C bytes per step, contiguous, starting on a
//            128-byte line (256 * C is a multiple of 128), 1 KiB per global_store_dwordx4 wave instruction.
// No workgroup barrier, 1.75 KiB of LDS per wave: occupancy is bounded by the 32 waves / CU limit, not by LDS.
// Every output element is written exactly once, zeros included (the reference's memset, GridWorld.cc:310, is fused
// into the stores): the output is the algorithmic traffic, 4 * VH * VW * C bytes per agent.
// Workgroups own contiguous spans of the cell sequence, and the span index is XCD-aware (blockIdx b runs on XCD
// b % 8): each XCD walks one contiguous eighth of the agents, so spatially ordered groups keep its part of the map
// in its own L2.
constexpr int RENDER_WAVES = 4;

// (bx of nb workgroups of the launch work on this group: the render spans first, then the feature rows)
// (TURN: turn_mode -- the window is laid out in the agent's frame; a template parameter so that the ordinary kernel carries none of it)
// (CELLS16: the policy kernels' input format -- every window cell one 16-byte vector of 8 bf16: the C channels rounded to nearest
// even, zeros, and 1.0 in channel 7 (conv1's bias rides on it, magent_amd/csrc/policy.hip).  2.7 KB per agent instead of 4.7, a
// lane stores its own cell: no hand-over between lanes)
typedef __attribute__((ext_vector_type(8))) __bf16 cell16_t;
template <bool VEC4, bool NT, int U, bool PACKED, bool TURN, bool CELLS16 = false>
__device__ __forceinline__ void render_block(const RenderWorld &W, const RenderArgs &R, const RenderPlan &P, int bx, int nb) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VHW = R.VH * R.VW, C = R.C, G = W.G;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *strip = (float *)smem + (size_t)wave * P.strip_floats;      // [64][C], wave-private

    if (bx >= P.spans) {   // the trailing workgroups write the group's feature rows (3 % of the bytes)
        features_body<VEC4>(W, R, P, bx - P.spans, nb - P.spans);
        return;
    }
    int span = bx;
    if (P.xcd_chunk > 0 && span < P.xcd_chunk * 8) span = (span & 7) * P.xcd_chunk + (span >> 3);
    const GroupDev Gd = W.grp;
    const TypeDev T = W.type;
    const unsigned char *mask = W.mask + T.mask_off;
    const unsigned total_cells = (unsigned)R.n * (unsigned)VHW;
    const size_t total_floats = (size_t)total_cells * C;
    const int q_per_step = 16 * C;                                     // float4 per 64-cell step

    // U consecutive steps per wave iteration: all their loads are issued before the first expansion, so a wave
    // keeps U independent (x/y -> viewcell) load chains in flight
    for (int it = wave * U; it < P.steps_per_span; it += RENDER_WAVES * U) {
        const unsigned step0 = (unsigned)span * P.steps_per_span + it;
        if (step0 * 64u >= total_cells) break;
        int cellv[U], xv[U], yv[U], dirv[U];
        bool valid[U];
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned k = (step0 + u) * 64u + lane;
            valid[u] = k < total_cells;
            const int a = valid[u] ? (int)fdiv_u32(k, P.div_vhw) : 0;
            cellv[u] = valid[u] ? (int)(k - a * VHW) : 0;
            xv[u] = Gd.x[a]; yv[u] = Gd.y[a];
            dirv[u] = TURN ? Gd.dir[a] : DIR_NORTH;
        }
        int2 recv[U];
        float miniv[U][MAXG];   // minimap value of this window position for channel block b (group (g + b) % G)
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int vy = fdiv_u32(cellv[u], P.div_vw);
            const int vx = cellv[u] - vy * R.VW;
            int ox = T.view_x1 + vx, oy = T.view_y1 + vy;            // window cell -> offset in the agent's frame ...
            int bx = xv[u], by = yv[u];                               // ... counted from the body's reference corner
            if (TURN) {                                               // ... -> offset on the map (Map.cc:129-207)
                dir_rotate(dirv[u], ox, oy, ox, oy);
                saved_to_real(dirv[u], T.bw, T.bl, xv[u], yv[u], bx, by);
            }
            const int mx = bx + ox, my = by + oy;
            const bool in = valid[u] && mask[cellv[u]] && mx >= 0 && mx < W.w && my >= 0 && my < W.h;
            if (PACKED) {
                const unsigned v = in ? ((const unsigned *)W.viewcell)[my * W.w + mx] : VC_EMPTY;
                recv[u] = v >= VC_FOOD ? make_int2(v == VC_WALL ? OCC_WALL : v == VC_FOOD ? OCC_FOOD : OCC_EMPTY, 0)
                                       : make_int2((int)(v >> 30), (int)(v & 0x3FFFFFFFu));
            } else {
                recv[u] = in ? W.viewcell[my * W.w + mx] : make_int2(OCC_EMPTY, 0);
            }
            if (R.minimap) {
                int j = R.g;
#pragma unroll
                for (int b = 0; b < MAXG; b++)
                    if (b < G) { miniv[u][b] = R.mini[j * VHW + cellv[u]]; j = (j + 1 == G) ? 0 : j + 1; }
            }
        }
#pragma unroll
        for (int u = 0; u < U; u++) {
            const unsigned k0 = (step0 + u) * 64u;
            if (k0 >= total_cells) break;
            const int cell = cellv[u];
            const int code = recv[u].x & 0xff;
            const float hp = __int_as_float(recv[u].y);
            bool self = false;
            if (R.minimap) self = cell == (int)(fdiv_u32(yv[u], P.div_scale_h) * R.VW + fdiv_u32(xv[u], P.div_scale_w));
            // ---- expand: channel layout [wall | (has, hp[, minimap]) of group (g + b) % G for b = 0..G-1]
            // (GridWorld.cc:897-913); the loop is over wave-uniform values only -- no lane diverges, nothing is loaded
            float *dst = strip + lane * C;
            dst[0] = code == (OCC_WALL & 0xff) ? 1.0f : 0.0f;
            if (R.food) dst[1] = code == (OCC_FOOD & 0xff) ? 1.0f : 0.0f;   // food has a presence channel only (Map.cc:190-196)
            {
                int j = R.g;
                const int stride = R.minimap ? 3 : 2;
                float *blocks = dst + 1 + R.food;
#pragma unroll
                for (int b = 0; b < MAXG; b++)
                    if (b < G) {
                        const bool m = code == j;
                        float *d = blocks + b * stride;
                        d[0] = m ? 1.0f : 0.0f;
                        d[1] = m ? hp : 0.0f;
                        if (R.minimap) {
                            float v = miniv[u][b];
                            if (self && v == v) v += 1.0f;   // NaN stays the same NaN (x86 propagates the operand)
                            d[2] = v;                        // unmasked copy + self marker (GridWorld.cc:374-383)
                        }
                        j = (j + 1 == G) ? 0 : j + 1;
                    }
            }
            if (CELLS16) {
                cell16_t v;
#pragma unroll
                for (int e = 0; e < 7; e++) v[e] = (__bf16)(e < C ? dst[e] : 0.0f);      // (the lane's own strip entries, just written)
                v[7] = (__bf16)1.0f;
                if (valid[u]) __builtin_nontemporal_store(v, (cell16_t *)R.view + (k0 + lane));
                continue;
            }
            // wave-private LDS hand-over between lanes: LDS ops of one wave execute in order; the fences keep the
            // compiler from moving accesses across the hand-over
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // ---- store
            const size_t f0 = (size_t)k0 * C;                          // first float of this step
            const size_t remain = total_floats - f0;
            if (VEC4) {
                const int nq = remain >= (size_t)(64 * C) ? q_per_step : (int)(remain >> 2);
                v4f *out4 = (v4f *)(R.view + f0);
                const v4f *src4 = (const v4f *)strip;
                for (int q = lane; q < nq; q += 64) {
                    v4f f4 = src4[q];
                    if (NT) __builtin_nontemporal_store(f4, out4 + q);
                    else out4[q] = f4;
                }
                if (remain < (size_t)(64 * C))                         // < 4 trailing floats of the whole tensor
                    for (int e = (nq << 2) + lane; e < (int)remain; e += 64) R.view[f0 + e] = strip[e];
            } else {
                const int ne = remain >= (size_t)(64 * C) ? 64 * C : (int)remain;
                for (int e = lane; e < ne; e += 64) R.view[f0 + e] = strip[e];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // the strip is reused by the next step
            __builtin_amdgcn_wave_barrier();
        }
    }
}

template <bool VEC4, bool NT, int U, bool PACKED, bool TURN>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render(WorldView W, RenderArgs R, RenderPlan P) {
    render_block<VEC4, NT, U, PACKED, TURN>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}
template <bool PACKED, bool TURN>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_cells16(WorldView W, RenderArgs R, RenderPlan P) {
    render_block<true, true, 1, PACKED, TURN, true>(render_world(W, R.g), R, P, blockIdx.x, gridDim.x);
}
// ---- the battle-shaped observation, software-pipelined (round 3).
// Two groups, minimap channels, 7 channels, packed view cells, no turn_mode, 16-byte aligned output: BASELINE's battle / the
// bench workload.  Same flat cell sequence, same strips, same stores as render_block; what changes is everything in front of them:
//   * what depends only on the window position (offset in the map, view-range mask, the two minimap values) is a 16-byte LDS
//     table entry, made once per workgroup;
//   * what depends only on the agent (x, y, its own minimap cell) is an LDS table of the <= 14 agents a block of 32 steps spans,
//     refilled every 32 steps -- no global load and no division chain between a step's index and its view-cell address;
//   * (agent, cell) advance incrementally from step to step (a wave's steps are 256 cells apart);
//   * the view-cell load of the wave's NEXT step is issued before the current step is expanded and stored, so the only global
//     latency of a step is hidden behind the previous step's LDS hand-over and stores;
//   * the channel expansion is straight-line code for the one layout [wall | has, hp, minimap | has, hp, minimap].
struct RenderFastPos { int dxy; float m0, m1; int mask; };
constexpr int RF_BLOCK_STEPS = 32;
__host__ __device__ inline int render_fast_agents(int VHW) { return RF_BLOCK_STEPS * 64 / VHW + 2; }

template <bool CELLS16>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_fast(RenderWorld W, RenderArgs R, RenderPlan P) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int VHW = R.VH * R.VW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= P.spans) {   // the trailing workgroups write the group's feature rows
        features_body<true>(W, R, P, blockIdx.x - P.spans, gridDim.x - P.spans);
        return;
    }
    float *strip = (float *)smem + (size_t)wave * (64 * 7);
    RenderFastPos *wtab = (RenderFastPos *)((float *)smem + RENDER_WAVES * 64 * 7);
    int4 *atab = (int4 *)(wtab + VHW);
    int span = blockIdx.x;
    if (P.xcd_chunk > 0 && span < P.xcd_chunk * 8) span = (span & 7) * P.xcd_chunk + (span >> 3);
    const GroupDev Gd = W.grp;
    const TypeDev T = W.type;
    const unsigned total_cells = (unsigned)R.n * (unsigned)VHW;
    const size_t total_floats = (size_t)total_cells * 7;
    const unsigned *vc = (const unsigned *)W.viewcell;
    const unsigned g = (unsigned)R.g;

    for (int c = threadIdx.x; c < VHW; c += 64 * RENDER_WAVES) {
        const int vy = fdiv_u32(c, P.div_vw), vx = c - vy * R.VW;
        RenderFastPos e;
        e.dxy = ((T.view_y1 + vy) << 16) | ((T.view_x1 + vx) & 0xFFFF);
        e.m0 = R.mini[(int)g * VHW + c];
        e.m1 = R.mini[(1 - (int)g) * VHW + c];
        e.mask = W.mask[T.mask_off + c];
        wtab[c] = e;
    }
    const int jump_a = (64 * RENDER_WAVES) / VHW, jump_c = (64 * RENDER_WAVES) - jump_a * VHW;   // a wave's next step is 256 cells on

    for (int b0 = 0; b0 < P.steps_per_span; b0 += RF_BLOCK_STEPS) {
        const unsigned step_b0 = (unsigned)span * P.steps_per_span + b0;
        const unsigned k_b0 = step_b0 * 64u;
        if (k_b0 >= total_cells) break;
        const int nsteps = min(RF_BLOCK_STEPS, P.steps_per_span - b0);
        const unsigned k_end = min(k_b0 + (unsigned)nsteps * 64u, total_cells);
        const int a0 = fdiv_u32(k_b0, P.div_vhw), a1 = fdiv_u32(k_end - 1u, P.div_vhw);
        __syncthreads();                                   // (the previous block's readers of atab are done)
        for (int i = threadIdx.x; i <= a1 - a0; i += 64 * RENDER_WAVES) {
            const int x = Gd.x[a0 + i], y = Gd.y[a0 + i];
            atab[i] = make_int4(x, y, (int)(fdiv_u32(y, P.div_scale_h) * R.VW + fdiv_u32(x, P.div_scale_w)), 0);
        }
        __syncthreads();

        // ---- the wave's first step of this block: index by division, view cell requested
        int it = wave;
        unsigned k = (step_b0 + it) * 64u + lane;
        int a = fdiv_u32(min(k, total_cells - 1u), P.div_vhw);
        int cell = (int)(min(k, total_cells - 1u) - (unsigned)a * VHW);
        RenderFastPos wt = wtab[cell];
        int4 at = atab[a - a0];
        unsigned v = VC_EMPTY;
        {
            const int mx = at.x + ((wt.dxy << 16) >> 16), my = at.y + (wt.dxy >> 16);
            if (it < nsteps && k < total_cells && wt.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h) v = vc[my * W.w + mx];
        }
        for (; it < nsteps; it += RENDER_WAVES) {
            const unsigned k0 = (step_b0 + it) * 64u;
            if (k0 >= total_cells) break;
            // ---- next step: indices and the view-cell request (in flight while this step is expanded and stored)
            int a_n = a + jump_a, cell_n = cell + jump_c;
            if (cell_n >= VHW) { cell_n -= VHW; a_n++; }
            const unsigned k_n = k + 64u * RENDER_WAVES;
            const bool more = it + RENDER_WAVES < nsteps && k_n < total_cells;
            RenderFastPos wt_n = wt;
            int4 at_n = at;
            unsigned v_n = VC_EMPTY;
            if (more) {
                wt_n = wtab[cell_n];
                at_n = atab[a_n - a0];
                const int mx = at_n.x + ((wt_n.dxy << 16) >> 16), my = at_n.y + (wt_n.dxy >> 16);
                if (wt_n.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h) v_n = vc[my * W.w + mx];
            }
            // ---- expand this step's cell: [wall | has, hp, minimap of the observing group | has, hp, minimap of the other]
            const unsigned top = v >> 30;
            const float hp = __uint_as_float(v & 0x3FFFFFFFu);
            const bool mine = top == g, theirs = top == 1u - g;
            float m0 = wt.m0, m1 = wt.m1;
            if (cell == at.z) { if (m0 == m0) m0 += 1.0f; if (m1 == m1) m1 += 1.0f; }   // self marker; NaN stays the same NaN
            const float c0 = v == VC_WALL ? 1.0f : 0.0f, c1 = mine ? 1.0f : 0.0f, c2 = mine ? hp : 0.0f;
            const float c4 = theirs ? 1.0f : 0.0f, c5 = theirs ? hp : 0.0f;
            if (CELLS16) {
                cell16_t o;
                o[0] = (__bf16)c0; o[1] = (__bf16)c1; o[2] = (__bf16)c2; o[3] = (__bf16)m0; o[4] = (__bf16)c4; o[5] = (__bf16)c5; o[6] = (__bf16)m1;
                o[7] = (__bf16)1.0f;
                if (k < total_cells) __builtin_nontemporal_store(o, (cell16_t *)R.view + k);
            } else {
                float *dst = strip + lane * 7;
                dst[0] = c0; dst[1] = c1; dst[2] = c2; dst[3] = m0; dst[4] = c4; dst[5] = c5; dst[6] = m1;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t f0 = (size_t)k0 * 7;
                const size_t remain = total_floats - f0;
                const int nq = remain >= (size_t)(64 * 7) ? 16 * 7 : (int)(remain >> 2);
                v4f *out4 = (v4f *)(R.view + f0);
                const v4f *src4 = (const v4f *)strip;
                for (int q = lane; q < nq; q += 64) __builtin_nontemporal_store(src4[q], out4 + q);
                if (remain < (size_t)(64 * 7))
                    for (int e = (nq << 2) + lane; e < (int)remain; e += 64) R.view[f0 + e] = strip[e];
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            a = a_n; cell = cell_n; k = k_n; wt = wt_n; at = at_n; v = v_n;
        }
    }
}
// ---- the battle-shaped float32 observation at scale: ONE four-wave workgroup per CU, the whole launch sweeping the output together.
// Measured on MI355X (profiles/r03_render_experiments.md): HBM takes stores fastest when few waves per CU write, in lock step, into
// one narrow moving window -- a device memset's geometry -- and worst from 20-32 independent waves per CU writing 56 KB apart,
// which is what k_render / k_render_fast need to cover their latencies.  Here the launch is 256 persistent workgroups; round r of
// workgroup b is the 4 x SU consecutive steps starting at (r * 256 + b) * 4 * SU, so everything in flight lies within ~3.6 MB.
// What lets four waves per CU keep up: every load is unconditional (clamped address, result selected) and requested DV rounds
// ahead -- x / y one round further -- in a ring of register slots that is never copied (a copy of a register with a load in
// flight waits for the load); the only branches are wave-uniform; a wave carries SU steps through SU LDS strips at once so that
// their ds_write -> ds_read -> store round trips overlap.  (SU = 2, DV = 2 measured best; SU = 3 / 4 and two workgroups per CU lose.)
template <bool CELLS16, int DV, int SU>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_sweep2(RenderWorld W, RenderArgs R, RenderPlan P, int sweep) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int N = DV + 2;                         // ring slots: rounds r .. r + DV + 1
    const int VHW = R.VH * R.VW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((int)blockIdx.x >= sweep) {     // trailing workgroups: the feature rows (measured: better here than at the end of the sweeping
        features_body<true>(W, R, P, blockIdx.x - sweep, gridDim.x - sweep);   // workgroups, where four waves per CU crawl through them)
        return;
    }
    float *strips = (float *)smem + (size_t)wave * (SU * 64 * 7);
    RenderFastPos *wtab = (RenderFastPos *)((float *)smem + RENDER_WAVES * SU * 64 * 7);
    const GroupDev Gd = W.grp;
    const TypeDev T = W.type;
    const unsigned total_cells = (unsigned)R.n * (unsigned)VHW;
    const unsigned total_steps = (total_cells + 63u) / 64u;
    const size_t total_floats = (size_t)total_cells * 7;
    const unsigned *vc = (const unsigned *)W.viewcell;
    const unsigned g = (unsigned)R.g;
    const unsigned ncell_map = (unsigned)W.w * (unsigned)W.h;
    for (int c = threadIdx.x; c < VHW; c += 64 * RENDER_WAVES) {
        const int vy = fdiv_u32(c, P.div_vw), vx = c - vy * R.VW;
        RenderFastPos e;
        e.dxy = ((T.view_y1 + vy) << 16) | ((T.view_x1 + vx) & 0xFFFF);
        e.m0 = R.mini[(int)g * VHW + c];
        e.m1 = R.mini[(1 - (int)g) * VHW + c];
        e.mask = W.mask[T.mask_off + c];
        wtab[c] = e;
    }
    __syncthreads();
    // ring state per slot and step: agent, window cell, x, y (requested DV + 1 rounds ahead), view cell (DV rounds ahead)
    int ia[N][SU], ic[N][SU], x[N][SU], y[N][SU];
    unsigned v[N][SU], in[N][SU];
    // (P.xcd_chunk < 0, tuning: workgroup b -- which runs on XCD b % 8 -- takes slot (b % 8) * (sweep / 8) + b / 8 of the round, so that an XCD's
    // workgroups write one contiguous eighth of the window)
    const unsigned slot = (P.xcd_chunk < 0 && (sweep & 7) == 0) ? (blockIdx.x & 7u) * ((unsigned)sweep >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    auto first_step = [&](unsigned round) { return ((round * (unsigned)sweep + slot) * RENDER_WAVES + wave) * SU; };
    auto index = [&](unsigned round, int slot) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const unsigned kk = min((first_step(round) + u) * 64u + lane, total_cells - 1u);
            ia[slot][u] = (int)fdiv_u32(kk, P.div_vhw);
            ic[slot][u] = (int)(kk - (unsigned)ia[slot][u] * VHW);
            x[slot][u] = Gd.x[ia[slot][u]]; y[slot][u] = Gd.y[ia[slot][u]];
        }
    };
    auto request = [&](int slot) {
#pragma unroll
        for (int u = 0; u < SU; u++) {
            const RenderFastPos wt = wtab[ic[slot][u]];
            const int mx = x[slot][u] + ((wt.dxy << 16) >> 16), my = y[slot][u] + (wt.dxy >> 16);
            const bool inside = wt.mask && (unsigned)mx < (unsigned)W.w && (unsigned)my < (unsigned)W.h;
            in[slot][u] = inside ? 1u : 0u;
            v[slot][u] = vc[min((unsigned)(my * W.w + mx), ncell_map - 1u)];     // always in bounds; dropped below when outside
        }
    };
#pragma unroll
    for (int j = 0; j < N; j++) index(j, j);
#pragma unroll
    for (int j = 0; j < DV; j++) request(j);
    bool running = true;
    for (unsigned round = 0; running; round += N) {
#pragma unroll
        for (int s = 0; s < N; s++) {
            if (!running) break;
            const unsigned step0 = first_step(round + s);
            if (step0 >= total_steps) { running = false; break; }
            request((s + DV) % N);
            float cs[SU][7];
#pragma unroll
            for (int u = 0; u < SU; u++) {
                const int cell = ic[s][u];
                const RenderFastPos wt = wtab[cell];
                const unsigned v0 = in[s][u] ? v[s][u] : VC_EMPTY;
                const unsigned top = v0 >> 30;
                const float hp = __uint_as_float(v0 & 0x3FFFFFFFu);
                const bool mine = top == g, theirs = top == 1u - g;
                const bool self = cell == (int)(fdiv_u32(y[s][u], P.div_scale_h) * R.VW + fdiv_u32(x[s][u], P.div_scale_w));
                const float m0 = (self && wt.m0 == wt.m0) ? wt.m0 + 1.0f : wt.m0;
                const float m1 = (self && wt.m1 == wt.m1) ? wt.m1 + 1.0f : wt.m1;
                cs[u][0] = v0 == VC_WALL ? 1.0f : 0.0f; cs[u][1] = mine ? 1.0f : 0.0f; cs[u][2] = mine ? hp : 0.0f; cs[u][3] = m0;
                cs[u][4] = theirs ? 1.0f : 0.0f; cs[u][5] = theirs ? hp : 0.0f; cs[u][6] = m1;
            }
            const unsigned k_grp = step0 * 64u;
            if (CELLS16) {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    const unsigned k = k_grp + 64u * u + lane;
                    cell16_t o;
#pragma unroll
                    for (int e = 0; e < 7; e++) o[e] = (__bf16)cs[u][e];
                    o[7] = (__bf16)1.0f;
                    if (k < total_cells) __builtin_nontemporal_store(o, (cell16_t *)R.view + k);
                }
            } else {
#pragma unroll
                for (int u = 0; u < SU; u++) {
                    float *dst = strips + u * (64 * 7) + lane * 7;
#pragma unroll
                    for (int e = 0; e < 7; e++) dst[e] = cs[u][e];
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const size_t f_grp = (size_t)k_grp * 7;
                if (total_floats - f_grp >= (size_t)(SU * 64 * 7)) {
                    v4f q0[SU], q1[SU];
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        const v4f *src4 = (const v4f *)(strips + u * (64 * 7));
                        q0[u] = src4[lane];
                        q1[u] = src4[lane + (lane < 48 ? 64 : 0)];   // (lanes 48..63 re-read a vector they do not store: no branch around the read)
                    }
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        v4f *out4 = (v4f *)(R.view + f_grp + (size_t)u * (64 * 7));
                        __builtin_nontemporal_store(q0[u], out4 + lane);
                        if (lane < 48) __builtin_nontemporal_store(q1[u], out4 + lane + 64);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < SU; u++) {
                        const size_t f0 = f_grp + (size_t)u * (64 * 7);
                        if (f0 >= total_floats) break;
                        const size_t remain = total_floats - f0;
                        const float *strip_u = strips + u * (64 * 7);
                        const int nq = remain >= (size_t)(64 * 7) ? 16 * 7 : (int)(remain >> 2);
                        for (int q = lane; q < nq; q += 64) __builtin_nontemporal_store(((const v4f *)strip_u)[q], (v4f *)(R.view + f0) + q);
                        if (remain < (size_t)(64 * 7))
                            for (int e = (nq << 2) + lane; e < (int)remain; e += 64) R.view[f0 + e] = strip_u[e];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
            index(round + s + N, s);                   // the slot is free: round r + N moves in
        }
    }
}

__host__ __device__ inline size_t render_fast_lds(int VHW) {
    return (size_t)RENDER_WAVES * 64 * 7 * sizeof(float) + (size_t)VHW * sizeof(RenderFastPos) + (size_t)render_fast_agents(VHW) * sizeof(int4);
}
// the shapes k_render_fast takes
static bool render_fast_ok(const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4) {
    const int VHW = R.VH * R.VW;
    return vec4 && W.G == 2 && R.minimap && !R.food && R.C == 7 && W.vc_packed && !R.turn && VHW >= 16 && VHW <= 1024 &&
           render_fast_lds(VHW) <= 48 * 1024;
}

// several groups of a small world in one launch (blockIdx.y = slot): small worlds are bound by the number of launches
template <bool PACKED>
__global__ void __launch_bounds__(64 * RENDER_WAVES) k_render_multi(WorldView W, RenderMulti M) {
    const int k = blockIdx.y;
    if ((int)blockIdx.x >= M.blocks[k]) return;
    if (W.turn_mode) render_block<true, true, 1, PACKED, true>(render_world(W, M.R[k].g), M.R[k], M.P[k], blockIdx.x, M.blocks[k]);
    else render_block<true, true, 1, PACKED, false>(render_world(W, M.R[k].g), M.R[k], M.P[k], blockIdx.x, M.blocks[k]);
}

// ------------------------------------------------------------------------------------------------ block scan trio
// Exclusive prefix sum of a per-agent predicate over one group, SCAN_ITEMS elements per thread:
//   pass A  per-block totals            pass B  one block scans the totals (+ base)      pass C  per-element ranks
// In-wave ranks come from ballots (wave64), cross-wave from LDS.
constexpr int SCAN_THREADS = 256, SCAN_ITEMS = SCAN_ITEMS_HOST, SCAN_TILE = SCAN_THREADS * SCAN_ITEMS;
static_assert(SCAN_TILE == SCAN_TILE_HOST, "scan tile");

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

// calls emit(i, exclusive_rank) for every i in this block's tile with pred(i); returns the tile's total.
// (all SCAN_ITEMS predicates are evaluated first -- their loads are in flight together -- and the waves meet once: the earlier
// form, one item at a time with two barriers each, made every launch that used it a chain of 8 dependent round trips)
template <class Pred, class Emit>
__device__ __forceinline__ int block_rank(Pred pred, Emit emit, int n, int block_offset) {
    __shared__ int s_w[SCAN_ITEMS][SCAN_THREADS / 64];
    const int base = blockIdx.x * SCAN_TILE;
    const int wave = threadIdx.x >> 6;
    bool p[SCAN_ITEMS];
    int r[SCAN_ITEMS];
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        const int i = base + k * SCAN_THREADS + threadIdx.x;
        p[k] = i < n && pred(i);
    }
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int wtot;
        r[k] = wave_rank(p[k], wtot);
        if (lane_id() == 0) s_w[k][wave] = wtot;
    }
    __syncthreads();
    int run = block_offset;
#pragma unroll
    for (int k = 0; k < SCAN_ITEMS; k++) {
        int before = 0, all = 0;
#pragma unroll
        for (int v = 0; v < SCAN_THREADS / 64; v++) { const int t = s_w[k][v]; all += t; if (v < wave) before += t; }
        if (p[k]) emit(base + k * SCAN_THREADS + threadIdx.x, run + before + r[k]);
        run += all;
    }
    __syncthreads();       // (s_w may be written again by the caller's next use)
    return run - block_offset;
}

// One-workgroup form for small groups (n <= SOLO_MAX): a single 1024-thread workgroup walks the group in tiles and
// carries the running rank itself -- one launch instead of three when the whole job is launch-latency bound.
constexpr int SOLO_THREADS = 1024, SOLO_MAX = 32768;
// (MAGENT_SCAN_SOLO_MAX: tests lower it so that small worlds run the multi-block scans of the large ones)
static int scan_solo_max() {
    static const int v = std::getenv("MAGENT_SCAN_SOLO_MAX") ? std::max(0, std::min(SOLO_MAX, std::atoi(std::getenv("MAGENT_SCAN_SOLO_MAX")))) : SOLO_MAX;
    return v;
}

template <class Pred, class Emit>
__device__ __forceinline__ int solo_rank(Pred pred, Emit emit, int n, int base) {
    __shared__ int s_w[SOLO_THREADS / 64];
    const int wave = threadIdx.x >> 6;
    int run = base;
    for (int t0 = 0; t0 < n; t0 += SOLO_THREADS) {
        const int i = t0 + threadIdx.x;
        const bool p = i < n && pred(i);
        int wtot, r = wave_rank(p, wtot);
        if (lane_id() == 0) s_w[wave] = wtot;
        __syncthreads();
        int before = 0, all = 0;
        for (int v = 0; v < SOLO_THREADS / 64; v++) { int t = s_w[v]; all += t; if (v < wave) before += t; }
        if (p) emit(i, run + before + r);
        run += all;
        __syncthreads();
    }
    return run;
}

// pass B, folded into pass C: every block adds up the totals of the blocks before it (a few hundred ints from L2 at a
// million agents) -- one dependent launch less per scan than a separate scan of the block totals
__device__ __forceinline__ int block_prefix(const int *sums, int b) {
    __shared__ int s_p[16];
    int t = 0;
    for (int k = threadIdx.x; k < b; k += blockDim.x) t += sums[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) t += __shfl_down(t, d);
    if (lane_id() == 0) s_p[threadIdx.x >> 6] = t;
    __syncthreads();
    int tot = 0;
    for (int w = 0; w < (int)(blockDim.x >> 6); w++) tot += s_p[w];
    __syncthreads();
    return tot;
}

// ------------------------------------------------------------------------------------------------ set_action
// Classifies the action (move | turn | attack) and computes the agent's order key.  Agent::last_action (an input of the
// feature rows only) is NOT written here: `pend` holds the action until the step's first per-agent pass (starve_body) stores
// it, so that set_action -- and the attack resolution behind it -- may run on a side stream while the observation of another
// group is still being rendered from last_action (engine.hip: Env::side_stream).  An observation asked for between
// set_action and step gets it through k_commit_action first.
//   move  : key = (boundary << 31) | insertion index.  Reference: moves run stripe lists 0..S-1 then the boundary
//           list, each in insertion order (GridWorld.cc:605-613); interior moves of different stripes cannot
//           interact (margin 4 > max speed 3), so only "boundary after interior" + insertion order is observable.
//   attack: key = running sequence number in the attack list (the shuffle permutes these).  It is NOT assigned here: this launch
//           leaves, per tile of SCAN_TILE agents, the tile's attack count (`sums`, one array for all set_action calls of the step,
//           in call order) and the exclusive prefix of every wave's count inside the tile (`wpre`); whoever needs the number
//           computes it from those and one ballot (attack_seq, called by the step's first per-agent pass k_attack_rank).  Round 3
//           ran a second launch per call for it (k_set_action_c).  The list's length is the sum of ATT_SLOTS spread counters.
constexpr int SCAN_WAVES = SCAN_TILE / 64;
__global__ void __launch_bounds__(SCAN_THREADS) k_set_action_a(WorldView W, int g, const int *actions, int call_base, int *sums, int *wpre, int tile_off) {
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
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

// (launch.h: SeqPlan -- where a group's set_action call of this step left its tile counts in `sums` / `wpre`; -1: its sequence numbers
// are in `key` already -- the one-workgroup form k_set_action_solo assigns them itself -- or the group was given no actions)
// the sequence number of agent i's attack in the step's attack list (GridWorld.cc:435-445: list order = call order, then agent order).
// Called by EVERY thread of a 256-thread workgroup whose agents lie in one tile (a barrier and a ballot inside); `att`: i attacks
__device__ __forceinline__ int attack_seq(const int *sums, const int *wpre, int tile_off, int i, bool att) {
    const int tile = tile_off + i / SCAN_TILE;
    const int before = block_prefix(sums, tile);
    int wtot;
    const int r = wave_rank(att, wtot);
    return before + wpre[(size_t)tile * SCAN_WAVES + (i % SCAN_TILE) / 64] + r;
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

// (takes the group and the type, not the world: indexing the by-value kernel argument with a run-time group number would make
// the compiler keep a per-lane copy of the whole world description in scratch memory)
__device__ __forceinline__ void set_action_solo_body(const GroupDev &G, const TypeDev &T, int *counters, int large_map, int bandwidth,
                                                     const int *actions, int call_base) {
    const int base = counters[CTR_ATTACK];
    for (int i = threadIdx.x; i < G.n; i += SOLO_THREADS) {
        int act = actions[i];
        if (act < 0 || act >= T.n_move + T.n_turn + T.n_attack) {
            counters[CTR_BAD_ACTION] = 1;
            G.pend[i] = PEND_NONE;
        } else if (act < T.n_move + T.n_turn) {
            unsigned bound = 0;
            if (large_map) { int x_ = G.x[i] % bandwidth; bound = (x_ < 4 || x_ > bandwidth - 4) ? 1u : 0u; }
            G.pend[i] = (act < T.n_move ? PEND_MOVE : PEND_TURN) | act;
            G.key[i] = (bound << 31) | (unsigned)(call_base + i);
        } else {
            G.pend[i] = PEND_ATTACK | (act - T.n_move - T.n_turn);
        }
    }
    __syncthreads();   // base was read by every thread before the total is written back
    int total = solo_rank([&](int i) { return actions[i] >= T.n_move + T.n_turn; }, [&](int i, int r) { G.key[i] = (unsigned)r; }, G.n, base);
    if (threadIdx.x == 0) counters[CTR_ATTACK] = total;
}
__global__ void __launch_bounds__(SOLO_THREADS) k_set_action_solo(WorldView W, int g, const int *actions, int call_base) {
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    set_action_solo_body(G, T, W.counters, W.large_map, W.bandwidth, actions, call_base);
}

__device__ __forceinline__ int pend_action(int pend, const TypeDev &T) {   // the action number a pending action came from
    return (pend & PEND_ARG) + ((pend & ~PEND_ARG) == PEND_ATTACK ? T.n_move + T.n_turn : 0);
}
// last_action of a group whose actions are set but not stepped yet (an observation between set_action and step)
__global__ void __launch_bounds__(256) k_commit_action(GroupDev G, TypeDev T) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= G.n) return;
    const int pend = G.pend[i];
    if (pend != PEND_NONE) G.last_action[i] = pend_action(pend, T);
}

// ------------------------------------------------------------------------------------------------ attack shuffle
// The reference shuffles the attack list with `for i: j = (int)rng() % (i + 1); swap(buf[i], buf[j])`
// (GridWorld.cc:464-468), rng = minstd_rand0.  Exact parallel replay in two launches:
//   draw   j_i from the i-th engine output, by LCG skip-ahead: r_i = 16807^(i+1) * x0 mod (2^31 - 1).  Every step threads
//          itself onto the list of its slot (head[v] -> the steps k with j_k == v, in arrival order: one atomicExch) and offers
//          itself as the first LATER step that hits the slot (first[v] = the smallest m != v with j_m == v: one atomicMax of
//          0x7FFFFFFF - m, so that the rest state of head[] and first[] is zero).
//   chase  element i sits at j_i after step i; it is moved again by the first later step k whose j_k equals its
//          position, and then sits at k.  The first hop walks the list of slot j_i for the smallest entry above i (lists hold
//          ln(A / v) entries on average); from then on the element sits at the slot of the step that moved it and every
//          further hop is one load of first[].  The chain (expected length O(1), longest O(log A)) ends at the final position.
// (Round 1 built the lists with a counting sort -- count, scan, fill: three more launches -- and searched a bucket per hop.)
__device__ __forceinline__ unsigned mulmod31(unsigned a, unsigned b) {
    unsigned long long p = (unsigned long long)a * b;
    unsigned long long r = (p & 0x7FFFFFFFull) + (p >> 31);
    r = (r & 0x7FFFFFFFull) + (r >> 31);
    return (unsigned)(r >= 0x7FFFFFFFull ? r - 0x7FFFFFFFull : r);
}

// powtab: 16807^t mod (2^31 - 1) for t = 0..255, then 16807^(256 h) for h = 0, 1, ... (host-computed, engine.hip)
__device__ __forceinline__ void shuffle_draw_body(unsigned x0, int i, int *j, int *head, int *first, int *link, const unsigned *powtab) {
    const unsigned e = (unsigned)i + 1u;               // the i-th draw is x0 * 16807^(i+1): two table factors
    const unsigned acc = mulmod31(mulmod31(x0, powtab[256 + (e >> 8)]), powtab[e & 255u]);
    int ji = (int)(acc % (unsigned)(i + 1));   // (int)rng() % (i + 1): outputs are in [1, 2^31 - 2]
    j[i] = ji;
    link[i] = atomicExch(&head[ji], i + 1);    // entries are step + 1: 0 ends a list
    if (ji != i) atomicMax(&first[ji], 0x7FFFFFFF - i);
}
// the engine state after the shuffle's A draws: x <- 16807^A x
__device__ __forceinline__ unsigned rng_skip(unsigned x, unsigned n) {
    unsigned base = 16807u;
    while (n) { if (n & 1u) x = mulmod31(x, base); base = mulmod31(base, base); n >>= 1; }
    return x;
}
// (A: the length of this step's attack list)
__device__ __forceinline__ void shuffle_chase_body(int i, int A, const int *j, const int *head, const int *first, const int *link, int *rank) {
    int p = j[i];
    int nxt = 0x7FFFFFFF;
    for (int e = head[p]; e != 0; e = link[e - 1]) { const int k = e - 1; if (k > i && k < nxt && k < A) nxt = k; }
    if (nxt != 0x7FFFFFFF) {
        p = nxt;
        for (int f; (f = first[p]) != 0;) { const int m = 0x7FFFFFFF - f; if (m >= A) break; p = m; }
    }
    rank[i] = p;
}

// (the draws of this step's list and the hit words' zero-fill.  The list's length: counters[CTR_ATTACK] when the one-workgroup
// set_action left it there, else -- `tiled` -- the sum of the spread counters of k_set_action_a, which workgroup 0 then leaves in
// CTR_ATTACK for every later launch of the step)
__global__ void __launch_bounds__(256) k_shuffle_draw(int *counters, int *j, int *head, int *first, int *link, unsigned *hitbits, size_t ncell,
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

// ------------------------------------------------------------------------------------------------ attack phase
// rank[seq] = position of attack-list entry `seq` after the reference's shuffle (GridWorld.cc:464-468)
// (tlist / n_tlist, one-launch step with one-cell bodies: the attacker that sets the FIRST bit of a cell appends the agent standing
// there -- every target exactly once -- and the evaluation rounds visit the targets instead of scanning every agent)
__device__ __forceinline__ void attack_rank_body(const WorldView &W, int g, int i, const int *rank, unsigned *hitbits, int *tlist = nullptr,
                                                 int *n_tlist = nullptr, int seq = -1 /* >= 0: the attack's sequence number (else it is in `key`) */) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int pend = G.pend[i];
    const bool att = (pend & ~PEND_ARG) == PEND_ATTACK;
    const bool dead = G.dead[i];
    if (att) G.key[i] = (unsigned)rank[seq >= 0 ? (unsigned)seq : G.key[i]];
    G.drank_a[i] = dead ? -1 : RANK_INF;   // agents dead before the phase never act and are not on the map
    G.drank_b[i] = 0;                      // "inputs changed in round 0": everybody is evaluated in round 1
    // push one bit per (attacker group, attack offset) onto the target's cell: targets then enumerate only the
    // hits they actually receive (one word per target instead of a scan of every attack offset around it)
    if (att && !dead) {
        const int k = pend & PEND_ARG;
        const int2 tc = attack_target(W, G, T, i, k);
        int tx = tc.x, ty = tc.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            int o = W.occ[ty * W.w + tx];
            // Map::get_attack_obj (Map.cc:229-247).  In food_mode an attack aimed at a comrade is recorded too: it does no
            // damage, but once the comrade has been killed the food it leaves can be eaten by anybody
            if ((o >= 0 && (T.attack_in_group || ref_group(o) != g || W.food_mode)) || o == OCC_FOOD) {
                if (!tlist) atomicOr(&hitbits[ty * W.w + tx], 1u << (T.attack_bit + k));
                else if (atomicOr(&hitbits[ty * W.w + tx], 1u << (T.attack_bit + k)) == 0u) tlist[atomicAdd(n_tlist, 1)] = o;
            }
        }
    }
    if (W.food_mode) { G.eat[i] = -1.0f; G.fcell[i] = -1; }
}
__global__ void __launch_bounds__(256) k_attack_rank(WorldView W, const int *rank, unsigned *hitbits, int *shuf_head, int *shuf_first,
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
    attack_rank_body(W, g, i, rank, hitbits, nullptr, nullptr, seq);
}

// The hits that land on cell (cx, cy), appended to a thread-private LDS list (stride NT): bit (attack_bit[ga] + k) of
// the cell's word is set iff the agent standing at cell - delta(ga, k) attacks it with offset k.
__device__ __forceinline__ int gather_hits(const WorldView &W, unsigned bits, int cx, int cy, unsigned *s_rank, int *s_ref, int NT, int tid, int nh, int kmax) {
    for (int ga = 0; ga < W.G; ga++) {
        const TypeDev TA = W.type[ga];
        if (TA.n_attack == 0) continue;
        unsigned mine = (bits >> TA.attack_bit) & (TA.n_attack >= 32 ? 0xFFFFFFFFu : ((1u << TA.n_attack) - 1u));
        const GroupDev A = W.grp[ga];
        while (mine) {
            int k = __ffs(mine) - 1;
            mine &= mine - 1;
            int2 d = W.delta[TA.attack_off + k];
            if (!W.turn_mode) {
                int o = W.occ[(cy - d.y) * W.w + (cx - d.x)];   // the attacker's own top-left cell
                int ai = ref_index(o);
                s_rank[nh * NT + tid] = A.key[ai]; s_ref[nh * NT + tid] = o;
                nh++;
                continue;
            }
            // turn_mode: the bit does not say which way the attacker faces -- one candidate per direction, each checked
            // (several of them can be real: two agents facing different ways reach one cell with the same offset number)
            for (int dir = 0; dir < DIR_NUM; dir++) {
                int ax, ay, px, py;
                dir_rotate(dir, d.x, d.y, ax, ay);
                real_to_saved(dir, TA.bw, TA.bl, cx - ax, cy - ay, px, py);    // reference corner -> the body's top-left cell
                if (px < 0 || py < 0 || px >= W.w || py >= W.h) continue;
                const int o = W.occ[py * W.w + px];
                if (o < 0 || ref_group(o) != ga) continue;
                const int ai = ref_index(o);
                if (A.dir[ai] != dir || A.pend[ai] != (PEND_ATTACK | k) || A.x[ai] != px || A.y[ai] != py) continue;
                if (nh >= kmax) { W.counters[CTR_HIT_OVERFLOW] = 1; continue; }   // (reported at the end of the step, never silent)
                s_rank[nh * NT + tid] = A.key[ai]; s_ref[nh * NT + tid] = o;
                nh++;
            }
        }
    }
    return nh;
}
// insertion sort of the list by rank (ranks are unique)
__device__ __forceinline__ void sort_hits(unsigned *s_rank, int *s_ref, int NT, int tid, int nh) {
    for (int a = 1; a < nh; a++) {
        unsigned r = s_rank[a * NT + tid]; int f = s_ref[a * NT + tid];
        int b = a - 1;
        while (b >= 0 && s_rank[b * NT + tid] > r) {
            s_rank[(b + 1) * NT + tid] = s_rank[b * NT + tid];
            s_ref[(b + 1) * NT + tid] = s_ref[b * NT + tid];
            b--;
        }
        s_rank[(b + 1) * NT + tid] = r; s_ref[(b + 1) * NT + tid] = f;
    }
}
// the cell an attacker aims at (its pending action is an attack)
__device__ __forceinline__ int attack_cell(const WorldView &W, const GroupDev *gtab, int a) {
    const GroupDev A = gtab[ref_group(a)];
    const int ai = ref_index(a);
    const int2 tc = attack_target(W, A, W.type[ref_group(a)], ai, A.pend[ai] & PEND_ARG);
    return tc.y * W.w + tc.x;
}
// food_mode: one attacker eats from what is left on a cell (Map.cc:292-303).  `eat` of an attacker is written by the
// owner of its target cell only; a change sends the attacker back into evaluation.
__device__ __forceinline__ bool set_eat(const WorldView &W, const GroupDev *gtab, int a, float e, int round, int *flagp) {
    const GroupDev A = gtab[ref_group(a)];
    const int ai = ref_index(a);
    if (A.eat[ai] == e) return false;
    A.eat[ai] = e;
    A.drank_b[ai] = round;
    if (flagp) *flagp = 1;
    return true;
}

// Exact parallel form of the sequential attack loop.  For a target t the incoming hits are found by PULLING:
// for every attacker group g' and attack offset d of g', the only agent that can hit t with d stands at
// pos(t) - d; it hits iff its pending action is "attack with offset d".  Hits are sorted by rank (LDS) and replayed
// in order: a hit counts iff its attacker is still alive at that rank (death_rank[attacker] > rank).  death_rank
// (drank_a) is iterated IN PLACE to its fixed point, which is unique because every event only depends on events of
// lower rank: an agent is re-evaluated in round r only if one of its inputs changed in round r - 1 or earlier in
// round r (drank_b holds the last round in which an input changed), so after the first round only the neighbourhood
// of the deaths is touched; a round without any change leaves every agent consistent with its inputs.
// (s_rank / s_ref: the thread's hit list, stride ATT_THREADS, slot tid; flagp: where to report a change, or null)
__device__ __forceinline__ void attack_eval_body(const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int g, int i,
                                                 int round /* 1, 2, ... within this step */, const unsigned *hitbits,
                                                 unsigned *s_rank, int *s_ref, int ATT_THREADS, int tid, int *flagp, int kmax) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int dr_me_cur = G.drank_a[i];
    if (dr_me_cur == -1) return;                      // dead before the phase
    if (G.drank_b[i] < round - 1) return;             // no input has changed since my last evaluation
    const int pend = G.pend[i];
    const bool attacker = (pend & ~PEND_ARG) == PEND_ATTACK;

    const int x = G.x[i], y = G.y[i];
    // ---- gather incoming hits: bit (attack_bit[ga] + k) of my cell's word is set iff the agent standing at
    // pos - delta(ga, k) attacks me with offset k
    int nh = 0;
    const int2 fp = body_dims(W, G, T, i);
    for (int by = 0; by < fp.y; by++)
        for (int bx = 0; bx < fp.x; bx++) {      // an attacker hits ONE cell; a multi-cell body collects from all of its cells
            const int cx = x + bx, cy = y + by;
            unsigned bits = hitbits[cy * W.w + cx];
            if (bits) nh = gather_hits(W, bits, cx, cy, s_rank, s_ref, ATT_THREADS, tid, nh, kmax);
        }
    if (nh == 0) return;                              // nobody hits me: I stay alive (RANK_INF, the initial value)
    sort_hits(s_rank, s_ref, ATT_THREADS, tid, nh);
    // ---- own attack (needed for kill_supply replay and, in APPLY, for the attacker-side results)
    unsigned my_rank = 0xFFFFFFFFu;
    int tgt = -1;          // packed ref of my target at phase start, -1 = blank / wall / out of board / same group
    int aimed = -1;        // the agent on the cell I aim at, comrade or not (food_mode: it may leave food for me)
    if (attacker) {
        my_rank = G.key[i];
        const int2 tc = attack_target(W, G, T, i, pend & PEND_ARG);
        int tx = tc.x, ty = tc.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            int o = W.occ[ty * W.w + tx];
            if (o >= 0) aimed = o;
            if (o >= 0 && (T.attack_in_group || ref_group(o) != g)) tgt = o;
        }
    }
    // death rank of my target as of the current iterate
    int tgt_dr = RANK_INF;
    if (tgt >= 0) tgt_dr = gtab[ref_group(tgt)].drank_a[ref_index(tgt)];
    const bool kill = W.any_kill_supply && tgt >= 0 && (unsigned)tgt_dr == my_rank;
    // what my own attack feeds me at my rank (add_hp: capped at the type's hp even when it adds nothing): the kill supply,
    // or in food_mode what I eat (the owner of the food says how much; -1 = my attack meets no food)
    const float eaten = W.food_mode && attacker ? G.eat[i] : -1.0f;
    const bool supply = kill || eaten >= 0.0f;
    const float bonus = kill ? ttab[ref_group(tgt)].kill_supply : eaten;

    // ---- replay in rank order
    float hp = G.hp[i];
    int dr = RANK_INF, kd = nh;                        // kd: which hit kills me
    bool supplied = !supply;
    for (int k = 0; k < nh; k++) {
        unsigned r = s_rank[k * ATT_THREADS + tid];
        if (!supplied && my_rank < r) { hp = fminf(T.hp, hp + bonus); supplied = true; }
        int a = s_ref[k * ATT_THREADS + tid];
        const GroupDev A = gtab[ref_group(a)];
        int adr = A.drank_a[ref_index(a)];
        // the attacker is alive when its turn comes iff it did not die at an EARLIER rank.  adr == r happens only when
        // the attacker is this very agent hitting its own body (in-group attack of a body whose range covers its own
        // cells) and that hit is the fatal one: the attack did run (RANK_INF >= any rank)
        if ((unsigned)adr >= r && (ref_group(a) != g || T.attack_in_group)) {   // (food_mode lists comrades' attacks too: no damage)
            hp -= ttab[ref_group(a)].damage;
            if (hp < 0.0f) { dr = (int)r; kd = k; break; }   // death iff hp < 0 strictly (GridWorld.h:205)
        }
    }
    // the kill supply of my own attack: normally skipped once I am dead -- except when I killed MYSELF, where
    // Map::do_attack still feeds the (dead) attacker (Map.cc:266-274)
    const bool self_kill = tgt == ref_pack(g, i) && (unsigned)dr == my_rank;
    if (!supplied && (dr == RANK_INF || self_kill)) hp = fminf(T.hp, hp + bonus);

    if (W.food_mode) {
        // Killed: my food_supply lies on the cell the fatal hit landed on (the rest of my body is cleared), and the later
        // hits on that cell eat from it, in rank order, until less than 0.1 is left (Map.cc:276-303).  Everybody else who
        // hits me meets an agent or nothing: their `eat` goes back to -1.
        int c_food = -1;
        float food = 0.0f;
        bool present = false;
        if (kd < nh) { c_food = attack_cell(W, gtab, s_ref[kd * ATT_THREADS + tid]); food = T.food_supply; present = true; }
        for (int k = 0; k < nh; k++) {
            const int a = s_ref[k * ATT_THREADS + tid];
            float e = -1.0f;
            if (k > kd && present && attack_cell(W, gtab, a) == c_food) {
                const unsigned r = s_rank[k * ATT_THREADS + tid];
                if ((unsigned)gtab[ref_group(a)].drank_a[ref_index(a)] >= r) {       // alive at its turn
                    e = fminf(ttab[ref_group(a)].eat_ability, food);
                    food -= e;
                    if ((double)food < 0.1) present = false;
                }
            }
            set_eat(W, gtab, a, e, round, flagp);
        }
        G.fcell[i] = present ? c_food : -1;
        G.fleft[i] = food;
    }

    G.mv[i] = __float_as_uint(hp);                    // for k_attack_apply: final once the death ranks are
    if (dr != dr_me_cur) {
        G.drank_a[i] = dr;
        // who reads my death rank: my target (is its attacker alive at that rank?) and, for the kill supply, my attackers
        const int reader = W.food_mode ? aimed : tgt;
        if (reader >= 0) gtab[ref_group(reader)].drank_b[ref_index(reader)] = round;
        if (W.any_kill_supply)
            for (int k = 0; k < nh; k++) { const int a = s_ref[k * ATT_THREADS + tid]; gtab[ref_group(a)].drank_b[ref_index(a)] = round; }
        if (flagp) *flagp = 1;                        // (multi-launch driver: only the last round of a batch reports)
    }
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

// The converged phase applied: hp, death, rewards, last_op / op_obj.  Nothing is replayed here: every agent that is hit
// left the hp of its LAST evaluation in `mv` (that evaluation saw the final death ranks -- otherwise the agent would
// have been marked and evaluated again), and the attacker-side results only need the death ranks.
__device__ __forceinline__ void attack_apply_body(const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int g, int i, const unsigned *hitbits) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int dr = G.drank_a[i];
    if (dr == -1) return;                             // dead before the phase
    const int pend = G.pend[i];
    const bool attacker = (pend & ~PEND_ARG) == PEND_ATTACK;
    const int x = G.x[i], y = G.y[i];
    bool hit = false;
    const int2 fp = body_dims(W, G, T, i);
    for (int by = 0; by < fp.y; by++)
        for (int bx = 0; bx < fp.x; bx++) hit |= hitbits[(y + by) * W.w + x + bx] != 0;
    if (!hit && !attacker) return;
    unsigned my_rank = 0xFFFFFFFFu;
    int tgt = -1, tgt_dr = RANK_INF;
    if (attacker) {
        my_rank = G.key[i];
        const int2 tc = attack_target(W, G, T, i, pend & PEND_ARG);
        int tx = tc.x, ty = tc.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            int o = W.occ[ty * W.w + tx];
            if (o >= 0 && (T.attack_in_group || ref_group(o) != g)) tgt = o;
        }
        if (tgt >= 0) tgt_dr = gtab[ref_group(tgt)].drank_a[ref_index(tgt)];
    }
    const float eaten = W.food_mode && attacker ? G.eat[i] : -1.0f;   // >= 0: my attack ate (food_mode)
    float hp;
    if (hit) hp = __uint_as_float(G.mv[i]);
    else {                                            // nobody hit me: only my own kill, or what I eat, feeds me (Map.cc:266-303)
        hp = G.hp[i];
        if (W.any_kill_supply && tgt >= 0 && (unsigned)tgt_dr == my_rank) hp = fminf(T.hp, hp + ttab[ref_group(tgt)].kill_supply);
        else if (eaten >= 0.0f) hp = fminf(T.hp, hp + eaten);
    }
    const bool self_kill = tgt == ref_pack(g, i) && (unsigned)dr == my_rank;
    float nr = G.next_reward[i];
    float own = 0.0f;                                  // what my own attack adds to my reward
    bool acted = false;
    if (attacker && (unsigned)dr >= my_rank) {         // alive at my turn (GridWorld.cc:479-480)
        acted = true;
        if (eaten >= 0.0f) {                           // food: do_attack returns 0.0 (Map.cc:292-303, GridWorld.cc:505)
            own = 0.0f + T.attack_penalty;
        } else if (tgt < 0 || (unsigned)tgt_dr < my_rank) {   // blank, or the target died before my turn (Map.cc:229-231)
            own = T.attack_penalty;
        } else {
            float reward = 0.0f;
            if ((unsigned)tgt_dr == my_rank) { G.last_op[i] = OP_KILL; reward = ttab[ref_group(tgt)].kill_reward; }
            else G.last_op[i] = OP_ATTACK;
            G.op_obj[i] = tgt;
            own = reward + T.attack_penalty;           // add_reward(reward + attack_penalty) (GridWorld.cc:505)
        }
    }
    G.hp[i] = hp;
    if (dr != RANK_INF) {
        G.dead[i] = 1;                                 // counted, and taken off the map, by starve_body: other lanes of
                                                       // THIS launch still find their targets through the map
        // dead_penalty overwrites what was accumulated (GridWorld.h:207); only a self-inflicted death is followed by
        // the attacker's own add_reward (the overwrite happens inside do_attack, the add after it)
        nr = self_kill ? T.dead_penalty + own : T.dead_penalty;
    } else if (acted) nr += own;
    G.next_reward[i] = nr;
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

// ------------------------------------------------------------------------------------------------ starve / recover
// (device function: runs at the head of the move-preparation launch -- one dependent launch less per step)
__device__ __forceinline__ void starve_body(const WorldView &W, int g, const GroupDev &G, const TypeDev &T, int i, int slot) {
    bool died = false;
    if (i < G.n) {      // Agent::set_action's `last_action = act` (see k_set_action_a)
        const int pend = G.pend[i];
        if (pend != PEND_NONE) G.last_action[i] = pend_action(pend, T);
    }
    // first: the agents that died in this step's attack phase leave the map (Map::remove_agent, Map.cc:272) -- here, in
    // the launch after the attack's, because the attack kernels find attackers through the phase-start map
    // (and are counted here, one atomic per wave, together with the starved)
    if (i < G.n && W.counters[CTR_ATTACK] != 0) {
        const int dr = G.drank_a[i];
        if (dr != -1 && dr != RANK_INF) {
            died = true;
            const int2 fp = body_dims(W, G, T, i);
            cells_clear(W, G.x[i], G.y[i], fp.x, fp.y);
            if (W.food_mode && G.fcell[i] >= 0) {   // Map.cc:276-283
                W.occ[G.fcell[i]] = OCC_FOOD; W.food[G.fcell[i]] = G.fleft[i];
                if (W.live_paint) vc_store(W, G.fcell[i], OCC_FOOD, 0u);
            }
        }
    }
    if (i < G.n && !G.dead[i]) {
        float hp = G.hp[i];
        if (T.step_recover > 0) hp = fminf(T.hp, hp + T.step_recover);
        else {
            hp -= -T.step_recover;
            if (hp < 0.0f) {
                died = true; G.dead[i] = 1; G.next_reward[i] = T.dead_penalty;
                const int2 fp = body_dims(W, G, T, i);
                cells_clear(W, G.x[i], G.y[i], fp.x, fp.y);
            }
        }
        G.hp[i] = hp;
    }
    int wtot; wave_rank(died, wtot);
    if (wtot && lane_id() == 0) atomicAdd(&W.counters[dead_slot(g, slot)], wtot);
}

// ------------------------------------------------------------------------------------------------ move phase
// Exact parallel form of "first come in key order, vacate-then-enter chains" (Map.cc:313-333):
//   a cell empty at phase start goes to its lowest-key contender; a cell occupied by O is freed at key(O) iff O's
//   own move succeeds, and then goes to the lowest-key contender with key > key(O).  Which contender that would be
//   is static (64-bit atomic umin of {key, ref} per cell); whether O leaves is a chain of such dependencies that
//   only points to lower keys, resolved by pointer jumping.
// tgt (= drank_a, free after the attack phase): target cell of a move candidate, -1 otherwise.
// (every lane of a wave calls this, i >= n included: starve_body counts the dead with a wave ballot)
__device__ __forceinline__ void move_prep_body(const WorldView &W, int g, int i, int slot) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    starve_body(W, g, G, T, i, slot);
    if (i >= G.n) return;
    int t = -1;
    int pend = G.pend[i];
    // turn_mode, 1x1 bodies (GridWorld.cc:544-571, Map::do_turn Map.cc:361-406): the body turns about its own cell, nothing
    // can be in the way.  The reference takes the turn's payload from move_base, so `wise` = 2 * action - 1 is an odd number
    // >= 1: the direction changes by wise (mod 4).  Turns come after starvation and before the moves; a mover does not turn.
    if (W.turn_mode && !G.dead[i] && (pend & ~PEND_ARG) == PEND_TURN)
        G.dir[i] = (G.dir[i] + (pend & PEND_ARG) * 2 - 1 + DIR_NUM) % DIR_NUM;
    if (!G.dead[i] && (pend & ~PEND_ARG) == PEND_MOVE) {
        int2 d = agent_delta(W, G, i, T.move_off, pend & PEND_ARG);
        int nx = G.x[i] + d.x, ny = G.y[i] + d.y;
        // is_blank_area bounds (Map.cc:455) for a 1x1 body; a zero move "succeeds" in place and never vacates
        if ((d.x | d.y) != 0 && nx >= 0 && ny >= 0 && nx + 1 < W.w && ny + 1 < W.h && W.occ[ny * W.w + nx] != OCC_WALL)
            t = ny * W.w + nx;
    }
    G.drank_a[i] = t;
    G.mv[i] = MV_FAIL;
}
__global__ void __launch_bounds__(256) k_move_prep(WorldView W, unsigned *claim_words, size_t n_words) {
    if (attack_open(W)) return;
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // move rounds start
    // the claim words back to "nobody" (they held the attack phase's hit bits until now)
    for (size_t k = ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < n_words;
         k += (size_t)gridDim.x * gridDim.y * blockDim.x) claim_words[k] = 0xFFFFFFFFu;
    move_prep_body(W, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, blockIdx.x % DEAD_SLOTS);
}

__device__ __forceinline__ void move_claim_body(const WorldView &W, const GroupDev *gtab, int g, int i) {
    const GroupDev &G = W.grp[g];
    int c = G.drank_a[i];
    if (c < 0) return;
    unsigned key = G.key[i];
    int o = W.occ[c];
    G.drank_b[i] = o;            // the phase-start content of my target cell, for k_move_commit (which rewrites the map)
    bool ok = o == OCC_EMPTY;
    if (o >= 0) {
        const GroupDev O = gtab[ref_group(o)];
        int oi = ref_index(o);
        ok = O.drank_a[oi] >= 0 && O.key[oi] < key;   // the occupant may leave, and before my turn
    }
    if (ok) atomicMin(&W.claim[c], ((unsigned long long)key << 32) | (unsigned)ref_pack(g, i));
}
__global__ void __launch_bounds__(256) k_move_claim(WorldView W, const GroupDev *gtab) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) move_claim_body(W, gtab, g, i);
}

__device__ __forceinline__ void move_init_body(const WorldView &W, int g, int i) {
    const GroupDev &G = W.grp[g];
    int c = G.drank_a[i];
    if (c < 0) return;
    unsigned long long cl = W.claim[c];
    if ((unsigned)cl != (unsigned)ref_pack(g, i)) return;          // not the static winner: stays MV_FAIL
    int o = W.occ[c];
    if (o == OCC_EMPTY) G.mv[i] = MV_OK;
    else G.mv[i] = (unsigned)o;                                    // succeeds iff the occupant o succeeds
}
__global__ void __launch_bounds__(256) k_move_init(WorldView W) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) move_init_body(W, g, i);
}

// Whether a mover's chain of dependencies ends in success: its move succeeds iff the phase-start occupant of its target succeeds,
// iff ... -- every link points to a LOWER key, so the walk terminates; links are read-only once k_move_init has run.  The chain
// is walked by whoever needs the answer (round 2 resolved every agent by pointer jumping first: three to six more dependent
// launches -- or workgroup barriers, in the one-launch step -- to shorten chains that are one or two links long on average;
// a long "conga line" now costs its length in dependent loads to the agents at its tail, and nothing to anybody else).
__device__ __forceinline__ unsigned move_resolve(const GroupDev *gtab, unsigned m) {
    while (m < MV_OK) m = gtab[ref_group((int)m)].mv[ref_index((int)m)];
    return m;
}

// End of the 1x1 move phase, one launch.
// Successful moves: leave the old cell, enter the new one.  A cell that a successful mover leaves is either entered by
// the static winner of that cell (which succeeds exactly when the leaver does, and then writes the cell itself) or by
// nobody (no claim on it: the leaver clears it) -- no cell is written by two agents.
// Failed moves: collide bookkeeping (Map.cc:334-353) from what k_move_claim saved of the phase-start map (drank_b),
// the claims and the move states -- nothing that this launch writes.
__device__ __forceinline__ void move_commit_body(const WorldView &W, const GroupDev *gtab, int g, int i) {
    const GroupDev &G = W.grp[g];
    const int c = G.drank_a[i];
    if (c >= 0) {
        if (move_resolve(gtab, G.mv[i]) == MV_OK) {
            const int old = G.y[i] * W.w + G.x[i];
            if (W.claim[old] == CLAIM_NONE) { W.occ[old] = OCC_EMPTY; if (W.live_paint) vc_store(W, old, OCC_EMPTY, 0u); }
            W.occ[c] = ref_pack(g, i);
            const int ny = c / W.w;
            G.x[i] = c - ny * W.w; G.y[i] = ny;
        } else {
            const int o = G.drank_b[i];
            int blocker;
            if (o == OCC_FOOD) { G.pend[i] = PEND_NONE; return; }          // food blocks, but get_collide only sees agents (Map.cc:493)
            if (o == OCC_EMPTY) blocker = (int)(unsigned)W.claim[c];       // lost an empty cell to the lowest key
            else {
                const GroupDev O = gtab[ref_group(o)];
                int oi = ref_index(o);
                bool left_before = move_resolve(gtab, O.mv[oi]) == MV_OK && O.key[oi] < G.key[i];
                blocker = left_before ? (int)(unsigned)W.claim[c] : o;
            }
            G.last_op[i] = OP_COLLIDE;
            G.op_obj[i] = blocker;
        }
    }
    G.pend[i] = PEND_NONE;   // end of step: pending actions are consumed (also done by k_finish for the generic path)
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

// ------------------------------------------------------------------------------------------------ the step of plain games
// Plain games -- one-cell bodies, no turn_mode / food_mode / goals / kill_supply: battle, gather, every BASELINE configuration but the
// reference's own 1M harness -- have a pipeline of their own behind the shuffle (round 4).  Five kinds of per-agent launches where the
// generic step has nine, and half the dependent gathers per launch:
//   k_plain_rank   every agent: its record {order key | rank in the shuffled attack list, death rank = "never", the cell its move is aimed
//                  at}.  An attacker looks its target up ONCE, here: it keeps the target's reference (`atk`) and
//                  hands the target its hit -- {rank, attacker} into the target's own slot for (attacker group, attack offset), one bit
//                  into the target's hit mask.  Nobody looks an attacker up through the map afterwards.
//   k_plain_eval   the death-rank fixed point (attack_eval_body for this case): an agent reads its mask (coalesced), its own slots, and
//                  the death rank of each attacker -- one gather per hit where the generic form has three (map, key, death rank) plus two
//                  for its own target.  A round whose predecessor changed nothing returns at once.
//   k_strike       what k_attack_apply, starve_body, k_rule (rules that pay the attacker), k_move_prep and k_move_claim do in five
//                  launches: every agent finishes its own attack phase from the converged death ranks, starves / recovers, is paid by the
//                  rules, and claims its target cell.  Nobody writes the map in this pass, so occupants are still found through the
//                  phase-start map; whether an occupant is still there when the moves begin -- it may have been killed, or starve -- is
//                  decided by the claimant from the occupant's record and `mv` (what starve_body would do with it).
//   k_plain_init   who won its cell, and on whom its move depends (k_move_init without the map lookup: k_strike saved what it saw);
//                  the agents that died in this step leave the map here, after its last reader
//   k_plain_commit k_move_commit on the records
// Per-agent state that other agents read lives in ONE 16-byte record per agent (`rec`: a claimant reads its occupant's key, death rank
// and target with one request).  No per-cell pass is left in the step -- at BASELINE config 5's 3536 x 3536 cells the two fills of the
// generic step are 150 MB per step: the hit masks are per agent and cleaned by their owners (k_strike), and a claim word carries the
// EPOCH of the step that wrote it in its top bits (claim_word below), counting DOWN from step to step: a word of an earlier step loses
// every atomicMin against this step's claims and reads as "nobody" -- nothing is cleaned; every 63rd step the host refills the array
// (engine.hip: scratch_for).  The "inputs changed" stamps of the rounds count on across steps in the same way (PlainWorld::round_base).
constexpr unsigned MV_DIED = 0xFFFFFFFBu;   // move status between k_strike and k_plain_init: killed or starved in this step, still on the map
constexpr unsigned MV_FAIL_SAME = 0xFFFFFFFCu;   // MV_FAIL of an agent whose hp this step left as it was: its painted cell is current (k_plain_commit)
// claim word of the plain pipeline: [63:58] epoch (0..62; 63 = the fill pattern: nobody) | [57:30] order key (boundary bit, 27-bit insertion
// index) | [29:0] agent reference.  Smaller = earlier: a later step's epoch is smaller, so stale words never win
__device__ __forceinline__ unsigned long long claim_word(int epoch, unsigned key, int ref) {
    const unsigned long long k28 = ((unsigned long long)(key >> 31) << 27) | (key & 0x7FFFFFFu);
    return ((unsigned long long)epoch << 58) | (k28 << 30) | (unsigned)ref;
}
__device__ __forceinline__ bool claim_live(unsigned long long w, int epoch) { return (int)(w >> 58) == epoch; }
__device__ __forceinline__ int claim_ref(unsigned long long w) { return (int)(w & 0x3FFFFFFFu); }

__global__ void __launch_bounds__(256) k_plain_rank(WorldView W, PlainWorld PW, const PlainGroup *ptab, const int *rank, int *shuf_head, int *shuf_first,
                                                   const int *sums, const int *wpre, SeqPlan P) {
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // attack rounds start
    const int A = W.counters[CTR_ATTACK];
    // the shuffle's list heads and first-hit words have been read for the last time (k_shuffle_chase): back to zero for their next use
    for (int k = (blockIdx.y * gridDim.x + blockIdx.x) * blockDim.x + threadIdx.x; k < A; k += gridDim.x * gridDim.y * blockDim.x) {
        shuf_head[k] = 0; shuf_first[k] = 0;
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
        key = (unsigned)rank[seq >= 0 ? seq : (int)key];
        const int k = pend & PEND_ARG;
        const int2 d = W.delta[T.attack_off + k];
        const int tx = x + d.x, ty = y + d.y;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) {
            const int o = W.occ[ty * W.w + tx];
            if (o >= 0 && (T.attack_in_group || ref_group(o) != g)) {       // Map::get_attack_obj (Map.cc:229-247)
                tgt = o;
                const PlainGroup TG = ptab[ref_group(o)];
                const int slot = T.attack_bit + k;
                TG.hlist[(size_t)ref_index(o) * PW.S + slot] = make_uint2(key, (unsigned)ref_pack(g, i));
                atomicOr(&TG.hmask[ref_index(o)], 1u << slot);
            }
        }
    } else if (att) {
        key = (unsigned)rank[seq >= 0 ? seq : (int)key];   // (dead before the step: its list entry exists, and does nothing)
    } else if (!dead && (pend & ~PEND_ARG) == PEND_MOVE) {
        const int2 d = W.delta[T.move_off + (pend & PEND_ARG)];
        const int nx = x + d.x, ny = y + d.y;
        // is_blank_area bounds (Map.cc:455) for a 1x1 body; a zero move "succeeds" in place and never vacates
        if ((d.x | d.y) != 0 && nx >= 0 && ny >= 0 && nx + 1 < W.w && ny + 1 < W.h && W.occ[ny * W.w + nx] != OCC_WALL) t = ny * W.w + nx;
    }
    PW.g[g].rec[i] = make_int4((int)key, dead ? -1 : RANK_INF, t, (int)MV_FAIL);
    PW.g[g].atk[i] = tgt;
    // hp as the attack phase leaves it unless somebody hits me (k_plain_eval overwrites it then): only claimants that must know whether
    // their occupant is about to starve read it of an agent that was not hit -- types that recover never starve
    if (!(T.step_recover > 0)) G.mv[i] = __float_as_uint(G.hp[i]);
}

// (s_rank / s_ref: the thread's hit list, stride NT, slot tid -- sort_hits)
__global__ void __launch_bounds__(256) k_plain_eval(WorldView W, PlainWorld PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag) {
    if (W.counters[CTR_ATTACK] == 0) return;
    // nobody's death rank changed in the round before: nobody is stamped for this one
    if (round > 1 && W.counters[CTR_ROUND_CHANGED + ((round - 1) & (ROUND_SLOTS - 1))] == 0) return;
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
        const int adr = ptab[ref_group(a)].rec[ref_index(a)].y;
        if ((unsigned)adr >= r) {
            hp -= ttab[ref_group(a)].damage;
            if (hp < 0.0f) { dr = (int)r; break; }       // death iff hp < 0 strictly (GridWorld.h:205)
        }
    }
    G.mv[i] = __float_as_uint(hp);                       // final once the death ranks are: k_strike takes it from here
    if (dr != dr_cur) {
        PW.g[g].rec[i].y = dr;
        const int reader = PW.g[g].atk[i];               // who reads my death rank: my target (is its attacker alive at that rank?)
        if (reader >= 0) gtab[ref_group(reader)].drank_b[ref_index(reader)] = PW.round_base + round;
        if (flag >= 0) W.counters[flag] = 1;             // (only the last round of a batch reports)
        W.counters[CTR_ROUND_CHANGED + (round & (ROUND_SLOTS - 1))] = 1;
    }
}

// rules of the shape Event(a, attack | kill, b) that pay receivers bound to `a` only: evaluated by the agent itself, in rule order,
// as soon as its own attack is known -- provided every last_op was OP_NULL when the step began (clear_dead has run since the last step:
// otherwise an event of the LAST step is paid again unless a collision overwrites it, which only the move phase knows; the host then
// runs k_rule behind the commit as before)
struct StrikeRules {
    int n;
    struct One { int ga, gb, op, rule_no, n_subj; float v[4]; } r[4];
};

__global__ void __launch_bounds__(256) k_strike(WorldView W, PlainWorld PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, StrikeRules R) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const bool attacked = W.counters[CTR_ATTACK] != 0;
    bool died = false;
    unsigned trig = 0;
    if (i < G.n) {
        const int pend = G.pend[i];
        if (pend != PEND_NONE) G.last_action[i] = pend_action(pend, T);     // Agent::set_action's `last_action = act` (see k_set_action_a)
        bool dead = G.dead[i];
        const int4 me = PW.g[g].rec[i];                  // {key | rank, death rank, move target, -}
        float hp = G.hp[i];
        const unsigned hp_before = __float_as_uint(hp);
        float nr = G.next_reward[i];
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
                if (tgt >= 0) tgt_dr = ptab[ref_group(tgt)].rec[ref_index(tgt)].y;
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
        G.hp[i] = hp;
        if (died) G.dead[i] = 1;
        // ---- calc_reward for the rules that pay their subject (rule_body; the reference visits the dead too, GridWorld.cc:681-692)
        for (int k = 0; k < R.n; k++) {
            if (R.r[k].ga != g) continue;
            if (op_obj >= 0 && ref_group(op_obj) == R.r[k].gb && last_op == R.r[k].op) {
                trig |= 1u << k;
                for (int q = 0; q < R.r[k].n_subj; q++) nr += R.r[k].v[q];
            }
        }
        G.next_reward[i] = nr;
        // ---- my move: the claim on its target cell (move_prep_body + move_claim_body)
        if (!dead && me.z >= 0) {
            const int c = me.z;
            const unsigned key = (unsigned)me.x;
            int o = W.occ[c];
            bool ok = o == OCC_EMPTY;
            if (o >= 0) {
                const int4 oc = ptab[ref_group(o)].rec[ref_index(o)];
                const float orec = ttab[ref_group(o)].step_recover;
                bool gone = attacked && oc.y != RANK_INF;                                       // killed in this step's attack phase
                if (!gone && !(orec > 0)) gone = __uint_as_float(gtab[ref_group(o)].mv[ref_index(o)]) - (-orec) < 0.0f;   // ... or about to starve
                if (gone) { ok = true; o = OCC_EMPTY; }                                          // the cell is empty when the moves begin
                else ok = oc.z >= 0 && (unsigned)oc.x < key;                                     // the occupant may leave, and before my turn
            }
            PW.g[g].atk[i] = o;              // what my target cell holds when the moves begin, for k_plain_init / k_plain_commit (a mover has no attack target)
            if (ok) atomicMin(&W.claim[c], claim_word(PW.epoch, key, ref_pack(g, i)));
        }
        // (every agent's move status starts here; k_plain_init raises the winners')
        PW.g[g].rec[i].w = died ? (int)MV_DIED : __float_as_uint(hp) == hp_before ? (int)MV_FAIL_SAME : (int)MV_FAIL;
    }
    int wtot;
    wave_rank(died, wtot);
    if (wtot && lane_id() == 0) atomicAdd(&W.counters[dead_slot(g, blockIdx.x % DEAD_SLOTS)], wtot);
    for (int k = 0; k < R.n; k++)
        if (__ballot((trig >> k) & 1u) && lane_id() == 0) W.counters[CTR_TRIGGER + R.r[k].rule_no] = 1;
}

__global__ void __launch_bounds__(256) k_plain_init(WorldView W, PlainWorld PW) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev &G = W.grp[g];
    if (i >= G.n) return;
    const int4 me = PW.g[g].rec[i];
    if ((unsigned)me.w == MV_DIED) {             // Map::remove_agent (Map.cc:272, GridWorld.cc:536): nobody reads the map in this launch
        cells_clear(W, G.x[i], G.y[i], 1, 1);
        PW.g[g].rec[i].w = (int)MV_FAIL;
        return;
    }
    const int c = me.z;
    if (c < 0 || G.dead[i]) return;
    const unsigned long long cl = W.claim[c];
    if (!claim_live(cl, PW.epoch) || claim_ref(cl) != ref_pack(g, i)) return;   // not the static winner (or no claim of mine): stays MV_FAIL
    const int o = PW.g[g].atk[i];
    PW.g[g].rec[i].w = o == OCC_EMPTY ? (int)MV_OK : o;             // succeeds iff the occupant o succeeds
}

__device__ __forceinline__ unsigned plain_resolve(const PlainGroup *ptab, unsigned m) {
    while (m < MV_DIED) m = (unsigned)ptab[ref_group((int)m)].rec[ref_index((int)m)].w;
    return m;
}
// (move_commit_body on the records: see there)
__global__ void __launch_bounds__(256) k_plain_commit(WorldView W, PlainWorld PW, const PlainGroup *ptab) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    const GroupDev &G = W.grp[g];
    if (i >= G.n) return;
    const int4 me = PW.g[g].rec[i];
    const int c = me.z;
    if (c >= 0 && !G.dead[i]) {
        if (plain_resolve(ptab, (unsigned)me.w) == MV_OK) {
            const int old = G.y[i] * W.w + G.x[i];
            if (!claim_live(W.claim[old], PW.epoch)) { W.occ[old] = OCC_EMPTY; if (W.live_paint) vc_store(W, old, OCC_EMPTY, 0u); }   // nobody claimed my cell
            W.occ[c] = ref_pack(g, i);
            const int ny = c / W.w;
            G.x[i] = c - ny * W.w; G.y[i] = ny;
        } else {
            const int o = PW.g[g].atk[i];
            int blocker;
            if (o == OCC_EMPTY) blocker = claim_ref(W.claim[c]);           // lost an empty cell to the lowest key
            else {
                const int4 oc = ptab[ref_group(o)].rec[ref_index(o)];
                const bool left_before = plain_resolve(ptab, (unsigned)oc.w) == MV_OK && (unsigned)oc.x < (unsigned)me.x;
                blocker = left_before ? claim_ref(W.claim[c]) : o;
            }
            G.last_op[i] = OP_COLLIDE;
            G.op_obj[i] = blocker;
        }
    }
    G.pend[i] = PEND_NONE;   // end of step: pending actions are consumed
    // live paint: every agent that moved or whose hp changed paints its cell (most agents of a battle stand at full hp: 4 of 5 stores saved)
    if (W.live_paint && (unsigned)me.w != MV_FAIL_SAME) repaint_body(W, G, W.type[g], g, i);
}

// ------------------------------------------------------------------------------------------------ move, generic bodies
// Bodies larger than one cell (Map::do_move with width x height rectangles, Map.cc:313-333, 454-470).  A mover m with
// target rectangle T(m) succeeds iff, at its turn, every cell of T(m) outside its own body is free:
//   * the cell's phase-start occupant O has left: O moves before m (key(O) < key(m)), O's move succeeds, and O's new
//     rectangle does not cover the cell again;
//   * no mover m' with key(m') < key(m) whose move succeeds has entered it (the cell lies in T(m')).
// Both conditions only look at lower keys, so the recursion is well founded and its unique solution is the
// sequential result; it is solved by sweeps that decide every agent whose lower-key dependencies are decided.
// Entrants are found by pulling: a mover of group g' with move k and top-left p enters cell c iff
// c - d_k - (bx, by) == p for some body offset -- checked against the map and the pending actions.
struct MoveProbe {
    bool blocked;     // some cell is definitely not free at m's turn
    bool undecided;   // a lower-key dependency is still unknown
    int blocker;      // first occupant Map::get_collide would meet (x outer, y inner), -1 if none
};

// every move candidate (other than `self`) with a lower key whose target rectangle covers cell (cx, cy), found by
// pulling: f(packed ref, move status) -> true stops the search
template <class F>
__device__ __forceinline__ void for_each_entrant(const WorldView &W, int cx, int cy, unsigned key, int self, F f) {
    for (int ga = 0; ga < W.G; ga++) {
        const TypeDev TA = W.type[ga];
        const GroupDev A = W.grp[ga];
        for (int k = 0; k < TA.n_move; k++) {
            const int2 d = W.delta[TA.move_off + k];
            if ((d.x | d.y) == 0) continue;
            for (int ax = 0; ax < TA.bw; ax++)
                for (int ay = 0; ay < TA.bl; ay++) {
                    const int px = cx - d.x - ax, py = cy - d.y - ay;
                    if (px < 0 || py < 0 || px >= W.w || py >= W.h) continue;
                    const int e = W.occ[py * W.w + px];
                    if (e < 0 || e == self || ref_group(e) != ga) continue;
                    const int ei = ref_index(e);
                    if (A.x[ei] != px || A.y[ei] != py) continue;          // not that body's top-left cell
                    if (A.pend[ei] != (PEND_MOVE | k) || A.drank_a[ei] < 0 || A.key[ei] >= key) continue;
                    if (f(e, A.mv[ei])) return;
                }
        }
    }
}

// turn_mode: the rectangle a candidate enters depends on the way it faces, so candidates are found by scanning the
// neighbourhood of the cell for bodies (each met once, at its top-left cell): f(packed ref) -> true stops the search
template <class F>
__device__ __forceinline__ void for_each_body_near(const WorldView &W, const GroupDev *gtab, int cx, int cy, F f) {
    const int y0 = max(0, cy - W.reach), y1 = min(W.h - 1, cy + W.reach), x0 = max(0, cx - W.reach), x1 = min(W.w - 1, cx + W.reach);
    for (int py = y0; py <= y1; py++)
        for (int px = x0; px <= x1; px++) {
            const int e = W.occ[py * W.w + px];
            if (e < 0) continue;
            const GroupDev &A = gtab[ref_group(e)];
            const int ei = ref_index(e);
            if (A.x[ei] != px || A.y[ei] != py) continue;                  // not that body's top-left cell
            if (f(e)) return;
        }
}
// every candidate of kind `kind` (PEND_MOVE / PEND_TURN), other than `self`, with a lower key whose target rectangle
// (top-left cell in drank_a, dimensions `transposed` or not with respect to the way it faces now) covers cell (cx, cy)
template <class F>
__device__ __forceinline__ void for_each_candidate_onto(const WorldView &W, const GroupDev *gtab, int cx, int cy, unsigned key, int self,
                                                        int kind, bool transposed, F f) {
    for_each_body_near(W, gtab, cx, cy, [&](int e) {
        if (e == self) return false;
        const GroupDev &A = gtab[ref_group(e)];
        const int ei = ref_index(e);
        const int t = A.drank_a[ei];
        if ((A.pend[ei] & ~PEND_ARG) != kind || t < 0 || A.key[ei] >= key) return false;
        int2 dm = dims_for_dir(W.type[ref_group(e)], A.dir[ei]);
        if (transposed) dm = make_int2(dm.y, dm.x);
        const int ty = t / W.w, tx = t - ty * W.w;
        if (cx < tx || cx >= tx + dm.x || cy < ty || cy >= ty + dm.y) return false;
        return f(e, A.mv[ei]);
    });
}
template <class F>
__device__ __forceinline__ void for_each_mover_onto(const WorldView &W, const GroupDev *gtab, int cx, int cy, unsigned key, int self, F f) {
    if (W.turn_mode) for_each_candidate_onto(W, gtab, cx, cy, key, self, PEND_MOVE, false, f);
    else for_each_entrant(W, cx, cy, key, self, f);
}

// MODE 0: is the move blocked? (stops at the first definite obstacle)   MODE 1: all moves are decided -- who is the
// collide object?   MODE 2 (can_absorb types present): the outcome depends on WHICH agent is met first, so the scan
// stops at the first cell that holds an agent or whose state is still unknown
template <int MODE>
__device__ __forceinline__ MoveProbe move_probe(const WorldView &W, const GroupDev *gtab, int g, int i, int tgt_cell, const unsigned *wanted) {
    const GroupDev G = W.grp[g];
    const TypeDev T = W.type[g];
    const unsigned key = G.key[i];
    const int ny = tgt_cell / W.w, nx = tgt_cell - ny * W.w;
    const int self = ref_pack(g, i);
    MoveProbe r{false, false, -1};
    const int2 fp = body_dims(W, G, T, i);
    for (int bx = 0; bx < fp.x; bx++)
        for (int by = 0; by < fp.y; by++) {
            const int cx = nx + bx, cy = ny + by, c = cy * W.w + cx;
            int occupant = -1;                       // who holds the cell when m's turn comes
            bool unknown = false;
            const int o = W.occ[c];
            if (o == OCC_WALL || o == OCC_FOOD) { r.blocked = true; continue; }   // walls and food block but are never a collide object
            if (o >= 0 && o != self) {
                const GroupDev O = gtab[ref_group(o)];
                const int oi = ref_index(o);
                const int ot = O.drank_a[oi];
                bool gone = false;
                if (ot >= 0 && O.key[oi] < key) {
                    const TypeDev TO = W.type[ref_group(o)];
                    const int2 od = body_dims(W, O, TO, oi);
                    const int oy = ot / W.w, ox = ot - oy * W.w;
                    const bool covers_again = cx >= ox && cx < ox + od.x && cy >= oy && cy < oy + od.y;
                    const unsigned st = O.mv[oi];
                    if (mv_taken(st)) gone = true;                       // taken in by a goal: off the map
                    else if (st == 0) unknown = !covers_again || W.any_absorb;
                    else if (st == MV_OK && !covers_again) gone = true;
                }
                if (!gone) occupant = o;             // (possibly only "maybe": flagged by `unknown`)
            }
            // entrants with lower keys -- only where some OTHER candidate's target rectangle covers the cell at all
            // (wanted[c] counts the candidates whose rectangle covers c; mine is one of them)
            if ((occupant < 0 || unknown) && wanted[c] > 1)
                for_each_mover_onto(W, gtab, cx, cy, key, self, [&](int e, unsigned st) {
                    if (st == MV_OK) { occupant = e; return true; }
                    if (st == 0) unknown = true;
                    return false;
                });
            if (MODE == 2) {
                if (unknown) { r.undecided = true; return r; }
                if (occupant >= 0) { r.blocked = true; r.blocker = occupant; return r; }
                continue;
            }
            if (occupant >= 0 && !unknown) {
                r.blocked = true;
                if (MODE == 1 && r.blocker < 0) r.blocker = occupant;
            } else if (unknown) r.undecided = true;
            if (MODE == 0 && r.blocked) return r;
        }
    return r;
}

// candidates: alive movers with a non-zero delta whose target rectangle is inside the map (Map.cc:455)
// (starve: false when the turn phase of this step has already run starvation -- turn_prep_body)
__device__ __forceinline__ void movg_prep_body(const WorldView &W, int g, int i, unsigned *wanted, int slot, bool starve) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    if (starve) starve_body(W, g, G, T, i, slot);
    if (i >= G.n) return;
    int t = -1;
    const int pend = G.pend[i];
    const int2 fp = body_dims(W, G, T, i);
    if (!G.dead[i] && (pend & ~PEND_ARG) == PEND_MOVE) {
        int2 d = agent_delta(W, G, i, T.move_off, pend & PEND_ARG);
        int nx = G.x[i] + d.x, ny = G.y[i] + d.y;
        if ((d.x | d.y) != 0 && nx >= 0 && ny >= 0 && nx + fp.x < W.w && ny + fp.y < W.h) t = ny * W.w + nx;
    }
    // goals that move themselves are outside the engine's scope (no shipped game gives them actions): reported, not guessed
    if (t >= 0 && T.can_absorb) { W.counters[CTR_UNSUPPORTED] = 1; t = -1; }
    G.drank_a[i] = t;
    G.mv[i] = t >= 0 ? 0u : MV_FAIL;      // 0 = undecided (the packed-dependency encoding of the 1x1 path is not used here)
    if (t >= 0) {
        const int ny = t / W.w, nx = t - ny * W.w;
        for (int by = 0; by < fp.y; by++)
            for (int bx = 0; bx < fp.x; bx++) atomicAdd(&wanted[(ny + by) * W.w + nx + bx], 1u);
    }
}
__global__ void __launch_bounds__(256) k_movg_prep(WorldView W, unsigned *wanted, int starve) {
    if (attack_open(W)) return;
    if ((blockIdx.x | blockIdx.y | threadIdx.x) == 0) W.counters[CTR_CHANGED] = 0;   // move rounds start
    movg_prep_body(W, blockIdx.y, blockIdx.x * blockDim.x + threadIdx.x, wanted, blockIdx.x % DEAD_SLOTS, starve != 0);
}

__device__ __forceinline__ void movg_sweep_body(const WorldView &W, const GroupDev *gtab, int g, int i, const unsigned *wanted, int *flagp) {
    const GroupDev &G = W.grp[g];
    const int t = G.drank_a[i];
    if (t < 0 || G.mv[i] != 0) return;
    if (!W.any_absorb) {
        MoveProbe r = move_probe<0>(W, gtab, g, i, t, wanted);
        if (r.blocked) G.mv[i] = MV_FAIL;
        else if (!r.undecided) G.mv[i] = MV_OK;
        else if (flagp) *flagp = 1;
        return;
    }
    // Map::do_move with goals (Map.cc:334-353): the collide object is the first agent met; a goal that is still free
    // takes the mover in, a taken one is bumped without any effect
    MoveProbe r = move_probe<2>(W, gtab, g, i, t, wanted);
    unsigned st = 0;
    if (!r.undecided) {
        if (!r.blocked) st = MV_OK;
        else if (r.blocker < 0 || !W.type[ref_group(r.blocker)].can_absorb) st = MV_FAIL;
        else {
            const int bg = ref_group(r.blocker), bi = ref_index(r.blocker);
            const GroupDev B = gtab[bg];
            if (B.absorbed[bi]) st = MV_SILENT;
            else {   // free at phase start: is it still free at my turn?  (goals do not move: their cells are static)
                const TypeDev TB = W.type[bg];
                const unsigned key = G.key[i];
                const int self = ref_pack(g, i), goal = r.blocker;
                const int2 gd = body_dims(W, B, TB, bi), md = body_dims(W, G, W.type[g], i);
                bool lost = false, unknown = false;
                for (int bx = 0; bx < gd.x && !lost; bx++)
                    for (int by = 0; by < gd.y && !lost; by++) {
                        const int cx = B.x[bi] + bx, cy = B.y[bi] + by;
                        const int ty = t / W.w, tx = t - ty * W.w;
                        const bool mine = cx >= tx && cx < tx + md.x && cy >= ty && cy < ty + md.y;
                        if (wanted[cy * W.w + cx] <= (mine ? 1u : 0u)) continue;   // no other candidate reaches this cell
                        for_each_mover_onto(W, gtab, cx, cy, key, self, [&](int, unsigned s2) {
                            if (mv_taken(s2) && mv_taken_by(s2) == goal) { lost = true; return true; }
                            if (s2 == 0) unknown = true;
                            return false;
                        });
                    }
                if (lost) st = MV_SILENT;
                else if (!unknown) st = MV_TAKEN_BIT | (unsigned)goal;
            }
        }
    }
    if (st) G.mv[i] = st;
    else if (flagp) *flagp = 1;
}
__global__ void __launch_bounds__(256) k_movg_sweep(WorldView W, const GroupDev *gtab, const unsigned *wanted, int flag) {
    if (attack_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_sweep_body(W, gtab, g, i, wanted, flag >= 0 ? &W.counters[flag] : nullptr);
}

// Map::get_collide for failed moves (Map.cc:334-353, 486-501): first agent met in the target rectangle
__device__ __forceinline__ void movg_collide_body(const WorldView &W, const GroupDev *gtab, int g, int i, const unsigned *wanted) {
    const GroupDev &G = W.grp[g];
    const int t = G.drank_a[i];
    if (t < 0) return;
    const unsigned st = G.mv[i];
    if (st == MV_FAIL) {
        MoveProbe r = move_probe<1>(W, gtab, g, i, t, wanted);
        if (r.blocker >= 0) { G.last_op[i] = OP_COLLIDE; G.op_obj[i] = r.blocker; }
    } else if (mv_taken(st)) {   // exactly one mover per goal ends up here
        const int goal = mv_taken_by(st);
        const GroupDev B = gtab[ref_group(goal)];
        const int bi = ref_index(goal);
        B.absorbed[bi] = 1;
        B.hp[bi] = B.hp[bi] * 2;
        G.dead[i] = 1;
        G.last_op[i] = OP_COLLIDE; G.op_obj[i] = goal;
        atomicAdd(&W.counters[CTR_TAKEN + g], 1);
    }
}
__global__ void __launch_bounds__(256) k_movg_collide(WorldView W, const GroupDev *gtab, const unsigned *wanted) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_collide_body(W, gtab, g, i, wanted);
}

__device__ __forceinline__ void movg_vacate_body(const WorldView &W, int g, int i) {
    const GroupDev &G = W.grp[g];
    if (G.drank_a[i] < 0 || !(G.mv[i] == MV_OK || mv_taken(G.mv[i]))) return;
    const int2 fp = body_dims(W, G, W.type[g], i);
    cells_clear(W, G.x[i], G.y[i], fp.x, fp.y);
}
__global__ void __launch_bounds__(256) k_movg_vacate(WorldView W) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_vacate_body(W, g, i);
}

__device__ __forceinline__ void movg_enter_body(const WorldView &W, int g, int i) {
    const GroupDev &G = W.grp[g];
    const int c = G.drank_a[i];
    if (c < 0 || G.mv[i] != MV_OK) return;
    const int ny = c / W.w, nx = c - ny * W.w;
    const int2 fp = body_dims(W, G, W.type[g], i);
    body_fill(W, nx, ny, fp.x, fp.y, ref_pack(g, i));
    G.x[i] = nx; G.y[i] = ny;
}
__global__ void __launch_bounds__(256) k_movg_enter(WorldView W) {
    if (step_open(W)) return;
    const int g = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < W.grp[g].n) movg_enter_body(W, g, i);
}

// ------------------------------------------------------------------------------------------------ turn, generic bodies
// turn_mode with bodies larger than one cell (GridWorld.cc:544-571, Map::do_turn Map.cc:361-406).  A body turns about its
// reference corner (turn_x/y_offset are 0, AgentType.cc:108: the corner cell stays where it is) and its footprint is
// transposed; the turn happens iff the new rectangle is inside the map and, at the turner's place in the order (stripe lists,
// then the boundary list, each in insertion order -- the same key as for moves), free of everybody but itself.  Like the
// generic move: every cell of the new rectangle outside the own body must be free of its phase-start occupant (which must
// have turned away earlier, successfully, without covering the cell again) and of every lower-key turner that turned onto
// it.  The recursion only looks at lower keys; sweeps decide whoever has its dependencies decided.
// drank_a: top-left cell of the new rectangle (turn candidates), -1 otherwise;  mv: 0 undecided, MV_OK, MV_FAIL.
__device__ __forceinline__ int turned_dir(int dir, int pend) { return (dir + (pend & PEND_ARG) * 2 - 1 + DIR_NUM) % DIR_NUM; }

__device__ __forceinline__ void turn_prep_body(const WorldView &W, int g, int i, unsigned *wanted, int slot) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    starve_body(W, g, G, T, i, slot);        // starvation comes before the turns (GridWorld.cc:519-542)
    if (i >= G.n) return;
    int t = -1;
    const int pend = G.pend[i];
    if (!G.dead[i] && (pend & ~PEND_ARG) == PEND_TURN) {
        const int dir = G.dir[i], ndir = turned_dir(dir, pend);
        int rx, ry, nx, ny;
        saved_to_real(dir, T.bw, T.bl, G.x[i], G.y[i], rx, ry);
        real_to_saved(ndir, T.bw, T.bl, rx, ry, nx, ny);
        const int2 nd = dims_for_dir(T, ndir);
        if (nx >= 0 && ny >= 0 && nx + nd.x < W.w && ny + nd.y < W.h) {   // is_blank_area's bounds (Map.cc:455)
            t = ny * W.w + nx;
            for (int by = 0; by < nd.y; by++)
                for (int bx = 0; bx < nd.x; bx++) atomicAdd(&wanted[(ny + by) * W.w + nx + bx], 1u);
        }
    }
    G.drank_a[i] = t;
    G.mv[i] = t >= 0 ? 0u : MV_FAIL;
}

__device__ __forceinline__ void turn_sweep_body(const WorldView &W, const GroupDev *gtab, int g, int i, const unsigned *wanted, int *flagp) {
    const GroupDev &G = W.grp[g];
    const TypeDev &T = W.type[g];
    const int t = G.drank_a[i];
    if (t < 0 || G.mv[i] != 0) return;
    const unsigned key = G.key[i];
    const int self = ref_pack(g, i);
    const int dir = G.dir[i];
    const int2 od = dims_for_dir(T, dir), nd = make_int2(od.y, od.x);
    const int ox = G.x[i], oy = G.y[i], ny = t / W.w, nx = t - ny * W.w;
    bool blocked = false, unknown = false;
    for (int by = 0; by < nd.y && !blocked; by++)
        for (int bx = 0; bx < nd.x && !blocked; bx++) {
            const int cx = nx + bx, cy = ny + by, c = cy * W.w + cx;
            if (cx >= ox && cx < ox + od.x && cy >= oy && cy < oy + od.y) continue;     // my own body
            const int o = W.occ[c];
            if (o == OCC_WALL || o == OCC_FOOD) { blocked = true; break; }
            if (o >= 0) {                      // the phase-start occupant: gone iff it turned away before me
                const GroupDev &X = gtab[ref_group(o)];
                const int xi = ref_index(o), xt = X.drank_a[xi];
                if (!((X.pend[xi] & ~PEND_ARG) == PEND_TURN && xt >= 0 && X.key[xi] < key)) { blocked = true; break; }
                const unsigned st = X.mv[xi];
                if (st == MV_FAIL) { blocked = true; break; }
                const int2 xd = dims_for_dir(W.type[ref_group(o)], X.dir[xi]);            // (its footprint now; the new one is transposed)
                const int ty = xt / W.w, tx = xt - ty * W.w;
                const bool covers_again = cx >= tx && cx < tx + xd.y && cy >= ty && cy < ty + xd.x;
                if (st == MV_OK) { if (covers_again) { blocked = true; break; } }
                else unknown = true;
                // (a decided turner has ALREADY re-laid its body when the sweeps of a later launch read `dir`: decisions are
                //  only committed after the sweeps converge -- turn_vacate / turn_enter -- so `dir` is the phase-start one here)
            }
            if (wanted[c] > 1)                 // somebody else's new rectangle covers the cell too
                for_each_candidate_onto(W, gtab, cx, cy, key, self, PEND_TURN, true, [&](int e, unsigned st) {
                    if (e == o) return false;                                   // (the occupant was dealt with above)
                    if (st == MV_OK) { blocked = true; return true; }
                    if (st == 0) unknown = true;
                    return false;
                });
        }
    if (blocked) G.mv[i] = MV_FAIL;
    else if (!unknown) G.mv[i] = MV_OK;
    else if (flagp) *flagp = 1;
}

__device__ __forceinline__ void turn_vacate_body(const WorldView &W, int g, int i) {
    const GroupDev &G = W.grp[g];
    if (G.drank_a[i] < 0 || G.mv[i] != MV_OK) return;
    const int2 od = dims_for_dir(W.type[g], G.dir[i]);
    cells_clear(W, G.x[i], G.y[i], od.x, od.y);
}
// (wanted: the counters of the cells this candidate asked for go back to zero -- the move phase uses the same array)
__device__ __forceinline__ void turn_enter_body(const WorldView &W, int g, int i, unsigned *wanted) {
    const GroupDev &G = W.grp[g];
    const int t = G.drank_a[i];
    if (t < 0) return;
    const int2 od = dims_for_dir(W.type[g], G.dir[i]), nd = make_int2(od.y, od.x);
    const int ny = t / W.w, nx = t - ny * W.w;
    if (wanted)
        for (int by = 0; by < nd.y; by++)
            for (int bx = 0; bx < nd.x; bx++) wanted[(ny + by) * W.w + nx + bx] = 0u;
    if (G.mv[i] != MV_OK) return;
    body_fill(W, nx, ny, nd.x, nd.y, ref_pack(g, i));
    G.dir[i] = turned_dir(G.dir[i], G.pend[i]);
    G.x[i] = nx; G.y[i] = ny;
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

// ------------------------------------------------------------------------------------------------ reward rules
// Event(a, op, b) with 'any' symbols: every agent i of group(a), in index order, whose last_op == op and whose
// op_obj is in group(b) triggers the rule once (RewardEngine.cc:373-414).  Receivers that are the subject are added
// by the subject's own thread; receivers that are the object are counted with an int atomic and replayed as `hits`
// sequential float adds of the same value -- order-independent, hence exact.
struct RuleBatch { RuleArgs r[4]; };   // rules that pay different groups and no objects: one launch, blockIdx.y = rule

// (bodies with a wave ballot at the end are called by every lane, i >= n included)
__device__ __forceinline__ void rule_body(const WorldView &W, const RuleArgs &A, int i) {
    const GroupDev &G = W.grp[A.ga];
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
__global__ void __launch_bounds__(256) k_rule(WorldView W, RuleBatch B) {
    if (step_open(W)) return;
    rule_body(W, B.r[blockIdx.y], blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ void rule_obj_body(const WorldView &W, const RuleArgs &A, int i) {
    const GroupDev &G = W.grp[A.gb];
    if (i >= G.n) return;
    int h = G.hits[i];
    if (!h) return;
    float nr = G.next_reward[i];
    for (; h > 0; h--) for (int k = 0; k < A.n_obj; k++) nr += A.v_obj[k];
    G.next_reward[i] = nr;
    G.hits[i] = 0;
}
__global__ void __launch_bounds__(256) k_rule_obj(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    rule_obj_body(W, A, blockIdx.x * blockDim.x + threadIdx.x);
}

// General single-iterator rule (launch.h RuleProg): agent i of group ga is bound to x; if the expression has a second
// symbol y, it is bound to i's op_obj, and i is skipped when it has none or one of another group
// (RewardEngine.cc:246-262).  The expression is evaluated on that binding; receivers: x adds in place, y is counted
// and replayed by k_rule_obj.
__device__ __forceinline__ void rule_prog_body(const WorldView &W, const GroupDev *gtab, const RuleProg &P, int i) {
    const GroupDev &G = W.grp[P.ga];
    bool trig = false;
    if (i < G.n) {
        int ent[2] = {ref_pack(P.ga, i), -1};
        bool bound = true;
        if (P.has_obj) {
            const int o = G.op_obj[i];
            bound = o >= 0 && ref_group(o) == P.gb;
            ent[1] = o;
        }
        if (bound) {
            unsigned stack = 0;                       // bit k = value k of the evaluation stack
            int sp = 0;
            for (int k = 0; k < P.n; k++) {
                const int op = P.op[k];
                bool v;
                if (op == 0 || op == 1) {              // and / or
                    const bool b1 = (stack >> (sp - 1)) & 1u, b0 = (stack >> (sp - 2)) & 1u;
                    sp -= 2;
                    v = op == 0 ? (b0 && b1) : (b0 || b1);
                } else if (op == 2) {                  // not
                    sp -= 1;
                    v = !((stack >> sp) & 1u);
                } else {
                    const int e = ent[P.a[k][0]];
                    const GroupDev S = gtab[ref_group(e)];
                    const int si = ref_index(e);
                    if (op == 8) v = S.dead[si] != 0;                                                  // die
                    else if (op == 4) v = S.x[si] == P.a[k][1] && S.y[si] == P.a[k][2];                // at
                    else if (op == 5) v = S.x[si] > P.a[k][1] && S.x[si] < P.a[k][3] && S.y[si] > P.a[k][2] && S.y[si] < P.a[k][4];   // in
                    else v = S.last_op[si] == op && S.op_obj[si] == ent[P.a[k][1]];                    // kill / collide / attack
                }
                stack = (stack & ~(1u << sp)) | ((v ? 1u : 0u) << sp);
                sp++;
            }
            if (stack & 1u) {
                trig = true;
                if (P.n_subj) {
                    float nr = G.next_reward[i];
                    for (int k = 0; k < P.n_subj; k++) nr += P.v_subj[k];
                    G.next_reward[i] = nr;
                }
                if (P.n_obj) atomicAdd(&gtab[P.gb].hits[ref_index(ent[1])], 1);
            }
        }
    }
    if (__ballot(trig) && lane_id() == 0) W.counters[CTR_TRIGGER + P.rule_no] = 1;
}
__global__ void __launch_bounds__(256) k_rule_prog(WorldView W, const GroupDev *gtab, RuleProg P) {
    if (step_open(W)) return;
    rule_prog_body(W, gtab, P, blockIdx.x * blockDim.x + threadIdx.x);
}

// Event(x, op, c) & Event(y, op_y, c): the reference's search (RewardEngine.cc:216-306) binds x over its group, then y
// over its group skipping the agent bound to x, re-binds c to y's target, and pays the receivers once per ordered
// pair (i, j) with  last_op[i] == op, last_op[j] == op_y, op_obj[i] == op_obj[j] in c's group.  Per agent t that is
//   [v_y x #{partners i < t}] [v_x x #{partners j}] [v_y x #{partners i > t}]
// when x and y share a group and predicate (t plays both parts; pairs come in (i, j) order), and a run of one value
// otherwise.  The partners are found through a per-target list: head in the target's `hits`, links in `mv`.
__device__ __forceinline__ int pair_roles(const WorldView &W, const RuleArgs &A, int g, int i) {
    const GroupDev &G = W.grp[g];
    int o = G.op_obj[i];
    if (o < 0 || ref_group(o) != A.gb) return 0;
    int op = G.last_op[i];
    return ((g == A.ga && op == A.op) ? 1 : 0) | ((g == A.gy && op == A.op_y) ? 2 : 0);
}

__device__ __forceinline__ void pair_link_body(const WorldView &W, const RuleArgs &A, int g, int i) {
    const GroupDev &G = W.grp[g];
    if (i >= G.n || !pair_roles(W, A, g, i)) return;
    G.mv[i] = (unsigned)atomicExch(&W.grp[A.gb].hits[ref_index(G.op_obj[i])], ref_pack(g, i) + 1);
}
__global__ void __launch_bounds__(256) k_pair_link(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    pair_link_body(W, A, blockIdx.y ? A.gy : A.ga, blockIdx.x * blockDim.x + threadIdx.x);
}

__device__ __forceinline__ void pair_pay_body(const WorldView &W, const RuleArgs &A, int g, int i) {
    const GroupDev &G = W.grp[g];
    bool trig = false;
    if (i < G.n) {
        const int mine = pair_roles(W, A, g, i);
        if (mine) {
            int y_lt = 0, y_gt = 0, n_x = 0;   // pairs in which this agent is y (partner before / after it), is x
            for (int r = W.grp[A.gb].hits[ref_index(G.op_obj[i])]; r != 0;) {
                const int ug = ref_group(r - 1), ui = ref_index(r - 1);
                r = (int)W.grp[ug].mv[ui];
                if (ug == g && ui == i) continue;
                const int theirs = pair_roles(W, A, ug, ui);
                if ((mine & 1) && (theirs & 2)) n_x++;
                if ((mine & 2) && (theirs & 1)) { if (ug == g && ui > i) y_gt++; else y_lt++; }
            }
            if (n_x | y_lt | y_gt) {
                trig = true;
                float nr = G.next_reward[i];
                for (; y_lt > 0; y_lt--) for (int k = 0; k < A.n_y; k++) nr += A.v_y[k];
                for (; n_x > 0; n_x--) for (int k = 0; k < A.n_subj; k++) nr += A.v_subj[k];
                for (; y_gt > 0; y_gt--) for (int k = 0; k < A.n_y; k++) nr += A.v_y[k];
                G.next_reward[i] = nr;
            }
        }
    }
    if (__ballot(trig) && lane_id() == 0) W.counters[CTR_TRIGGER + A.rule_no] = 1;
}
__global__ void __launch_bounds__(256) k_pair_pay(WorldView W, RuleArgs A) {
    if (step_open(W)) return;
    pair_pay_body(W, A, blockIdx.y ? A.gy : A.ga, blockIdx.x * blockDim.x + threadIdx.x);
}

// the object's share (one run of v_obj per ordered pair) and the reset of the list heads
__device__ __forceinline__ void pair_obj_body(const WorldView &W, const RuleArgs &A, int i) {
    const GroupDev &G = W.grp[A.gb];
    if (i >= G.n) return;
    int r = G.hits[i];
    if (!r) return;
    G.hits[i] = 0;
    if (!A.n_obj) return;
    int nx = 0, ny = 0, nboth = 0;
    while (r != 0) {
        const int ug = ref_group(r - 1), ui = ref_index(r - 1);
        const int roles = pair_roles(W, A, ug, ui);
        nx += roles & 1; ny += (roles >> 1) & 1; nboth += roles == 3;
        r = (int)W.grp[ug].mv[ui];
    }
    int pairs = nx * ny - nboth;
    if (!pairs) return;
    float nr = G.next_reward[i];
    for (; pairs > 0; pairs--) for (int k = 0; k < A.n_obj; k++) nr += A.v_obj[k];
    G.next_reward[i] = nr;
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

// ... then stable compaction into the alternate buffers + init_reward + re-index the map (groups with deaths), or
// Agent::init_reward alone (groups without)
// (M.vh > 0: the minimap of the NEXT observations rides along -- every block adds the survivors it handles to an LDS histogram of
// their minimap cells and flushes it with one global atomic per non-empty bin; k_clear_finish / k_mini_norm divide.  That is
// k_minimap + k_minimap_norm, two launches per cycle, gone: the positions pass through this kernel anyway)
__global__ void __launch_bounds__(SCAN_THREADS) k_clear_compact(WorldView W, ClearArgs A, const int *sums, MiniArgs M, int *counts) {
    extern __shared__ int s_hist[];
    const int g = blockIdx.y;
    const GroupDev G = W.grp[g];
    if ((int)(blockIdx.x * SCAN_TILE) >= G.n) return;
    const float step_reward = W.type[g].step_reward;
    const int VHW = M.vh * M.vw;
    if (VHW > 0) {
        for (int k = threadIdx.x; k < VHW; k += SCAN_THREADS) s_hist[k] = 0;
        __syncthreads();
    }
    if (A.mode[g] == 1) {
        for (int k = 0; k < SCAN_ITEMS; k++) {
            const int i = blockIdx.x * SCAN_TILE + k * SCAN_THREADS + threadIdx.x;
            if (i < G.n) {
                G.last_reward[i] = G.next_reward[i]; G.next_reward[i] = step_reward; G.last_op[i] = OP_NULL; G.op_obj[i] = -1;
                if (VHW > 0) atomicAdd(&s_hist[(G.y[i] / M.scale_h) * M.vw + G.x[i] / M.scale_w], 1);
            }
        }
    } else if (A.mode[g] == 2) {
        const ClearArgs::Alt D = A.dst[g];
        const int bw = W.type[g].bw, bl = W.type[g].bl;
        // (the single-buffered state goes back to its rest values at every agent's OWN index -- all that matters are the positions below
        // the new size, and each is some thread's own; `dead` is read by that thread alone in this launch: no second pass for it)
        block_rank([&](int i) {
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
                   G.n, block_prefix(sums + A.sums_off[g], blockIdx.x));
    }
    if (VHW > 0) {
        __syncthreads();
        for (int k = threadIdx.x; k < VHW; k += SCAN_THREADS)
            if (s_hist[k]) atomicAdd(&counts[((blockIdx.x % MINI_COPIES) * W.G + g) * VHW + k], s_hist[k]);   // MINI_COPIES histograms: same-address atomics serialise
    }
}

// mini[j][cell] = float(count) / float(n_j) exactly as the reference divides (k_minimap_norm), from the histogram k_clear_compact
// left; the histogram goes back to zero
__device__ __forceinline__ void mini_norm_body(const WorldView &Wn, const MiniArgs &M, int *counts, int k) {
    const int VHW = M.vh * M.vw;
    if (k >= Wn.G * VHW) return;
    const int tot = Wn.grp[k / VHW].n;
    int cnt = 0;
    for (int c = 0; c < MINI_COPIES; c++) { cnt += counts[c * Wn.G * VHW + k]; counts[c * Wn.G * VHW + k] = 0; }
    M.out[k] = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(cnt, 1 << 24), (float)(unsigned)tot);
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

// The minimap of the next observations in one workgroup: an LDS histogram of every group (s_hist: [NG][VHW] counts, then
// [NG] agents left out), then count / total exactly as the reference divides (k_minimap; GridWorld.cc:331-360)
__device__ __forceinline__ void minimap_one_workgroup(const GroupDev *grp, int NG, const MiniArgs &M, int *s_hist, int nthreads) {
    const int tid = threadIdx.x, VHW = M.vh * M.vw;
    for (int k = tid; k < NG * VHW + NG; k += nthreads) s_hist[k] = 0;
    __syncthreads();
    for (int g = 0; g < NG; g++) {
        const GroupDev &G = grp[g];
        for (int i = tid; i < G.n; i += nthreads) {
            if (M.skip && G.absorbed[i]) { atomicAdd(&s_hist[NG * VHW + g], 1); continue; }
            atomicAdd(&s_hist[g * VHW + (G.y[i] / M.scale_h) * M.vw + G.x[i] / M.scale_w], 1);
        }
    }
    __syncthreads();
    for (int k = tid; k < NG * VHW; k += nthreads) {
        const int g = k / VHW;
        const int tot = grp[g].n - (M.skip ? s_hist[NG * VHW + g] : 0);
        M.out[k] = tot == 0 ? __int_as_float(0xFFC00000) : __fdiv_rn((float)min(s_hist[k], 1 << 24), (float)(unsigned)tot);
    }
}

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
        SOLO_EACH(g, i) attack_rank_body(W, g, i, S.rank, S.hit, listed ? S.slink : nullptr, &s_ntgt);
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
            S.rec->triggers = mask;
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
    if (W.vc_packed) hipLaunchKernelGGL(k_paint<true>, dim3(blocks), dim3(256), 0, s, W, gtab, ttab);
    else hipLaunchKernelGGL(k_paint<false>, dim3(blocks), dim3(256), 0, s, W, gtab, ttab);
}

void launch_minimap(hipStream_t s, const WorldView &W, const RenderArgs &R, int *counts, float *mini) {
    int VHW = R.VH * R.VW;
    const int skip = W.type[R.g].can_absorb;
    int *left_out = counts;      // the first MAXG ints of the buffer; the histogram follows
    counts += MAXG;
    if (skip) (void)hipMemsetAsync(left_out, 0, sizeof(int) * MAXG, s);
    int mx = 1;
    for (int g = 0; g < W.G; g++) mx = W.grp[g].n > mx ? W.grp[g].n : mx;
    int bx = (mx + 255) / 256;
    if (bx > 128) bx = 128;      // every block ends with one global atomic per non-empty bin: few, fat blocks
    // (folding the normalisation into the histogram's last block -- ticket counter -- was measured: +12 us, the tickets serialise)
    hipLaunchKernelGGL(k_minimap, dim3(bx, W.G), dim3(256), VHW * sizeof(int), s, W, R, counts, left_out, skip);
    hipLaunchKernelGGL(k_minimap_norm, dim3((W.G * VHW + 255) / 256), dim3(256), 0, s, R, W.G, counts, mini, left_out, skip);
}

// returns the kernel taken: 0 k_render / k_render_cells16 (every game), 1 k_render_fast, 4 k_render_sweep2
int launch_render(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4, bool nt) {
    if (R.n <= 0) return 0;
    size_t lds = (size_t)RENDER_WAVES * P.strip_floats * sizeof(float);
    dim3 grid(P.spans + P.feat_blocks), block(64 * RENDER_WAVES);
    const bool packed = W.vc_packed != 0;   // must match launch_paint
    if (render_fast_ok(W, R, P, vec4 && nt)) {
        // MAGENT_RENDER_FAST: 0 generic kernels | 1 k_render_fast | 4 k_render_sweep2 | unset: bf16 cells -> 1; float32 -> 4 at scale, else generic
        static const int forced = std::getenv("MAGENT_RENDER_FAST") ? std::atoi(std::getenv("MAGENT_RENDER_FAST")) : -1;
        static const int sweep_fixed = std::getenv("MAGENT_RENDER_SWEEP") ? std::atoi(std::getenv("MAGENT_RENDER_SWEEP")) : 0;   // (tests: few workgroups, many rounds)
        static const int su_env = std::getenv("MAGENT_RENDER_SU") ? std::atoi(std::getenv("MAGENT_RENDER_SU")) : 2;
        static const int dv_env = std::getenv("MAGENT_RENDER_DEPTH") ? std::atoi(std::getenv("MAGENT_RENDER_DEPTH")) : 2;
        const long long steps = ((long long)R.n * R.VH * R.VW + 63) / 64;
        const int VHW = R.VH * R.VW;
        int mode = forced;
        if (mode < 0) mode = R.cells16 ? 1 : (steps >= 256ll * RENDER_WAVES * 2 * 8 ? 4 : 0);   // (a sweep wants >= 8 rounds of 256 workgroups)
        if (mode == 4) {
            const int SUv = su_env >= 3 ? 3 : su_env >= 2 ? 2 : 1;
            const int sweep = (int)std::min<long long>(sweep_fixed > 0 ? sweep_fixed : 256, (steps + RENDER_WAVES * SUv - 1) / (RENDER_WAVES * SUv));
            dim3 sgrid(sweep + P.feat_blocks);
            static const size_t pad = std::getenv("MAGENT_RENDER_PAD") ? (size_t)std::atoi(std::getenv("MAGENT_RENDER_PAD")) : 0;   // (tuning: extra LDS caps workgroups per CU)
            static bool allowed = false;
            if (pad && !allowed) {
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(k_render_sweep2<false, 2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
                allowed = true;
            }
            const size_t sl = (size_t)RENDER_WAVES * SUv * 64 * 7 * sizeof(float) + (size_t)VHW * sizeof(RenderFastPos) + pad;
            static const bool xcd_slots = std::getenv("MAGENT_RENDER_XCD") && std::atoi(std::getenv("MAGENT_RENDER_XCD")) != 0;   // (tuning)
            RenderPlan Ps = P;
            if (xcd_slots) Ps.xcd_chunk = -1;
#define SW2(C16, DVV, SUV) hipLaunchKernelGGL((k_render_sweep2<C16, DVV, SUV>), sgrid, block, sl, s, render_world(W, R.g), R, Ps, sweep)
#define SW2D(C16, SUV) do { if (dv_env <= 1) SW2(C16, 1, SUV); else if (dv_env == 2) SW2(C16, 2, SUV); else SW2(C16, 3, SUV); } while (0)
            if (R.cells16) { if (SUv == 3) SW2D(true, 3); else if (SUv == 2) SW2D(true, 2); else SW2D(true, 1); }
            else { if (SUv == 3) SW2D(false, 3); else if (SUv == 2) SW2D(false, 2); else SW2D(false, 1); }
#undef SW2D
#undef SW2
            return 4;
        }
        if (mode == 1) {
            const size_t fl = render_fast_lds(VHW);
            if (R.cells16) hipLaunchKernelGGL((k_render_fast<true>), grid, block, fl, s, render_world(W, R.g), R, P);
            else hipLaunchKernelGGL((k_render_fast<false>), grid, block, fl, s, render_world(W, R.g), R, P);
            return 1;
        }
    }
    if (R.cells16) {
        if (R.turn) { if (packed) hipLaunchKernelGGL((k_render_cells16<true, true>), grid, block, lds, s, W, R, P); else hipLaunchKernelGGL((k_render_cells16<false, true>), grid, block, lds, s, W, R, P); }
        else { if (packed) hipLaunchKernelGGL((k_render_cells16<true, false>), grid, block, lds, s, W, R, P); else hipLaunchKernelGGL((k_render_cells16<false, false>), grid, block, lds, s, W, R, P); }
        return 0;
    }
#define RENDER_LAUNCH(V, N, UU, PK) hipLaunchKernelGGL((k_render<V, N, UU, PK, false>), grid, block, lds, s, W, R, P)
#define RENDER_PK(V, N, UU) do { if (packed) RENDER_LAUNCH(V, N, UU, true); else RENDER_LAUNCH(V, N, UU, false); } while (0)
    if (R.turn) {      // turn_mode: one step per wave iteration, scalar or 16-byte stores
        if (vec4 && packed) hipLaunchKernelGGL((k_render<true, true, 1, true, true>), grid, block, lds, s, W, R, P);
        else if (vec4) hipLaunchKernelGGL((k_render<true, true, 1, false, true>), grid, block, lds, s, W, R, P);
        else if (packed) hipLaunchKernelGGL((k_render<false, false, 1, true, true>), grid, block, lds, s, W, R, P);
        else hipLaunchKernelGGL((k_render<false, false, 1, false, true>), grid, block, lds, s, W, R, P);
    } else if (!vec4) RENDER_PK(false, false, 1);
    else if (P.unroll == 2) { if (nt) RENDER_PK(true, true, 2); else RENDER_PK(true, false, 2); }
    else if (P.unroll == 4) { if (nt) RENDER_PK(true, true, 4); else RENDER_PK(true, false, 4); }
    else { if (nt) RENDER_PK(true, true, 1); else RENDER_PK(true, false, 1); }
#undef RENDER_PK
#undef RENDER_LAUNCH
    return 0;
}

void launch_render_multi(hipStream_t s, const WorldView &W, const RenderMulti &M) {
    size_t lds = 0;
    int mx = 0;
    for (int k = 0; k < M.n; k++) { lds = std::max(lds, (size_t)RENDER_WAVES * M.P[k].strip_floats * sizeof(float)); mx = std::max(mx, M.blocks[k]); }
    if (M.n <= 0 || mx <= 0) return;
    dim3 grid(mx, M.n), block(64 * RENDER_WAVES);
    if (W.vc_packed) hipLaunchKernelGGL((k_render_multi<true>), grid, block, lds, s, W, M);
    else hipLaunchKernelGGL((k_render_multi<false>), grid, block, lds, s, W, M);
}

void launch_commit_action(hipStream_t s, const GroupDev &G, const TypeDev &T) {
    if (G.n > 0) hipLaunchKernelGGL(k_commit_action, dim3((G.n + 255) / 256), dim3(256), 0, s, G, T);
}

void launch_features(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4) {
    if (R.n <= 0) return;
    unsigned total = (unsigned)R.n * (unsigned)R.F;
    int fb = (int)std::min<unsigned>((total / 4 + 255) / 256 + 1, 16384);   // ~1 float4 per thread: latency-bound gathers
    if (vec4) hipLaunchKernelGGL((k_features<true>), dim3(fb), dim3(256), 0, s, W, R, P);
    else hipLaunchKernelGGL((k_features<false>), dim3(fb), dim3(256), 0, s, W, R, P);
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
// ================================================================================================ repeated set_action: the literal loop
// GridWorld::set_action APPENDS to the step's action lists (GridWorld.cc:403-454): a group that is given actions twice before a step
// has every agent act twice -- two entries in the shuffled attack list, two moves in list order, the second from wherever the first
// one ended.  The parallel phases above rest on "one pending action per agent"; no caller of the reference does this, so the case
// is served by the reference's own sequential loops on ONE lane of the device, exact by construction and slow (about a microsecond
// per list entry).  One-cell bodies without turn_mode, food_mode and goals; everything else still refuses.
//   attack loop GridWorld.cc:464-507 (Map::get_attack_obj Map.cc:209-252, Map::do_attack Map.cc:255-310, Agent::be_attack
//   GridWorld.h:203-209), starve GridWorld.cc:519-542, moves GridWorld.cc:574-613 (Map::do_move Map.cc:313-358)
__global__ void __launch_bounds__(64) k_step_serial(WorldView W, const SerialCall *calls, int n_calls, int2 *alist, int4 *mlist, int4 *msorted, int n_sep) {
    if (threadIdx.x != 0) return;
    const int bandwidth = W.bandwidth;
    int A = 0, M = 0;
    // ---- the lists, in call order (Agent::set_action stores last_action at once: the last call wins)
    for (int c = 0; c < n_calls; c++) {
        const int g = calls[c].g;
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        const int *act = calls[c].actions;
        for (int i = 0; i < G.n; i++) {
            const int a = act[i];
            if (a < 0 || a >= T.n_move + T.n_attack) { W.counters[CTR_BAD_ACTION] = 1; continue; }
            G.last_action[i] = a;
            if (a < T.n_move) {
                int list = n_sep;                                            // the boundary list runs last
                if (W.large_map) { const int x_ = G.x[i] % bandwidth; if (!(x_ < 4 || x_ > bandwidth - 4)) list = G.x[i] / bandwidth; }
                mlist[M++] = make_int4(ref_pack(g, i), a, list, 0);
            } else alist[A++] = make_int2(ref_pack(g, i), a - T.n_move);
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
        if (G.dead[i]) continue;
        const int2 d = W.delta[T.attack_off + k];
        const int tx = G.x[i] + d.x, ty = G.y[i] + d.y;
        int o = OCC_EMPTY;
        if (tx >= 0 && tx < W.w && ty >= 0 && ty < W.h) o = W.occ[ty * W.w + tx];
        if (o < 0 || (!T.attack_in_group && ref_group(o) == g)) { G.next_reward[i] += T.attack_penalty; continue; }
        const int tg = ref_group(o), ti = ref_index(o);
        const GroupDev &V = W.grp[tg];
        const TypeDev &TV = W.type[tg];
        float reward = 0.0f;
        V.hp[ti] -= T.damage;
        if (V.hp[ti] < 0.0f) { V.dead[ti] = 1; V.next_reward[ti] = TV.dead_penalty; }
        if (V.dead[ti]) {
            G.last_op[i] = OP_KILL; G.op_obj[i] = o;
            W.occ[V.y[ti] * W.w + V.x[ti]] = OCC_EMPTY;
            W.counters[dead_slot(tg, 0)] += 1;
            G.hp[i] = fminf(T.hp, G.hp[i] + TV.kill_supply);
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
                    W.occ[G.y[i] * W.w + G.x[i]] = OCC_EMPTY;
                    W.counters[dead_slot(g, 0)] += 1;
                }
            }
        }
    }
    // ---- moves: stripe lists 0 .. n_sep - 1, then the boundary list, each in insertion order (a stable counting sort by list)
    int start[40];
    for (int l = 0; l <= n_sep; l++) start[l] = 0;
    for (int e = 0; e < M; e++) start[mlist[e].z]++;
    for (int l = 0, run = 0; l <= n_sep; l++) { const int c = start[l]; start[l] = run; run += c; }
    for (int e = 0; e < M; e++) msorted[start[mlist[e].z]++] = mlist[e];
    for (int e = 0; e < M; e++) {
        const int g = ref_group(msorted[e].x), i = ref_index(msorted[e].x);
        const GroupDev &G = W.grp[g];
        const TypeDev &T = W.type[g];
        if (G.dead[i]) continue;
        const int2 d = W.delta[T.move_off + msorted[e].y];
        const int nx = G.x[i] + d.x, ny = G.y[i] + d.y;
        if (nx < 0 || ny < 0 || nx + 1 >= W.w || ny + 1 >= W.h) continue;     // Map::is_blank_area's bounds; no collide object out there
        const int c = ny * W.w + nx;
        const int o = W.occ[c];
        if (o == OCC_EMPTY || o == ref_pack(g, i)) {
            W.occ[G.y[i] * W.w + G.x[i]] = OCC_EMPTY;
            W.occ[c] = ref_pack(g, i);
            G.x[i] = nx; G.y[i] = ny;
        } else if (o >= 0) { G.last_op[i] = OP_COLLIDE; G.op_obj[i] = o; }    // Map::get_collide: agents only (walls are no objects)
    }
    // ---- the step's pending actions are consumed
    for (int g = 0; g < W.G; g++) for (int i = 0; i < W.grp[g].n; i++) W.grp[g].pend[i] = PEND_NONE;
}
// the actions a group's pending actions came from (the first call of a step, when a second one follows)
__global__ void __launch_bounds__(256) k_pend_to_actions(GroupDev G, TypeDev T, int *out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < G.n) out[i] = G.pend[i] == PEND_NONE ? T.n_move + T.n_turn + T.n_attack : pend_action(G.pend[i], T);   // (an action outside the space stays one)
}
void launch_step_serial(hipStream_t s, const WorldView &W, const SerialCall *calls, int n_calls, int2 *alist, int4 *mlist, int4 *msorted, int n_sep) {
    hipLaunchKernelGGL(k_step_serial, dim3(1), dim3(64), 0, s, W, calls, n_calls, alist, mlist, msorted, n_sep);
}
void launch_pend_to_actions(hipStream_t s, const GroupDev &G, const TypeDev &T, int *out) {
    if (G.n > 0) hipLaunchKernelGGL(k_pend_to_actions, dim3((G.n + 255) / 256), dim3(256), 0, s, G, T, out);
}

void launch_step_report(hipStream_t s, int *counters, StepRecord *rec, int seq, int NG) {
    hipLaunchKernelGGL(k_step_report, dim3(1), dim3(64), 0, s, counters, rec, seq, NG);
}
void launch_step_reset(hipStream_t s, int *counters) { hipLaunchKernelGGL(k_step_reset, dim3(1), dim3(64), 0, s, counters); }
void launch_set_rng(hipStream_t s, int *counters, unsigned x) { hipLaunchKernelGGL(k_set_rng, dim3(1), dim3(64), 0, s, counters, x); }
void launch_set_counter(hipStream_t s, int *counters, int index, int value, int unless_index) {
    hipLaunchKernelGGL(k_set_counter, dim3(1), dim3(64), 0, s, counters, index, value, unless_index);
}

// (the hit bits have an array of their own, WorldView::hitbits -- until round 4 they shared the move phase's claim words)
void launch_attack_rank(hipStream_t s, const WorldView &W, const int *rank, const ShuffleBufs &B, bool clear_hitbits, const int *sums, const int *wpre,
                        const SeqPlan &P) {
    if (clear_hitbits) (void)hipMemsetAsync(W.hitbits, 0, sizeof(unsigned) * (size_t)W.w * W.h, s);   // (else k_shuffle_draw did it, or the fused step keeps them zero)
    hipLaunchKernelGGL(k_attack_rank, grid_all(W, 256), dim3(256), 0, s, W, rank, W.hitbits, B.head, B.first, sums, wpre, P);
}
static int att_threads(int kmax) {
    static const int forced = getenv("MAGENT_ATT_THREADS") ? atoi(getenv("MAGENT_ATT_THREADS")) : 0;
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
// the step of plain games behind the shuffle: k_plain_rank, rounds of k_plain_eval, then k_strike, k_plain_init, k_plain_commit
// (launch_plain_tail).  `rules`: the compiled rules, if every one of them pays the attacker of one event only (fused_rules); else
// null, and launch_rules runs behind the commit as usual
bool fused_rules(const RuleArgs *rules, int n) {
    if (n > 4) return false;
    // (attack-phase events only: `collide` is decided by the move phase, behind k_strike)
    for (int k = 0; k < n; k++) if (rules[k].pair || rules[k].prog >= 0 || rules[k].n_obj || (rules[k].op != OP_ATTACK && rules[k].op != OP_KILL)) return false;
    return true;
}
void launch_plain_rank(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const int *rank, const ShuffleBufs &B, const int *sums,
                       const int *wpre, const SeqPlan &P) {
    hipLaunchKernelGGL(k_plain_rank, grid_all(W, 256), dim3(256), 0, s, W, PW, ptab, rank, B.head, B.first, sums, wpre, P);
}
size_t plain_eval_lds(int kmax) { return (size_t)kmax * 256 * 8; }
bool plain_eval_lds_ok(int kmax) {
    if (plain_eval_lds(kmax) <= (48u << 10)) return true;
    return hipFuncSetAttribute(reinterpret_cast<const void *>(k_plain_eval), hipFuncAttributeMaxDynamicSharedMemorySize, (int)plain_eval_lds(kmax)) == hipSuccess;
}
void launch_plain_eval(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag) {
    hipLaunchKernelGGL(k_plain_eval, grid_all(W, 256), dim3(256), plain_eval_lds(PW.kmax), s, W, PW, ptab, gtab, ttab, round, flag);
}
void launch_plain_tail(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab,
                       const RuleArgs *rules, int n_rules) {
    StrikeRules R{};
    if (rules) {
        R.n = n_rules;
        for (int k = 0; k < n_rules; k++) {
            R.r[k].ga = rules[k].ga; R.r[k].gb = rules[k].gb; R.r[k].op = rules[k].op; R.r[k].rule_no = rules[k].rule_no; R.r[k].n_subj = rules[k].n_subj;
            for (int q = 0; q < 4; q++) R.r[k].v[q] = rules[k].v_subj[q];
        }
    }
    dim3 g = grid_all(W, 256);
    hipLaunchKernelGGL(k_strike, g, dim3(256), 0, s, W, PW, ptab, gtab, ttab, R);
    hipLaunchKernelGGL(k_plain_init, g, dim3(256), 0, s, W, PW);
    hipLaunchKernelGGL(k_plain_commit, g, dim3(256), 0, s, W, PW, ptab);
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
void launch_clear_compact(hipStream_t s, const WorldView &W, const ClearArgs &A, int *sums, const MiniArgs &M, int *counts) {
    int mx = 1;
    bool any = false;
    for (int g = 0; g < W.G; g++) { mx = std::max(mx, W.grp[g].n); any |= A.mode[g] == 2; }
    dim3 grid((mx + SCAN_TILE - 1) / SCAN_TILE, W.G);
    if (any) hipLaunchKernelGGL(k_clear_count, grid, dim3(SCAN_THREADS), 0, s, W, A, sums);
    hipLaunchKernelGGL(k_clear_compact, grid, dim3(SCAN_THREADS), sizeof(int) * (size_t)M.vh * M.vw, s, W, A, sums, M, counts);
}
void launch_mini_norm(hipStream_t s, const WorldView &Wn, const MiniArgs &M, int *counts) {
    hipLaunchKernelGGL(k_mini_norm, dim3((Wn.G * M.vh * M.vw + 255) / 256), dim3(256), 0, s, Wn, M, counts);
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
size_t render_strip_lds(const RenderPlan &P) { return (size_t)RENDER_WAVES * P.strip_floats * sizeof(float); }
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
void launch_clear_finish(hipStream_t s, const WorldView &Wn, const ClearArgs &A, GroupDev *gtab, TypeDev *ttab, const MiniArgs &M, int *counts) {
    const int blocks = std::max(1, (Wn.G * M.vh * M.vw + 255) / 256);
    hipLaunchKernelGGL(k_clear_finish, dim3(blocks), dim3(256), 0, s, Wn, A, gtab, ttab, M, counts);
}

}  // namespace magent_amd
