// kernels_dev.h -- the device-side bodies shared by the three kernel translation units (render.hip, step.hip, cycle.hip): small helpers,
// the painted-map stores, the render's per-cell machinery, the block scans, and every phase of GridWorld::step as a `*_body` device
// function -- ONE implementation, called by the multi-launch kernels of step.hip and by the one-launch step of cycle.hip alike.
// Nothing here is a kernel (each __global__ lives in exactly one .hip) or keeps state.
//
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
#pragma once
#include "engine.h"
#include "launch.h"
#include "tune.h"
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
// A pointer that a kernel reads from MEMORY -- a device table's, an item's of a batch (pipe.hip) -- is a GENERIC pointer to the compiler, and what
// goes through it a FLAT instruction: counted on the LDS counter as well as on the memory counter, so that a wait for an LDS read or a scalar
// load also waits for every store in flight (the sweeping render's whole design is that it does not: with FLAT stores its batched form ran at
// half its speed, profiles/r06_summary.md), and without the scalar-base addressing of the GLOBAL instructions.  Pointers in kernel arguments
// are known to be global; glob() says the same of one that is not (all of them are device or pinned host memory -- never LDS, never scratch).
// (A cast to the global address space and straight back is folded away before it tells anybody anything; the way through an integer is not.)
template <class T> __device__ __forceinline__ T *glob(T *p) { return (T *)(__attribute__((address_space(1))) T *)(unsigned long long)p; }
__device__ __forceinline__ GroupDev glob_group(GroupDev G) {
    G.x = glob(G.x); G.y = glob(G.y); G.id = glob(G.id); G.last_action = glob(G.last_action); G.op_obj = glob(G.op_obj); G.pend = glob(G.pend);
    G.hp = glob(G.hp); G.next_reward = glob(G.next_reward); G.last_reward = glob(G.last_reward); G.dead = glob(G.dead); G.last_op = glob(G.last_op);
    G.absorbed = glob(G.absorbed); G.dir = glob(G.dir); G.key = glob(G.key); G.drank_a = glob(G.drank_a); G.drank_b = glob(G.drank_b); G.mv = glob(G.mv);
    G.hitf = glob(G.hitf); G.hits = glob(G.hits); G.eat = glob(G.eat); G.fleft = glob(G.fleft); G.fcell = glob(G.fcell);
    return G;
}
__device__ __forceinline__ bool attack_open(const WorldView &W) { return glob(W.counters)[CTR_OPEN_ATTACK] != 0; }
__device__ __forceinline__ bool step_open(const WorldView &W) { return (W.counters[CTR_OPEN_ATTACK] | W.counters[CTR_OPEN_MOVE]) != 0; }

// ------------------------------------------------------------------------------------------------ paint
// viewcell[c] = {group | EMPTY | WALL, bits(hp / type.hp)}: one pass over the map, coalesced 4 B in / 8 B out.
// The division is the reference's `p->get_hp() / p->get_type().hp` (Map.cc:197), IEEE round-to-nearest.
// With at most 3 groups the record packs into ONE 32-bit word: hp / type.hp lies in [0, 1] (hp is capped at type.hp
// and agents with hp < 0 are off the map), so the two top bits of its float pattern are free for the group; EMPTY and
// WALL are the two all-ones-ish sentinels.  Half the footprint: the 1000 x 1000 map is 4 MB and lives in an XCD's L2.
constexpr unsigned VC_EMPTY = 0xFFFFFFFFu, VC_WALL = 0xFFFFFFFEu, VC_FOOD = 0xFFFFFFFDu;

// one cell of the painted copy, in whichever format the game uses (must match k_paint)
__device__ __forceinline__ void vc_store(const WorldView &W, int c, int code, unsigned hpbits) {
    if (W.vc_packed) {
        ((unsigned *)glob(W.viewcell))[c] = code == OCC_EMPTY ? VC_EMPTY : code == OCC_WALL ? VC_WALL : code == OCC_FOOD ? VC_FOOD : (((unsigned)code << 30) | hpbits);
        if (code >= 0 && (hpbits >> 30)) glob(W.counters)[CTR_PACK_OVERFLOW] = 1;
    } else glob(W.viewcell)[c] = make_int2(code, (int)hpbits);
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
    v = (R.minimap && rel == R.NA + 1) ? a.fx : v;                         // and is overwritten (GridWorld.cc:390-392)
    v = (R.minimap && rel == R.NA + 2) ? a.fy : v;                         // (goal_mode: its two slots come last and stay zero, :929-930)
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

__host__ __device__ inline size_t render_fast_lds(int VHW) {
    return (size_t)RENDER_WAVES * 64 * 7 * sizeof(float) + (size_t)VHW * sizeof(RenderFastPos) + (size_t)render_fast_agents(VHW) * sizeof(int4);
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

__device__ __forceinline__ int pend_action(int pend, const TypeDev &T) {   // the action number a pending action came from
    return (pend & PEND_ARG) + ((pend & ~PEND_ARG) == PEND_ATTACK ? T.n_move + T.n_turn : 0);
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
__device__ __forceinline__ int shuffle_chase_pos(int i, int A, const int *j, const int *head, const int *first, const int *link) {
    int p = j[i];
    int nxt = 0x7FFFFFFF;
    for (int e = head[p]; e != 0; e = link[e - 1]) { const int k = e - 1; if (k > i && k < nxt && k < A) nxt = k; }
    if (nxt != 0x7FFFFFFF) {
        p = nxt;
        for (int f; (f = first[p]) != 0;) { const int m = 0x7FFFFFFF - f; if (m >= A) break; p = m; }
    }
    return p;
}
__device__ __forceinline__ void shuffle_chase_body(int i, int A, const int *j, const int *head, const int *first, const int *link, int *rank) {
    rank[i] = shuffle_chase_pos(i, A, j, head, first, link);
}

// ------------------------------------------------------------------------------------------------ attack phase
// rank[seq] = position of attack-list entry `seq` after the reference's shuffle (GridWorld.cc:464-468)
// (tlist / n_tlist, one-launch step with one-cell bodies: the attacker that sets the FIRST bit of a cell appends the agent standing
// there -- every target exactly once -- and the evaluation rounds visit the targets instead of scanning every agent)
__device__ __forceinline__ void attack_rank_body(const WorldView &W, const GroupDev *gtab, int g, int i, const int *rank, unsigned *hitbits, int *tlist = nullptr,
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
                // the agent whose cell it is hears of it: "some hit word of my body is set" without a look at every one of them
                // (several attackers may say so at once: the same byte, the same value)
                if (o >= 0) glob(gtab[ref_group(o)].hitf)[ref_index(o)] = 1;
            }
        }
    }
    if (W.food_mode) { G.eat[i] = -1.0f; G.fcell[i] = -1; }
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
    if (!G.hitf[i]) return;                           // nobody hits me: I stay alive (RANK_INF, the initial value) -- no hit word is looked at
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
    if (tgt >= 0) tgt_dr = glob(gtab[ref_group(tgt)].drank_a)[ref_index(tgt)];
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
                if ((unsigned)glob(gtab[ref_group(a)].drank_a)[ref_index(a)] >= r) {       // alive at its turn
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
        if (reader >= 0) glob(gtab[ref_group(reader)].drank_b)[ref_index(reader)] = round;
        if (W.any_kill_supply)
            for (int k = 0; k < nh; k++) { const int a = s_ref[k * ATT_THREADS + tid]; glob(gtab[ref_group(a)].drank_b)[ref_index(a)] = round; }
        if (flagp) *flagp = 1;                        // (multi-launch driver: only the last round of a batch reports)
    }
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
    const bool hit = G.hitf[i] != 0;                  // (= some hit word of my body is set: attack_rank_body)
    if (hit) G.hitf[i] = 0;                           // zero between steps
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
        if (tgt >= 0) tgt_dr = glob(gtab[ref_group(tgt)].drank_a)[ref_index(tgt)];
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

// Whether a mover's chain of dependencies ends in success: its move succeeds iff the phase-start occupant of its target succeeds,
// iff ... -- every link points to a LOWER key, so the walk terminates; links are read-only once k_move_init has run.  The chain
// is walked by whoever needs the answer (round 2 resolved every agent by pointer jumping first: three to six more dependent
// launches -- or workgroup barriers, in the one-launch step -- to shorten chains that are one or two links long on average;
// a long "conga line" now costs its length in dependent loads to the agents at its tail, and nothing to anybody else).
__device__ __forceinline__ unsigned move_resolve(const GroupDev *gtab, unsigned m) {
    while (m < MV_OK) m = glob(gtab[ref_group((int)m)].mv)[ref_index((int)m)];
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

// ------------------------------------------------------------------------------------------------ the step of plain games
// Plain games -- one-cell bodies, no turn_mode / food_mode / goals / kill_supply: battle, gather, every BASELINE configuration but the
// reference's own 1M harness -- have a pipeline of their own behind the shuffle (round 4).  Four kinds of per-agent launches where the
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
//   k_plain_commit who moves -- the winner of its cell's claim whose target is empty or left by a winner in turn, asked of k_strike's records
//                  (k_move_init and k_move_commit without the map lookups: k_strike saved what it saw) -- the map and the painted map
//                  brought up to date; the agents that died in this step leave the map here, after its last reader
// Per-agent state that other agents read lives in ONE 16-byte record per agent (`rec`: a claimant reads its occupant's key, death rank
// and target with one request).  No per-cell pass is left in the step -- at BASELINE config 5's 3536 x 3536 cells the two fills of the
// generic step are 150 MB per step: the hit masks are per agent and cleaned by their owners (k_strike), and a claim word carries the
// EPOCH of the step that wrote it in its top bits (claim_word below), counting DOWN from step to step: a word of an earlier step loses
// every atomicMin against this step's claims and reads as "nobody" -- nothing is cleaned; every 63rd step the host refills the array
// (engine.hip: scratch_for).  The "inputs changed" stamps of the rounds count on across steps in the same way (PlainWorld::round_base).
constexpr unsigned MV_DIED = 0xFFFFFFFBu;   // move status between k_strike and k_plain_commit: killed or starved in this step, still on the map
constexpr unsigned MV_FAIL_SAME = 0xFFFFFFFCu;   // MV_FAIL of an agent whose hp this step left as it was: its painted cell is current (k_plain_commit)
// claim word of the plain pipeline: [63:58] epoch (0..62; 63 = the fill pattern: nobody) | [57:30] order key (boundary bit, 27-bit insertion
// index) | [29:0] agent reference.  Smaller = earlier: a later step's epoch is smaller, so stale words never win
__device__ __forceinline__ unsigned long long claim_word(int epoch, unsigned key, int ref) {
    const unsigned long long k28 = ((unsigned long long)(key >> 31) << 27) | (key & 0x7FFFFFFu);
    return ((unsigned long long)epoch << 58) | (k28 << 30) | (unsigned)ref;
}
__device__ __forceinline__ bool claim_live(unsigned long long w, int epoch) { return (int)(w >> 58) == epoch; }
__device__ __forceinline__ int claim_ref(unsigned long long w) { return (int)(w & 0x3FFFFFFFu); }

// rules of the shape Event(a, attack | kill, b) that pay receivers bound to `a` only: evaluated by the agent itself, in rule order,
// as soon as its own attack is known -- provided every last_op was OP_NULL when the step began (clear_dead has run since the last step:
// otherwise an event of the LAST step is paid again unless a collision overwrites it, which only the move phase knows; the host then
// runs k_rule behind the commit as before)
// (struct StrikeRules: launch.h)


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

// every move candidate (other than `self`) with a lower key whose target rectangle covers cell c: the cell's list (movg_prep_body: every
// candidate threads one node per cell of its target rectangle onto that cell -- `head[c]` = last node + 1, 0 = none; a node = {next link,
// the candidate}).  f(packed ref, move status) -> true stops the walk.  (Rounds 1-4 found the entrants by PULLING -- every (group, move,
// body offset) that could reach the cell probed on the map: ~60 dependent probes per contested cell in a pursuit world, and the slowest
// lane sets a wave's time: k_movg_sweep 121 us at 1M agents.  A list is two or three entries long.)
template <class F>
__device__ __forceinline__ void for_each_entrant(const WorldView &W, const GroupDev *gtab, const unsigned *head, int c, unsigned key, int self, F f) {
    for (unsigned h = head[c]; h != 0u;) {
        const int2 nd = W.mv_nodes[h - 1u];
        h = (unsigned)nd.x;
        const int e = nd.y;
        if (e == self) continue;
        const GroupDev &A = gtab[ref_group(e)];
        const int ei = ref_index(e);
        if (A.key[ei] >= key) continue;
        if (f(e, A.mv[ei])) return;
    }
}
// is some OTHER candidate's target rectangle on cell c?  (`self` is a candidate onto c itself: its node is in the list)
__device__ __forceinline__ bool other_entrants(const WorldView &W, const unsigned *head, int c, int self) {
    const unsigned h = head[c];
    if (h == 0u) return false;
    const int2 nd = W.mv_nodes[h - 1u];
    return nd.y != self || nd.x != 0;
}
// (movg_prep_body) candidate `self` = (g, i) onto the k-th cell of its target rectangle
__device__ __forceinline__ void entrant_push(const WorldView &W, unsigned *head, int c, int g, int i, int cells, int k) {
    const unsigned node = (unsigned)W.node_base[g] + (unsigned)i * (unsigned)cells + (unsigned)k;
    const unsigned old = atomicExch(&head[c], node + 1u);
    W.mv_nodes[node] = make_int2((int)old, ref_pack(g, i));
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
__device__ __forceinline__ void for_each_mover_onto(const WorldView &W, const GroupDev *gtab, const unsigned *head, int cx, int cy, unsigned key, int self, F f) {
    for_each_entrant(W, gtab, head, cy * W.w + cx, key, self, f);     // (turn_mode too: a candidate registers the rectangle it would enter as it faces)
}

// MODE 0: is the move blocked? (stops at the first definite obstacle)   MODE 1: all moves are decided -- who is the
// collide object?   MODE 2 (can_absorb types present): the outcome depends on WHICH agent is met first, so the scan
// stops at the first cell that holds an agent or whose state is still unknown
template <int MODE>
__device__ __forceinline__ MoveProbe move_probe(const WorldView &W, const GroupDev *gtab, int g, int i, int tgt_cell, const unsigned *head) {
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
            // entrants with lower keys: the other candidates on the cell's list (mine is one of its nodes)
            if ((occupant < 0 || unknown) && other_entrants(W, head, c, self))
                for_each_mover_onto(W, gtab, head, cx, cy, key, self, [&](int e, unsigned st) {
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
__device__ __forceinline__ void movg_prep_body(const WorldView &W, int g, int i, unsigned *head, int slot, bool starve) {
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
    // goals that move: Env::set_action_device sends a step in which goals were given actions through k_step_serial; reached only through
    // env_cycle_many (actions handed to the fused cycle), where it is reported, not guessed
    if (t >= 0 && T.can_absorb) { W.counters[CTR_UNSUPPORTED] = 1; t = -1; }
    G.drank_a[i] = t;
    G.mv[i] = t >= 0 ? 0u : MV_FAIL;      // 0 = undecided (the packed-dependency encoding of the 1x1 path is not used here)
    if (t >= 0) {
        const int ny = t / W.w, nx = t - ny * W.w;
        for (int by = 0; by < fp.y; by++)
            for (int bx = 0; bx < fp.x; bx++) entrant_push(W, head, (ny + by) * W.w + nx + bx, g, i, T.bw * T.bl, by * fp.x + bx);
    }
}

__device__ __forceinline__ void movg_sweep_body(const WorldView &W, const GroupDev *gtab, int g, int i, const unsigned *head, int *flagp) {
    const GroupDev &G = W.grp[g];
    const int t = G.drank_a[i];
    if (t < 0 || G.mv[i] != 0) return;
    if (!W.any_absorb) {
        MoveProbe r = move_probe<0>(W, gtab, g, i, t, head);
        if (r.blocked) G.mv[i] = MV_FAIL;
        else if (!r.undecided) G.mv[i] = MV_OK;
        else if (flagp) *flagp = 1;
        return;
    }
    // Map::do_move with goals (Map.cc:334-353): the collide object is the first agent met; a goal that is still free
    // takes the mover in, a taken one is bumped without any effect
    MoveProbe r = move_probe<2>(W, gtab, g, i, t, head);
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
                const int2 gd = body_dims(W, B, TB, bi);
                bool lost = false, unknown = false;
                for (int bx = 0; bx < gd.x && !lost; bx++)
                    for (int by = 0; by < gd.y && !lost; by++) {
                        const int cx = B.x[bi] + bx, cy = B.y[bi] + by;
                        for_each_mover_onto(W, gtab, head, cx, cy, key, self, [&](int, unsigned s2) {      // (a cell nobody else reaches: an empty walk)
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

// Map::get_collide for failed moves (Map.cc:334-353, 486-501): first agent met in the target rectangle
__device__ __forceinline__ void movg_collide_body(const WorldView &W, const GroupDev *gtab, int g, int i, const unsigned *head) {
    const GroupDev &G = W.grp[g];
    const int t = G.drank_a[i];
    if (t < 0) return;
    const unsigned st = G.mv[i];
    if (st == MV_FAIL) {
        MoveProbe r = move_probe<1>(W, gtab, g, i, t, head);
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

__device__ __forceinline__ void movg_vacate_body(const WorldView &W, int g, int i) {
    const GroupDev &G = W.grp[g];
    if (G.drank_a[i] < 0 || !(G.mv[i] == MV_OK || mv_taken(G.mv[i]))) return;
    const int2 fp = body_dims(W, G, W.type[g], i);
    cells_clear(W, G.x[i], G.y[i], fp.x, fp.y);
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
                if (P.n_obj) atomicAdd(&glob(gtab[P.gb].hits)[ref_index(ent[1])], 1);
            }
        }
    }
    if (__ballot(trig) && lane_id() == 0) W.counters[CTR_TRIGGER + P.rule_no] = 1;
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

// ================================================================================================ launchers
static inline dim3 grid_all(const WorldView &W, int threads) {
    int mx = 1;
    for (int g = 0; g < W.G; g++) mx = W.grp[g].n > mx ? W.grp[g].n : mx;
    return dim3((mx + threads - 1) / threads, W.G);
}

}  // namespace magent_amd
