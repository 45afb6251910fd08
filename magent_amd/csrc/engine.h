// engine.h -- data model shared by the HIP kernels (render.hip, step.hip, cycle.hip over kernels_dev.h) and the host engine (engine.hip).
//
// MI355X-first layout (not the reference's AoS heap objects, GridWorld.h:131-253 / Map.h:23-29):
//   * every group is a set of struct-of-arrays device buffers indexed by the agent's position in the group
//     (the reference's vector<Agent*> order), so all per-agent kernels load/store fully coalesced;
//   * the map is one int32 per cell (`occ`): EMPTY, WALL or a packed agent reference (group, index);
//   * `viewcell` is a painted copy of the map for the observation renderer: {group code, hp / type.hp} per cell
//     (one 32-bit word when there are <= 3 groups, else 8 bytes), so the renderer does ONE load per cell and no
//     dependent gather;
//   * order-dependent phases (attack, move) do not keep lists: each agent carries its pending action and an order
//     key, and the sequential result is recovered by fixed-point / pointer-jumping kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace magent_amd {

constexpr int MAXG = 8;                 // groups per environment
constexpr int REF_SHIFT = 27;           // packed agent reference = (group << 27) | index
constexpr int REF_MASK = (1 << REF_SHIFT) - 1;
constexpr int OCC_EMPTY = -1;
constexpr int OCC_WALL = -2;
constexpr int OCC_FOOD = -3;            // food_mode: what an agent killed by an attack leaves on the attacked cell (Map.cc:276-283)

// pending action of an agent for the coming step: (kind << 16) | payload
constexpr int PEND_NONE = 0;
constexpr int PEND_MOVE = 1 << 16;
constexpr int PEND_ATTACK = 2 << 16;
constexpr int PEND_TURN = 3 << 16;      // turn_mode: payload = the action number itself (the reference subtracts move_base, GridWorld.cc:430-433)
constexpr int PEND_ARG = 0xFFFF;

// EventOp values the engine stores in last_op (reference grid_def.h:18-24)
constexpr int OP_KILL = 3, OP_COLLIDE = 6, OP_ATTACK = 7, OP_NULL = 11;

constexpr int RANK_INF = 0x7FFFFFFF;    // "never dies in the attack phase"
constexpr unsigned MV_FAIL = 0xFFFFFFFFu, MV_OK = 0xFFFFFFFEu;  // move status; anything else = "depends on ref"
// generic move resolution with can_absorb types: the mover bumped into a goal that is already taken (nothing happens),
// or is itself taken in by goal `ref` (it dies and leaves the map at its turn): 0x80000000 | ref, ref < 2^30
constexpr unsigned MV_SILENT = 0xFFFFFFFDu, MV_TAKEN_BIT = 0x80000000u;
__host__ __device__ inline bool mv_taken(unsigned s) { return (s >> 30) == 2u; }
__host__ __device__ inline int mv_taken_by(unsigned s) { return (int)(s & 0x3FFFFFFFu); }
constexpr unsigned long long CLAIM_NONE = ~0ull;

// turn_mode: the way an agent faces (grid_def.h:15); moves, attack offsets and the observation window are given in the
// agent's frame.  NORTH is the frame of the map -- every agent faces north when turn_mode is off.
constexpr int DIR_EAST = 0, DIR_SOUTH = 1, DIR_WEST = 2, DIR_NORTH = 3, DIR_NUM = 4;
// Map::rela_to_abs without the centre (Map.cc:515-532): an offset in the agent's frame -> an offset on the map
__host__ __device__ inline void dir_rotate(int dir, int rx, int ry, int &ax, int &ay) {
    switch (dir) {
        case DIR_NORTH: ax = rx; ay = ry; break;
        case DIR_SOUTH: ax = -rx; ay = -ry; break;
        case DIR_WEST: ax = ry; ay = -rx; break;
        default: ax = -ry; ay = rx; break;   // EAST
    }
}

// Map::save_to_real / real_to_save (Map.cc:553-587): the stored top-left cell of a wd x ln body <-> its reference corner, the
// cell the agent-frame offsets (attack, view) are counted from
__host__ __device__ inline void saved_to_real(int dir, int wd, int ln, int x, int y, int &rx, int &ry) {
    switch (dir) {
        case DIR_NORTH: rx = x; ry = y; break;
        case DIR_SOUTH: rx = x + wd - 1; ry = y + ln - 1; break;
        case DIR_WEST: rx = x; ry = y + wd - 1; break;
        default: rx = x + ln - 1; ry = y; break;   // EAST
    }
}
__host__ __device__ inline void real_to_saved(int dir, int wd, int ln, int rx, int ry, int &x, int &y) {
    switch (dir) {
        case DIR_NORTH: x = rx; y = ry; break;
        case DIR_SOUTH: x = rx - wd + 1; y = ry - ln + 1; break;
        case DIR_WEST: x = rx; y = ry - wd + 1; break;
        default: x = rx - ln + 1; y = ry; break;   // EAST
    }
}

__host__ __device__ inline int ref_pack(int g, int i) { return (g << REF_SHIFT) | i; }
__host__ __device__ inline int ref_group(int r) { return r >> REF_SHIFT; }
__host__ __device__ inline int ref_index(int r) { return r & REF_MASK; }

// per-type constants (reference AgentType.h:14-46), by group
struct TypeDev {
    float hp, damage, step_recover, kill_supply, kill_reward, dead_penalty, attack_penalty, step_reward;
    float food_supply, eat_ability;   // food_mode (AgentType.h:30)
    int attack_in_group;
    int can_absorb;              // a "goal": the first mover that bumps into it is taken in (Map.cc:341-350)
    int bw, bl;                  // body width (x) and length (y) in cells; the agent's position is its top-left cell
    int n_move, n_attack;        // action layout: [0, n_move) moves, (turn_mode: n_turn = 2 turns,) then n_attack attacks
    int n_turn;
    int move_off, attack_off;    // offsets into WorldView::delta (int2 {dx,dy} per action payload)
    int attack_bit;              // first bit of this group's attack offsets in the per-cell hit word
    int view_w, view_h;          // observation window
    int view_x1, view_y1;        // window origin relative to the agent position (includes view_x/y_offset)
    int mask_off;                // offset into WorldView::mask (view_h * view_w bytes, 1 = inside the view range)
};

// device arrays of one group; n = current number of agents (dead ones included until clear_dead)
struct GroupDev {
    int n;
    int *x, *y, *id, *last_action, *op_obj, *pend;
    float *hp, *next_reward, *last_reward;
    unsigned char *dead, *last_op;
    unsigned char *absorbed;     // can_absorb types: this goal has taken a mover in (GridWorld.h:191-192)
    int *dir;                    // turn_mode: the way the agent faces (null when turn_mode is off: everybody faces north)
    unsigned *key;               // attack: sequence number -> rank after the shuffle; move: order key
    int *drank_a, *drank_b;      // attack fixed point: rank at which the agent dies (ping-pong)
    unsigned *mv;                // move resolution status / dependency (generic step); hp after the attack phase, as float bits (every step)
    unsigned char *hitf;         // generic attack phase: somebody's attack lands on me in this step (set by the attacker in attack_rank_body,
                                 // cleared by the owner in attack_apply_body; zero between steps) -- what saves every agent a look at the hit
                                 // words of all its body cells: 2.5 M random reads per pass in a 1 M-agent pursuit world
    int *hits;                   // reward rules: number of rule hits received as the object of an event
    // food_mode scratch of the attack phase: what my attack eats (-1 = it eats nothing), written by the owner of the
    // food; the cell on which I was killed and what is left of the food there (-1 = none)
    float *eat, *fleft;
    int *fcell;
};

struct WorldView {
    int w, h, G;
    int *occ;
    int2 *viewcell;
    unsigned long long *claim;
    unsigned *hitbits;           // per cell: one bit per (attacker group, attack offset) that hits it in this step's attack phase
    const int2 *delta;
    const unsigned char *mask;
    int *counters;               // CTR_* below: changed flag, attack count, dead_ct per group, gates, rule triggers
    TypeDev type[MAXG];
    GroupDev grp[MAXG];
    int any_kill_supply;
    int any_multicell;           // some group has a body larger than one cell (or can absorb): generic move resolution
    int any_absorb;              // some type is can_absorb
    int food_mode;               // GridWorld.cc:131
    float *food, *food_next;     // per cell: amount of food on OCC_FOOD cells; attack-phase scratch (-1 = eaten up)
    int large_map, bandwidth;    // reference large_map_mode striping (GridWorld.cc:75-85, 407-425)
    int turn_mode;               // GridWorld.cc:134
    int reach;                   // turn_mode: how far (in cells, per axis) the top-left cell of a body can be from a cell its move or turn enters
    int vc_packed;               // viewcell holds one 32-bit word per cell (<= 3 groups, no goals), else an int2
    int live_paint;              // the step keeps `viewcell` current itself (vacated cells, then every live agent's body)
    int plain;                   // one-cell bodies, no turn_mode / food_mode / goals / kill_supply: the games the fused step takes
    // generic move resolution: the candidates onto a cell, as a list -- node = {link to the next node + 1 (0: none), the candidate}; the list
    // heads are per cell (the step's `wanted` / hit words); a candidate of group g owns nodes node_base[g] + i * (bw * bl) + 0 .. bw * bl - 1
    int2 *mv_nodes;
    int node_base[MAXG];
};

// What a step reports to the host.  The one-launch step (k_step_solo) writes it straight into pinned host memory and
// publishes `seq` last; the host spins on `seq` instead of synchronising the stream.
struct alignas(16) StepRecord {
    int dead[MAXG], taken[MAXG]; // dead_ct per group as of this step (accumulated until clear_dead); movers taken in by goals
    unsigned long long triggers; // bit k: reward rule k fired in this step
    unsigned rng;                // engine RNG state after the step
    int last_a;                  // attack-list length
    int unsupported, pack_overflow, error, bad_action, hit_overflow;
    int rounds_attack, rounds_move;
    unsigned rounds_mask;        // plain games: bit r = round r of the death-rank fixed point changed a death rank (r + 1 rounds were needed)
    int open_attack, open_move;  // multi-launch step: the optimistic rounds of a phase ran out (the host continues from that state)
    int n_marks; unsigned long long marks[40];   // wall_clock64 (100 MHz) at the phase boundaries of k_step_solo (tuning aid)
    volatile int seq;            // == the step's sequence number once everything above is visible
};

constexpr int CTR_CHANGED = 0, CTR_ATTACK = 1, CTR_DEAD = 2 /* unused: see CTR_DEAD_SPREAD */, CTR_PACK_OVERFLOW = 12, CTR_TRIGGER = 16, CTR_TRIGGER_END = 64;
// movers taken in by goals, per group: dead, but not counted in the reference's dead_ct (Map.cc:345); cleared with CTR_DEAD
constexpr int CTR_TAKEN = 64, CTR_UNSUPPORTED = 72;
constexpr int CTR_HIT_OVERFLOW = 74;  // turn_mode: a target collected more hits than its list holds (the list is sized for the worst case otherwise)
constexpr int CTR_GOALS_ACT = 75;    // scratch of Env::set_action_device: some goal was given an action that is not the zero move
constexpr int CTR_BAD_ACTION = 73;   // set_action met an action outside [0, n_action): reported at the end of the step (the reference: UB)
// Deaths are counted in DEAD_SLOTS counters per group, each on its own cache line: device-scope atomics on ONE address
// serialise at ~15 ns apiece on this part (measured: 4.8k of them cost a 800k-agent step 70 us).  dead_ct of group g =
// sum over slots of counters[dead_slot(g, slot)]; the host adds them up after its one readback per step.
constexpr int DEAD_SLOTS = 16, CTR_DEAD_SPREAD = 128, CTR_ATT_SPREAD = CTR_DEAD_SPREAD + MAXG * DEAD_SLOTS * 16;
__host__ __device__ inline int dead_slot(int g, int slot) { return CTR_DEAD_SPREAD + (g * DEAD_SLOTS + slot) * 16; }
// the attack-list length as the tiled set_action leaves it: ATT_SLOTS partial counts, each on its own cache line (k_set_action_a adds one
// atomic per tile; k_shuffle_draw -- or the host, in the checked driver -- adds them up into CTR_ATTACK); zero between steps
constexpr int ATT_SLOTS = 64, CTR_TOTAL = CTR_ATT_SPREAD + ATT_SLOTS * 16;
__host__ __device__ inline int att_slot(int slot) { return CTR_ATT_SPREAD + slot * 16; }
// single-sync step: the fixed-point rounds are launched optimistically; the LAST round of a phase reports whether
// anything still changed.  A phase left open makes every later kernel of the step return at once, and the host
// continues from exactly that state (adjacent: cleared together)
constexpr int CTR_OPEN_ATTACK = 10; // the optimistic attack rounds ran out
constexpr int CTR_OPEN_MOVE = 11;   // the optimistic move rounds ran out
constexpr int CTR_RNG = 13;         // engine RNG state (minstd_rand0), advanced on the device by the attack shuffle
constexpr int CTR_LAST_A = 14;      // attack-list length of the last step (host information)
// plain games: slot (round & 31) is raised by every agent whose death rank changes in that round of the attack fixed point; a round
// whose predecessor raised nothing has nothing to do and returns at once (k_attack_eval).  Zero between steps.
constexpr int CTR_ROUND_CHANGED = 80, ROUND_SLOTS = 32;

// observation render parameters for one get_observation(group) call
struct RenderArgs {
    int g;                       // observing group
    int n;                       // agents to render
    int VH, VW, C, S;            // window, channels, S = VH*VW*C floats per agent
    int F, E, NA;                // feature size, embedding size, n_action
    int minimap;                 // minimap_mode
    int turn;                    // turn_mode: the window is laid out in the agent's frame
    int food;                    // food_mode: channel 1 shows food, the group blocks start at 2 (GridWorld.cc:915-924)
    int scale_w, scale_h;
    int chan_desc[32];           // per output channel: (kind << 8) | (code & 0xff); kind 0 = has, 1 = hp, 2 = minimap
    int totals[MAXG];            // group sizes (minimap divisor)
    const float *mini;           // float[G][VH*VW]: count / total per group (k_minimap)
    float *view, *feat;
    int cells16;                 // 1: `view` is bf16 [n][VH][VW][8] -- channels 0..C-1, zeros, and 1.0 in channel 7 (include/magent_policy.h)
};

}  // namespace magent_amd
