// launch.h -- host-visible launch wrappers of render.hip / step.hip / cycle.hip and the small plan structs they take.
#pragma once
#include "engine.h"

namespace magent_amd {

// n / d for a divisor fixed at launch time: q = (t + ((n - t) >> 1)) >> shift with t = umulhi(n, mul)
// (round-up method; exact for every 32-bit n).  d == 1 is flagged.
struct FastDiv {
    unsigned mul, shift, one;
};
inline FastDiv make_fastdiv(unsigned d) {
    FastDiv f{0, 0, 0};
    if (d <= 1) { f.one = 1; return f; }
    unsigned s = 0;
    while ((1ull << s) < d) s++;                         // s = ceil(log2 d), 1..32
    unsigned long long m = ((1ull << 32) * ((1ull << s) - d)) / d + 1;
    f.mul = (unsigned)m;
    f.shift = s - 1;
    return f;
}

// how the render kernel tiles one get_observation call
struct RenderPlan {
    int spans;           // workgroups; each owns `steps_per_span` consecutive 64-cell steps of the flat cell sequence
    int steps_per_span;
    int feat_blocks;     // trailing workgroups of the render launch that write the feature rows (0 = separate launch)
    int xcd_chunk;       // spans / 8 when the XCD-aware span mapping is on, else 0
    int strip_floats;    // 64 * C: wave-private LDS strip
    int unroll;          // 64-cell steps whose loads a wave keeps in flight together (1, 2, 4 or 8)
    FastDiv div_vhw, div_vw, div_f, div_scale_w, div_scale_h;
};

// the observations of several groups of a small world in one launch (k_render_multi)
constexpr int RENDER_MULTI_MAX = 4;
struct RenderMulti {
    int n;
    int blocks[RENDER_MULTI_MAX];          // workgroups of slot k: spans + feature blocks
    RenderArgs R[RENDER_MULTI_MAX];
    RenderPlan P[RENDER_MULTI_MAX];
};

// one reward rule, symbols 'any':  Event(a, op, b)                        -- pair == 0: ga/op = subject, gb = object
//                                   Event(x, op, c) & Event(y, op_y, c)     -- pair == 1: ga/op = x (the symbol the reference's
//                                       search binds first = the lower-numbered one), gy/op_y = y, gb = c
struct RuleArgs {
    int ga, gb, op, rule_no;
    int n_subj, n_obj;            // receivers that are the (first) subject / the object of the event
    float v_subj[4], v_obj[4];
    int pair, gy, op_y, n_y;
    float v_y[4];
    int prog;                     // >= 0: index of the rule's RuleProg (a general single-iterator expression), else -1
};

// A rule whose search iterates ONE 'any' symbol x (group ga); a second symbol y, if any, is inferred as x's op_obj
// (group gb).  The event expression in postfix: leaves test x (slot 0) or y (slot 1), and / or / not combine them.
struct RuleProg {
    int ga, gb, has_obj, rule_no, n_subj, n_obj;
    float v_subj[4], v_obj[4];
    int n;
    int op[24];                   // EventOp: 0 and, 1 or, 2 not, 3 kill, 4 at, 5 in, 6 collide, 7 attack, 8 die
    int a[24][5];                 // kill / collide / attack: {subject slot, object slot}; at: {slot, x, y}; in: {slot, x1, y1, x2, y2}; die: {slot}
};

// alternate copies of the arrays that survive clear_dead (compaction is a stable scatter into them, then they change places)
struct AltArrays { int *x, *y, *id, *last_action; float *hp, *next_reward, *last_reward; unsigned char *absorbed; int *dir; };

// the minimap of the next observations, made by the launch that ends a cycle (vh == 0: not asked for)
struct MiniArgs {
    int vh, vw, scale_w, scale_h, skip;
    float *out;                    // float[G][vh * vw]
};

// the one-launch step of small worlds (k_step_solo)
struct SoloStep {
    int *sj, *shead, *sfirst, *slink, *rank;   // shuffle scratch (head / first zero between steps; link doubles as the target list)
    const unsigned *powtab;
    unsigned *hit;                 // per cell: attack hit bits, then the generic move's `wanted` counters; zero between phases
    const RuleArgs *rules;         // device copies of the compiled reward rules
    const RuleProg *progs;
    int n_rules;
    int kmax, nt_eval;             // hit-list depth; threads that evaluate (kmax * nt_eval * 8 bytes of dynamic LDS)
    int max_rounds;                // bound of the fixed-point loops (reported as an error, never silently cut)
    StepRecord *rec;               // pinned host memory
    int seq;
    // ---- the rest of an environment cycle in the same launch (env_cycle_many); all optional
    const int *actions[MAXG];      // set_action(g) before the step, groups in ascending order (null: not in this launch)
    int call_base[MAXG];           // ... its insertion-order base (agents given actions by earlier calls of this step)
    float *rewards[MAXG];          // get_reward(g) after the step (null: skip)
    float group_reward[MAXG];
    int do_clear;                  // clear_dead after that, into `dst`; the device tables are refreshed
    AltArrays dst[MAXG];
    GroupDev *gtab_out; TypeDev *ttab_out;
    MiniArgs mini;                 // the minimap of the next observations
};

// one environment of a batched cycle launch (env_cycle_many: one workgroup of k_step_solo_batch per environment)
struct BatchItem {
    WorldView W;
    SoloStep S;
    RenderMulti M;
};
// one set_action call of a step in which some group was given actions more than once (k_step_serial)
struct SerialCall { int g; const int *actions; };
void launch_step_serial(hipStream_t s, const WorldView &W, const SerialCall *calls, int n_calls, int2 *alist, int4 *mlist, int4 *msorted, int n_sep, int4 *events);
void launch_touch_map(hipStream_t s, const WorldView &W);
void launch_pend_to_actions(hipStream_t s, const GroupDev &G, const TypeDev &T, int *out);
void launch_any_real_action(hipStream_t s, const int *actions, int n, const TypeDev &T, const int2 *delta, int *flag);
void launch_step_report(hipStream_t s, int *counters, StepRecord *rec, int seq, int NG);
void launch_commit_action(hipStream_t s, const GroupDev &G, const TypeDev &T);
void launch_cycle_batch(hipStream_t s, const BatchItem *d_items, int n_env, int slots, int max_blocks, size_t render_lds, size_t step_lds);
size_t render_strip_lds(const RenderPlan &P);
size_t solo_step_lds(const WorldView &W, const SoloStep &S);

void launch_set_tables(hipStream_t s, const WorldView &W, GroupDev *gtab, TypeDev *ttab);
void launch_paint(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab);
void launch_minimap(hipStream_t s, const WorldView &W, const RenderArgs &R, int *counts, float *mini);
int launch_render(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4, bool nt);   // -> which kernel ran (0 generic, 1 fast, 4 sweep)
void launch_render_multi(hipStream_t s, const WorldView &W, const RenderMulti &M);
void launch_features(hipStream_t s, const WorldView &W, const RenderArgs &R, const RenderPlan &P, bool vec4);
// scratch of the attack shuffle: four int arrays of (at least) n_max entries; head / first are zero between steps
struct ShuffleBufs { int *head, *first, *j, *link; };
void launch_shuffle(hipStream_t s, int n_max, int *counters, const ShuffleBufs &B, int *rank, unsigned *hitbits, size_t ncell, const unsigned *powtab, bool tiled);
void launch_set_rng(hipStream_t s, int *counters, unsigned x);
void launch_step_reset(hipStream_t s, int *counters);
void launch_set_counter(hipStream_t s, int *counters, int index, int value, int unless_index);
// where a group's set_action call of this step left its tile counts (step.hip: k_set_action_a; kernels_dev.h: attack_seq); -1: none / numbers in `key`
struct SeqPlan { int off[MAXG]; };
void launch_set_action(hipStream_t s, const WorldView &W, int g, const int *actions, int call_base, int *sums, int *wpre, int tile_off /* < 0: one-workgroup form */);
void launch_seq_assign(hipStream_t s, const WorldView &W, int g, const int *sums, const int *wpre, int tile_off, bool write_total);
void launch_attack_rank(hipStream_t s, const WorldView &W, const GroupDev *gtab, const int *rank, const ShuffleBufs &B, bool clear_hitbits, const int *sums, const int *wpre,
                        const SeqPlan &P);
void launch_attack_iter(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int round, int kmax, int flag /* counter to raise on a change, < 0 = none */);
void launch_attack_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int kmax);
// ---- the step of plain games (step.hip: "the step of plain games"): per-agent scratch of its own
//   rec   {x: order key (move) | rank in the shuffled attack list (attack), y: death rank, z: the cell the move is aimed at (-1: none),
//          w: move status / dependency}: what OTHER agents read of an agent
//   atk   the agent my attack lands on (-1: nobody); from k_strike on, for a mover: what its target cell holds when the moves begin
//   hmask bit (attacker group's attack_bit + offset) set: that attacker hits me in this step; hlist[agent][slot] = {rank, attacker}
struct PlainGroup { int4 *rec; int *atk; unsigned *hmask; uint2 *hlist; };
struct PlainWorld {
    PlainGroup g[MAXG];
    int S, kmax;          // slots per agent (attack offsets of all groups); most hits one agent can receive
    int *alive;           // k_strike leaves the survivors of every 256 agents here (alive_off[g] + block): clear_dead's compaction needs no count pass
    int alive_off[MAXG];
    int epoch;            // of this step's claim words (step.hip: claim_word): 62 - (plain step number mod 63)
    int round_base;       // + round = the "inputs changed" stamp of a round of this step (they count on from step to step)
};
// the rules k_strike evaluates itself (kernels_dev.h: "rules of the shape Event(a, attack | kill, b) that pay receivers bound to `a` only")
struct StrikeRules {
    int n;
    struct One { int ga, gb, op, rule_no, n_subj; float v[4]; } r[4];
};
StrikeRules strike_rules(const RuleArgs *rules /* null: none fused */, int n_rules);
bool fused_rules(const RuleArgs *rules, int n);
bool plain_eval_lds_ok(int kmax);
void launch_shuffle_draw(hipStream_t s, int n_max, int *counters, const ShuffleBufs &B, const unsigned *powtab, bool tiled);
void launch_plain_rank(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const ShuffleBufs &B, const int *sums,
                       const int *wpre, const SeqPlan &P);
void launch_plain_eval(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab, int round, int flag,
                       const ShuffleBufs &B);
void launch_plain_tail(hipStream_t s, const WorldView &W, const PlainWorld &PW, const PlainGroup *ptab, const GroupDev *gtab, const TypeDev *ttab,
                       const RuleArgs *rules /* null: not fused */, int n_rules, StepRecord *rec /* non-null: the step's report goes out before the moves */, int seq);
void launch_food_iter(hipStream_t s, const WorldView &W, const GroupDev *gtab, const TypeDev *ttab, int round, int kmax, int flag);
void launch_attack_events(hipStream_t s, const WorldView &W, int4 *ev);
void launch_move_prep(hipStream_t s, const WorldView &W, const GroupDev *gtab);   // starve / recover, then the move candidates
void launch_movg_prep(hipStream_t s, const WorldView &W, bool starve);
void launch_turn_prep(hipStream_t s, const WorldView &W);
void launch_turn_sweep(hipStream_t s, const WorldView &W, const GroupDev *gtab, int flag);
void launch_turn_apply(hipStream_t s, const WorldView &W);
void launch_movg_sweep(hipStream_t s, const WorldView &W, const GroupDev *gtab, int flag);
void launch_movg_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab);
void launch_move_apply(hipStream_t s, const WorldView &W, const GroupDev *gtab);
void launch_rule(hipStream_t s, const WorldView &W, const RuleArgs &A);
void launch_rules(hipStream_t s, const WorldView &W, const RuleArgs *rules, int n, const RuleProg *progs, const GroupDev *gtab);
void launch_finish(hipStream_t s, const WorldView &W);
void launch_get_reward(hipStream_t s, const GroupDev &G, float group_reward, float *out);
void launch_get_pos(hipStream_t s, const GroupDev &G, int *out);
void launch_get_alive(hipStream_t s, const GroupDev &G, unsigned char *out);
// clear_dead for every group in three launches (count, compact / init_reward, reset + device tables)
struct ClearArgs {
    int mode[MAXG];        // 0 nothing (empty group), 1 Agent::init_reward only, 2 compaction of the survivors
    int sums_off[MAXG];    // where the group's block totals start in `sums`
    int sums_per_tile;     // entries of `sums` per compaction tile: 1 (k_clear_count), or SCAN_ITEMS when k_strike left one per 256 agents
    typedef AltArrays Alt;
    Alt dst[MAXG];
};
void launch_clear_compact(hipStream_t s, const WorldView &W, const ClearArgs &A, int *sums, const MiniArgs &M, int *counts);
void launch_clear_finish(hipStream_t s, const WorldView &Wnew, const ClearArgs &A, GroupDev *gtab, TypeDev *ttab, const MiniArgs &M, int *counts);
// clear_dead's own minimap histogram (large worlds): MINI_COPIES copies of [G][VHW] counts, behind the ordinary histogram's
// [MAXG left-out counters][G][VHW] in the same buffer
constexpr int MINI_COPIES = 16;
void launch_mini_norm(hipStream_t s, const WorldView &Wnew, const MiniArgs &M, int *counts);
void launch_clear_solo_all(hipStream_t s, const WorldView &W, const ClearArgs &A, GroupDev *gtab, TypeDev *ttab, const MiniArgs &M);
bool compact_is_solo(int n);
bool attack_lds_ok(int kmax);
void launch_step_solo(hipStream_t s, const WorldView &W, const SoloStep &S);
int solo_step_static_lds();
bool solo_step_allow_lds(size_t bytes);
void launch_init_reward(hipStream_t s, const WorldView &W, int g);
void launch_compact(hipStream_t s, const WorldView &W, int g, const GroupDev &D, int new_n, int *sums);

// ---- many environments per launch through the pipeline of plain games (env_cycle_many on worlds beyond the one-launch step; pipe.hip).
// One PipeItem per environment, in a device array; every kernel of the batch finds its environment by blockIdx.z.  The whole cycle --
// observations, set_action, step, get_reward, clear_dead -- is ONE chain of launches without a host round trip in it: what the host would
// decide between the step and clear_dead (which groups compact, the sizes behind the compaction) the kernels read from the death counters.
struct PipeItem {
    WorldView W;
    PlainWorld PW;
    const PlainGroup *ptab; const GroupDev *gtab; const TypeDev *ttab;     // the environment's device tables
    ShuffleBufs B; const unsigned *powtab;
    int *sums, *wpre;                  // tile counts of this step's set_action calls (k_set_action_a's)
    SeqPlan P;                         // where each group's call leaves them (-1: the group is given no actions)
    const int *actions[MAXG]; int call_base[MAXG];
    StrikeRules R;
    StepRecord *rec; int seq;          // DEVICE memory: the step's report (the batch's last workgroup sends all of them to the host at once)
    int n_max;                         // agents of all groups: the bound of the attack list
    RenderMulti M;                     // the observations (M.n == 0: rendered by launches of the environment's own, ahead of the batch)
    // ---- get_reward + clear_dead (an environment whose attack rounds ran out is skipped: the host finishes its step and clears it)
    ClearArgs A;                       // (mode: decided on the device; sums_off / dst as Env::clear_dead sets them)
    MiniArgs Mi; int *counts;          // the next observations' minimap, folded into the compaction
    const int *alive_sums;             // k_strike's survivor counts
    float *rewards[MAXG]; float group_reward[MAXG];
    GroupDev *gtab_out; TypeDev *ttab_out;
    int *newn;                         // [MAXG] device scratch: group sizes behind the compaction (k_pipe_clear -> k_pipe_finish)
    int pad_[2];
};
// what the batch's last workgroup does: every environment's report to pinned host memory in one piece, then the word the host waits for
struct PipeCtl { const StepRecord *reports_d; StepRecord *reports_h; int *ticket; int *flag_h; int flag_seq; int n_env; };
constexpr int PIPE_REPORT_BYTES = 128;     // of a StepRecord: everything ahead of the tuning marks
struct PipeDims { int n_env, max_n, max_total, G, slots, render_blocks, rounds, kmax, sweep, hist_cells; size_t render_lds; };
void launch_pipe_upload(hipStream_t s, const PipeItem *h_items, PipeItem *d_items, int n_env);      // (h_items: pinned, device-visible)
void launch_pipe_cycle(hipStream_t s, const PipeItem *d_items, const PipeDims &D, const PipeCtl &C);
bool render_sweep_mini_ok(const WorldView &W, const RenderArgs &R);
size_t render_sweep_lds(int VHW, int C);

// agents per workgroup of the scan-based passes (set_action, clear_dead).  2 per thread: at 400k agents that is 782 workgroups --
// the earlier 8 per thread left 196, less than one per CU, and every such launch was bound by its own latency chain
#ifndef MAGENT_SCAN_ITEMS
#define MAGENT_SCAN_ITEMS 2
#endif
constexpr int SCAN_ITEMS_HOST = MAGENT_SCAN_ITEMS;
constexpr int SCAN_TILE_HOST = 256 * SCAN_ITEMS_HOST;
constexpr int ATTACK_KMAX_HOST = 32;     // attack offsets of all groups share one 32-bit word per cell

}  // namespace magent_amd
