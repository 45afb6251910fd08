// engine_host.h -- host-side classes of the engine (definitions in engine.hip, C-ABI in runtime_api.hip).
#pragma once
#include <condition_variable>
#include <functional>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "engine.h"
#include "launch.h"
#include "tune.h"

namespace magent_amd {

[[noreturn]] void fatal(const char *fmt, ...);

// std::default_random_engine of libstdc++ is minstd_rand0 (reference GridWorld.h:105): x <- 16807 x mod (2^31 - 1).
// seed(s): state = s mod m, 0 -> 1 (so seed(0) == seed(1)); outputs are in [1, 2^31 - 2].
struct MinStd {
    unsigned long long x = 1;
    void seed(unsigned long s) { x = s % 2147483647ul; if (x == 0) x = 1; }
    unsigned long long operator()() { x = (x * 16807ull) % 2147483647ull; return x; }
    void skip(unsigned n) {  // advance by n draws: x <- 16807^n x mod m
        unsigned long long b = 16807ull, a = x;
        while (n) { if (n & 1u) a = (a * b) % 2147483647ull; b = (b * b) % 2147483647ull; n >>= 1; }
        x = a;
    }
};

// bump allocator over large device blocks for the many small arrays of an environment (engine.hip: "device memory")
struct DevArena {
    static constexpr size_t BLOCK = 8u << 20, SMALL = 1u << 20;   // requests up to 1 MiB are carved from 8 MiB blocks
    std::vector<char *> blocks;
    size_t used = BLOCK;
    void *take(size_t bytes);
    bool owns(const void *p) const;
    void release();
};

struct HostRange {
    int width = 0, height = 0, count = 0, x1 = 0, y1 = 0, x2 = 0, y2 = 0;
    std::vector<unsigned char> in;
    std::vector<int> dx, dy;
    void circle(float radius, float inner_radius, int parity);
    void sector(float angle, float radius, int parity);
};

struct HostType {
    std::string name;
    int width = 1, length = 1;
    float speed = 1.0f, hp = 1.0f, view_radius = 1, view_angle = 360, attack_radius = 0, attack_angle = 0;
    float damage = 0, step_recover = 0, kill_supply = 0, food_supply = 0, eat_ability = 0;
    float step_reward = 0, kill_reward = 0, dead_penalty = 0, attack_penalty = 0;
    bool attack_in_group = false, can_absorb = false;
    int view_x_offset = 0, view_y_offset = 0, att_x_offset = 0, att_y_offset = 0;
    HostRange view, attack, move;
    int attack_base = 0, n_action = 0;
};

struct HostGroup {
    HostType *type = nullptr;
    int n = 0, cap = 0;
    GroupDev cur{}, alt{};       // alt holds the second copy of the arrays that clear_dead compacts
    TypeDev tdev{};
    float group_reward = 0;
    bool acted = false;          // set_action seen since the last step
    int h_dead = 0;              // dead_ct as of the last step (GridWorld.h Group::dead_ct)
    int h_taken = 0;             // movers taken in by goals: dead, but never counted in dead_ct (Map.cc:345)
    int indexed = 0;             // agents [0, indexed) have been through a clear_dead: Agent::index == position, else 0
    int sa_off = -1;             // this step's set_action call left its tile counts at d_asums[sa_off ...] (SeqPlan); -1: none / one-workgroup form
    PlainGroup pl{};             // scratch of the step of plain games (launch.h)
};

struct HostSymbol { int group = 0, index = 0; int ent_g = -1, ent_i = -1; };   // ent_*: the agent the host rule search last bound it to
struct HostRulePlan { std::vector<int> order, brings; };   // RewardRule::input_symbols / infer_obj (RewardEngine.cc:155-189); -1 = none
struct HostNode { int op = OP_NULL; std::vector<int> raw; };
struct HostRule { int on = 0; std::vector<int> recv; std::vector<float> val; bool terminal = false; };

// a few persistent worker threads that split one large memcpy (pinned staging -> caller's pageable buffer)
class CopyPool {
public:
    explicit CopyPool(int n_threads);
    ~CopyPool();
    void copy(void *dst, const void *src, size_t bytes);   // returns when done
private:
    void worker(int id);
    std::vector<std::thread> threads;
    std::mutex mu;
    std::condition_variable cv_go, cv_done;
    char *dst = nullptr; const char *src = nullptr; size_t bytes = 0;
    unsigned long long generation = 0;
    int pending = 0;
    bool stop = false;
};

class Env {
public:
    Env();
    ~Env();

    // reference interface (GridWorld.h:38-128)
    void set_config(const char *key, void *p);
    void register_agent_type(const char *name, int n, const char **keys, float *values);
    void new_group(const char *type_name, int *handle);
    void define_agent_symbol(int no, int group, int index);
    void define_event_node(int no, int op, int *inputs, int n);
    void add_reward_rule(int on, int *recv, float *val, int n, bool terminal);
    void reset();
    void add_agents(int group, int n, const char *method, const int *px, const int *py, const int *pdir);
    void set_goal(int group, const char *method);
    void observe_host(int g, float *view, float *feat);
    void set_action_host(int g, const int *actions);
    void step(int *done);
    void step_begin();            // enqueue the step without waiting (several environments can overlap)
    void step_end(int *done);     // wait for it
    void get_reward_host(int g, float *out);
    void clear_dead();
    void info_host(int g, const char *name, void *buf);
    void render();

    // device-resident extensions
    void observe_device(int g, float *view, float *feat, bool cells16 = false);   // cells16: `view` is bf16 [n][VH][VW][8] (RenderArgs::cells16)
    void set_action_device(int g, const int *d_act);
    void get_reward_device(int g, float *out);
    void info_device(int g, const char *name, void *out);
    // one environment cycle -- observe + set_action per group, step, rewards, clear_dead (examples/train_battle.py:61-109) --
    // in two launches for small worlds (k_render_multi, k_step_solo); NULL entries skip that call for that group
    void cycle(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, int *done);
    // ... and for many small environments in ONE pair of launches (one workgroup of k_step_solo_batch per environment)
    int group_count(int g) const { return g >= 0 && g < (int)groups.size() ? groups[g].n : 0; }
    static void cycle_many(Env **envs, int n_env, int n_group, float **view, float **feat, const int **actions, float **rewards, int *done,
                           const std::function<void(const std::vector<int> &)> &others);
    void sync();
    void profile_read(const char *name, int *n, float *ms);

    hipStream_t stream{};
    DevArena arena;
    std::shared_ptr<void> stream_owner;   // environments cycled together share one stream (Env::adopt_stream)
    // Side stream (large worlds): set_action and the part of the step that only READS the world -- the attack shuffle, the hit
    // gather and the death-rank fixed point -- run here, beside the observation renders on `stream` (engine.hip: side_stream)
    hipStream_t side{};
    hipEvent_t ev_state{}, ev_side{};
    bool overlap_enabled = false;         // MAGENT_TUNE overlap=1..3 turns it on (default: everything on `stream`, see engine.hip)
    int overlap_level = 3;                // (tuning) 1: set_action beside the renders, 2: + the attack shuffle, 3: + hit gather and death ranks
    bool side_dirty = false;              // work on `side` that `stream` has not waited for yet
    unsigned state_epoch = 1, marked_epoch = 0, side_epoch = 0;   // state-changing calls | ... covered by ev_state | ... waited for by `side`
    void enter();                         // head of every call that changes (or must see) the whole state: join + epoch
    void join_side();
    void mark_state();
    hipStream_t side_stream();
    bool side_wanted();
    hipStream_t action_stream();          // where the next set_action will read its actions
    int attack_round = 0;        // rounds of the attack fixed point launched in the current step (k_attack_eval)
    int prof_level = 0;          // 0 off, 1 every named phase, 2 only the observation render launches
    bool host_shuffle = false;
    int move_jump_batch = 3;
    int last_attack_iters = 0, last_move_iters = 0, fallback_steps = 0, fallback_attack = 0, fallback_move = 0, last_render_kernel = 0;
    bool checked_step = false;            // host-checked convergence instead of the single-sync driver
    // optimistic rounds of the single-sync driver: one pair / batch, two for 64 steps after a run-out (or fixed by env)
    int opt_attack_pairs = 1, opt_move_batches = 1, boost_attack = 0, boost_move = 0, boost_window = 64;
    bool boost_ran_out = false;           // the raised budget follows a step of this episode that ran out (not: the first 64 steps behind a reset)
    int round_hist[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};   // (env_get_info "round_hist")
    bool opt_fixed = false;
    // one-launch step of small worlds (k_step_solo): on by default, MAGENT_TUNE solo_step=0 keeps the multi-launch drivers
    bool solo_enabled = true;
    int solo_max_agents = 1536;          // ... for an environment stepping on its own; as one of a batch (env_cycle_many):
    int batch_solo_max = 16384;          // (Env::solo_ok: the measurements behind the two limits)
    double batch_us[4] = {0, 0, 0, 0};   // host time of env_cycle_many rounds led by this environment (info "batch_host_us")
    long batch_rounds = 0;
    int batch_width = 1;                 // environments rendered by the launch this one is part of (plan_render)

private:
    struct ProfSlot { std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; };
    struct ProfScope;
    std::map<std::string, ProfSlot> prof;
    std::vector<hipEvent_t> prof_pool;
    hipEvent_t prof_event();

    void use_device();
    void init_device();
    void free_group(HostGroup &g);
    void ensure_capacity(HostGroup &g, int need);
    void ensure_tables();
    WorldView view() const;
    int *read_counters();
    bool read_changed();
    void clear_changed();
    void compile_rules();
    void compile_rules_gpu();
    void plan_host_rules();
    void eval_rules_host();
    bool rules_on_host = false;           // some rule has a shape the kernels do not take: all rules run on the host
    std::vector<HostRulePlan> host_plans;
    std::vector<unsigned char> host_triggers;
    void compile_rule_program(size_t k);
    void enqueue_counters();
    bool step_pending = false, step_was_fast = false, step_was_solo = false, step_live_paint = false, live_paint_now = false;
    bool solo_ok(int total_n);
    void serial_add_call(int g, const int *d_act);
    void serial_step();
    std::vector<SerialCall> serial_calls;   // the calls of a step in which a group was given actions twice (k_step_serial)
    std::vector<int> step_calls;            // groups given actions in this step, in call order
    bool serial_calls_on = false;
    MiniArgs next_minimap();
    int *fold_counts();
    bool cycle_eligible(int n_group, float *const *view, float *const *feat, int *first_obs_out);
    bool cycle_prepare(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, BatchItem &item);
    void cycle_finish(int *done);
    void adopt_stream(Env &lead);
    bool cyc_next_mini = false, cyc_mini_skip = false; int cyc_mini_vh = 0, cyc_mini_vw = 0;
    BatchItem *batch_h = nullptr, *batch_d = nullptr; size_t batch_cap = 0;   // (lead environment of a batch)
    // many environments per launch through the pipeline of plain games (engine_batch.hip: "the pipeline, batched"; pipe.hip)
    bool pipe_eligible(int n_group, float *const *view, float *const *feat, const int *const *actions, int *total_out);
    void pipe_prepare(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, PipeItem &it, int rounds, bool sweep_ok);
    void pipe_after(float *const *rewards, const StepRecord &report, int *done);
    bool pipe_sweep_ok(float *const *view);
    // (lead environment of such a batch) the items, every environment's report on the device and in pinned memory, the last-workgroup ticket
    PipeItem *pipe_h = nullptr, *pipe_d = nullptr; StepRecord *reports_d = nullptr, *reports_h = nullptr; size_t pipe_cap = 0;
    int *pipe_ticket = nullptr, *pipe_flag = nullptr; int pipe_flag_seq = 0;
    int *d_newn = nullptr;                // [MAXG] group sizes behind the batched compaction (PipeItem::newn)
    bool pipe_folded = false;             // the batch's compaction makes the next minimap
    int pipe_sweep_shape = -1;            // bit g: group g's observation has the shape the sweeping render takes (-1: not looked at since the reset)
    int pipe_sweep_rounds = 0;            // ... with its observations written by the batch's sweeping render (k_pipe_render_sweep)
    int pipe_rounds = 0;                  // cycles this environment took through the batched pipeline (env_get_info "pipeline_stats")
    void wait_record(int seq);
    StepRecord *h_rec = nullptr;          // pinned: written by k_step_solo, spun on by step_end
    int step_seq = 0;
    int solo_nt_eval = 0;
    unsigned *d_hit = nullptr;            // per cell: hit bits / candidate-list heads of the one-launch step, zero between phases
    int2 *d_mvnodes = nullptr; size_t mvnodes_cap = 0;   // nodes of the generic move resolution's candidate lists (WorldView::mv_nodes)
    void move_nodes();
    bool claim_clean = false;             // every claim word is CLAIM_NONE (the one-launch step keeps it so)
    bool claim_epochs = false;            // ... or written only by steps of the plain pipeline in the current window of epochs (scratch_for)
    unsigned plain_epoch = 0;             // steps of the plain pipeline so far: claim-word epochs and round stamps derive from it
    bool hit_clean = false;               // every hit word is zero (both of those steps keep it so)
    void scratch_for(int path);
    int plain_steps = 0, plain_slots = 0;
    int pairs_two_steps = 0, pairs_one_steps = 0, claim_refills = 0;   // (env_get_info "pipeline_stats")
    bool plain_world = false, step_was_plain = false, step_fused_rules = false, ptab_valid = false;
    PlainGroup *d_ptab = nullptr;
    int *d_alive = nullptr; size_t alive_cap = 0;      // survivors per 256 agents, left by k_strike for clear_dead (PlainWorld::alive)
    int alive_off[MAXG] = {}, alive_n[MAXG] = {};
    bool alive_valid = false;
    bool map_scattered = false;           // a quarter or more of the agents were placed by add_agents("random"): neighbours in the group are not neighbours on the map
    long long placed_random = 0, placed_total = 0;
    bool map_warm = false;                // an observation render has walked the painted map since the last step (observe_device: touch_map)
    PlainWorld plain_view();
    void plain_arrays(HostGroup &g, size_t n, size_t cap);
    bool stale_events = false;            // a step has run since the last clear_dead: last_op / op_obj are not all OP_NULL / -1
    RuleArgs *d_rule_args = nullptr; RuleProg *d_rule_progs = nullptr;
    void shuffle_buffers(int n_max);
    void push_rng();
    ShuffleBufs shuffle_bufs() const;
    void attack_rounds_checked(const WorldView &W);
    void move_rounds_checked(const WorldView &W);
    void phase_tail(const WorldView &W, int from);
    bool rng_on_device = false;           // the device copy of the RNG state (CTR_RNG) is current
    void download_occ();
    void upload_occ();
    bool host_blank(int x, int y, int bw, int bl) const;
    void host_random_blank(int bw, int bl, int &ox, int &oy);
    void plan_render(int g, RenderArgs &R, RenderPlan &P, float *view, float *feat);
    bool prepare_render(int g, const WorldView &W, RenderArgs &R, RenderPlan &P, float *view, float *feat);
    long long mini_population(bool skip) const;
    MiniArgs mini_args(int vh, int vw, bool skip);
    bool mini_skip = false, solo_mini = false;
    void copy_out(void *host_dst, const void *dev_src, size_t bytes);
    void read_back(void *host_dst, const void *dev_src, size_t bytes);
    char *h_small = nullptr; size_t h_small_cap = 0;   // pinned bounce buffer of read_back
    int n_channel() const;
    int feature_size(int g) const;

    // configuration
    int width = 0, height = 0, embedding_size = 0, device_id = 0;
    bool minimap_mode = false, large_map_mode = false, food_mode = false, turn_mode = false, goal_mode = false;
    int bandwidth = 1, map_reach = 0;
    std::string render_dir;
    // text video dump (reference RenderGenerator.{h,cc}); host-side, off the hot path
    bool first_render = true;
    int file_ct = 0, frame_ct = 0, frame_per_file = 10000;
    struct AttackEvent { int id, x, y; };
    std::vector<AttackEvent> attack_events;
    int4 *d_events = nullptr; size_t events_cap = 0;
    int2 *serial_alist = nullptr; int4 *serial_mlist = nullptr; SerialCall *serial_dcalls = nullptr;   // scratch of the literal loop (serial_step)
    size_t serial_alist_cap = 0, serial_mlist_cap = 0, serial_dcalls_cap = 0;
    void gen_render_config();
    MinStd rng;
    std::map<std::string, HostType> types;
    std::vector<HostGroup> groups;
    std::vector<HostSymbol> symbols;
    std::vector<HostNode> nodes;
    std::vector<HostRule> rules;
    std::vector<RuleArgs> rule_args;
    std::vector<RuleProg> rule_progs;   // general single-iterator expressions (compile_rule_program)
    bool rules_compiled = false;
    int id_counter = 0, any_kill_supply = 0, any_multicell = 0, any_absorb = 0, move_seq_base = 0, attack_kmax = 1;

    // device state
    bool device_ready = false, tables_valid = false, paint_valid = false;
    // the minimap histogram depends on the view shape and on the positions only: both groups of a battle step share it
    bool mini_valid = false; int mini_vh = 0, mini_vw = 0; long long mini_pop = -1;
    size_t map_cells = 0;
    int *d_occ = nullptr;
    int2 *d_viewcell = nullptr;
    unsigned long long *d_claim = nullptr;
    float *d_food = nullptr;     // food_mode: per cell amount | attack-phase scratch
    int *d_counters = nullptr, *h_counters = nullptr;
    GroupDev *d_gtab = nullptr;
    TypeDev *d_ttab = nullptr;
    int2 *d_delta = nullptr;
    unsigned char *d_mask = nullptr;
    int *d_mini = nullptr; size_t mini_cap = 0;
    float *d_minif = nullptr; size_t minif_cap = 0;
    int *d_sums = nullptr; size_t sums_cap = 0;
    // attack counts per tile / per wave of the step's tiled set_action calls, in call order (k_set_action_a -> attack_seq)
    int *d_asums = nullptr, *d_wpre = nullptr; size_t asums_cap = 0, wpre_cap = 0;
    int sa_tiles = 0; bool step_sa_tiled = false;
    SeqPlan seq_plan() const;
    int *d_rank = nullptr, *h_rank = nullptr; size_t rank_cap = 0, hrank_cap = 0;
    int *d_shuf = nullptr; size_t shuf_cap = 0;
    unsigned *d_powtab = nullptr; size_t powtab_cap = 0;   // powers of 16807 for the shuffle draws
    int *d_actions = nullptr; size_t actions_cap = 0;
    float *d_stage_view = nullptr, *d_stage_feat = nullptr; size_t stage_view_cap = 0, stage_feat_cap = 0;
    unsigned char *d_stage_small = nullptr; size_t stage_small_cap = 0;
    std::vector<int> h_occ; bool h_occ_valid = false;
    // pipelined device -> pageable-host copy: pinned ring + worker threads
    static constexpr int COPY_RING = 3;
    static constexpr size_t COPY_CHUNK = 32u << 20;
    char *h_ring[COPY_RING] = {nullptr, nullptr, nullptr};
    hipEvent_t ring_ev[COPY_RING] = {};
    hipStream_t copy_stream{};
    CopyPool *pool = nullptr;
    std::vector<int> shuffle_perm;
};

}  // namespace magent_amd
