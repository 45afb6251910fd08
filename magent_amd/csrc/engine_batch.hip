// engine_batch.hip -- whole environment cycles in one call (env_cycle_many): the two-launch cycle of small worlds, many environments per launch
#include "engine_impl.h"

namespace magent_amd {

// ------------------------------------------------------------------------------------------------ one cycle, two launches
// Splits in three so that env_cycle_many can put MANY environments into one pair of launches:
//   cycle_prepare : eligibility, stale paint / minimap brought up to date, the launch descriptions of this environment
//   (the launches : Env::cycle for one environment, launch_cycle_batch for many)
//   cycle_finish  : the step record, the host mirror of what clear_dead did on the device
// can this environment's cycle run as the two-launch form (k_render_multi + k_step_solo)?  No device work: the batch asks
// before it decides whose stream an environment uses
bool Env::cycle_eligible(int n_group, float *const *view, float *const *feat, int *first_obs_out) {
    if (!device_ready) fatal("cycle called before reset");
    const int NG = (int)groups.size();
    if (n_group != NG) fatal("env_cycle_many: n_group (%d) differs from the number of groups (%d)", n_group, NG);
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    bool fused = solo_ok(total_n) && !step_pending && !serial_calls_on;   // (a group given actions twice: the literal loop, by the call sequence)
    // the observed groups must share one minimap (same window, same "skip absorbed" rule) to be rendered by one launch
    int n_obs = 0, first_obs = -1;
    for (int g = 0; g < NG && fused; g++) {
        if (!(view && view[g]) || groups[g].n == 0) continue;
        if (!feat || !feat[g] || (((uintptr_t)view[g]) & 15) || (((uintptr_t)feat[g]) & 15)) fused = false;
        if (first_obs < 0) first_obs = g;
        else if (minimap_mode && (groups[g].type->view.height != groups[first_obs].type->view.height ||
                                  groups[g].type->view.width != groups[first_obs].type->view.width ||
                                  groups[g].type->can_absorb != groups[first_obs].type->can_absorb)) fused = false;
        n_obs++;
    }
    if (n_obs > RENDER_MULTI_MAX) fused = false;
    if (first_obs_out) *first_obs_out = first_obs;
    return fused;
}

bool Env::cycle_prepare(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, BatchItem &item) {
    int first_obs = -1;
    if (!cycle_eligible(n_group, view, feat, &first_obs)) return false;
    // goals that are given actions may move: the call sequence (set_action_device sends such a step through the literal loop)
    for (int g = 0; actions && g < n_group && g < (int)groups.size(); g++) if (actions[g] && groups[g].type->can_absorb && groups[g].n > 0) return false;
    enter();
    move_nodes();
    alive_valid = false;            // (the one-launch cycle compacts by itself)
    const int NG = (int)groups.size();
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    WorldView &W = item.W;
    W = this->view();
    // ---- launch 1: the observations of every observed group
    RenderMulti &M = item.M;
    M = RenderMulti{};
    for (int g = 0; g < NG; g++) {
        if (!(view && view[g]) || groups[g].n == 0) continue;
        if (groups[g].acted) {      // env_set_action_device came first: the feature rows show the new last_action (as in observe_device)
            GroupDev G = groups[g].cur; G.n = groups[g].n;
            launch_commit_action(stream, G, groups[g].tdev);
        }
        const int k = M.n++;
        prepare_render(g, W, M.R[k], M.P[k], view[g], feat[g]);
        M.blocks[k] = M.P[k].spans + M.P[k].feat_blocks;
    }
    // ---- launch 2: set_action, step, get_reward, clear_dead, the next minimap
    shuffle_buffers(total_n);
    push_rng();
    scratch_for(0);
    {   // groups that were given their actions by env_set_action_device before this call (a NULL entry in `actions`): when the world was
        // beyond the one-launch step's limit for an environment on its own (but within the batch's), that call left tile counts -- the
        // sequence numbers and the attack list's length are written out here, as Env::step_begin does
        bool first = true;
        for (int g = 0; g < NG; g++)
            if (groups[g].sa_off >= 0) { launch_seq_assign(stream, W, g, d_asums, d_wpre, groups[g].sa_off, first); first = false; groups[g].sa_off = -1; }
        step_calls.clear();
    }
    step_live_paint = live_paint_now = paint_valid;
    W.live_paint = step_live_paint ? 1 : 0;
    const ShuffleBufs B = shuffle_bufs();
    SoloStep &S = item.S;
    S = SoloStep{};
    S.shead = B.head; S.sfirst = B.first; S.sj = B.j; S.slink = B.link;
    S.rank = d_rank; S.powtab = d_powtab; S.hit = d_hit;
    S.rules = d_rule_args; S.progs = d_rule_progs; S.n_rules = (int)rule_args.size();
    S.kmax = attack_kmax; S.nt_eval = solo_nt_eval; S.max_rounds = 1 << 20;
    S.rec = h_rec; S.seq = ++step_seq;
    for (int g = 0; g < NG; g++) {
        HostGroup &G = groups[g];
        if (actions && actions[g]) {
            if (G.acted) fatal("set_action called twice for group %d before step: the reference would execute both action lists; unsupported", g);
            G.acted = true;
            if (G.n > 0) { S.actions[g] = actions[g]; S.call_base[g] = move_seq_base; move_seq_base += G.n; }
        }
        if (rewards && rewards[g] && G.n > 0) { S.rewards[g] = rewards[g]; S.group_reward[g] = G.group_reward; }
        S.dst[g] = {G.alt.x, G.alt.y, G.alt.id, G.alt.last_action, G.alt.hp, G.alt.next_reward, G.alt.last_reward, G.alt.absorbed, G.alt.dir};
    }
    S.do_clear = 1;
    S.gtab_out = d_gtab; S.ttab_out = d_ttab;
    cyc_next_mini = false;
    if (minimap_mode && first_obs >= 0) {   // the next cycle observes the same groups: its minimap is made here
        const HostType &t = *groups[first_obs].type;
        S.mini = mini_args(t.view.height, t.view.width, t.can_absorb);
        cyc_next_mini = true; cyc_mini_vh = S.mini.vh; cyc_mini_vw = S.mini.vw; cyc_mini_skip = S.mini.skip != 0;
    }
    return true;
}

void Env::cycle_finish(int *done) {
    use_device();
    const int NG = (int)groups.size();
    wait_record(step_seq);
    const StepRecord &r = *h_rec;
    if (r.error) fatal("%s resolution did not converge", r.error == 1 ? "attack" : "move");
    if (r.unsupported) fatal("internal: a can_absorb agent moved on the parallel path (a set_action for goals switches the step to the literal loop)");
    if (r.pack_overflow) fatal("internal: hp / type.hp outside [0, 2) met the packed view-cell format");
    if (r.bad_action) fatal("set_action: an action outside [0, n_action) (the reference indexes its tables out of range here)");
    if (r.hit_overflow) fatal("a target collected more attack hits than the engine's hit lists hold (256)");
    if (rng_on_device) rng.x = r.rng;
    last_attack_iters = r.rounds_attack; last_move_iters = r.rounds_move; attack_round = r.rounds_attack;
    int live = 0;
    for (int g = 0; g < NG; g++) {
        HostGroup &G = groups[g];
        G.acted = false;
        G.group_reward = 0;
        if (G.n - r.dead[g] > 0) live++;
        const int gone = r.dead[g] + r.taken[g];
        if (gone > 0 && G.n > 0) {   // the survivors' arrays have changed places
            std::swap(G.cur.x, G.alt.x); std::swap(G.cur.y, G.alt.y); std::swap(G.cur.id, G.alt.id);
            std::swap(G.cur.hp, G.alt.hp); std::swap(G.cur.last_action, G.alt.last_action);
            std::swap(G.cur.last_reward, G.alt.last_reward); std::swap(G.cur.next_reward, G.alt.next_reward);
            std::swap(G.cur.absorbed, G.alt.absorbed); std::swap(G.cur.dir, G.alt.dir);
            G.n -= gone;
        }
        G.h_dead = 0; G.h_taken = 0;
        G.indexed = G.n;
    }
    *done = live < NG;   // GridWorld.cc:619-624
    for (size_t k = 0; k < rules.size(); k++) if (((r.triggers >> k) & 1ull) && rules[k].terminal) *done = 1;
    stale_events = false;      // (the cycle's own clear_dead has reset every last_op)
    move_seq_base = 0;
    h_occ_valid = false;
    tables_valid = true;
    paint_valid = step_live_paint;
    live_paint_now = false;
    mini_valid = cyc_next_mini;
    if (cyc_next_mini) { mini_vh = cyc_mini_vh; mini_vw = cyc_mini_vw; mini_skip = cyc_mini_skip; mini_pop = mini_population(cyc_mini_skip); }
}

void Env::cycle(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, int *done) {
    static thread_local BatchItem item;
    if (!cycle_prepare(n_group, view, feat, actions, rewards, item)) {   // the general path: the same calls one after the other
        const int NG = (int)groups.size();
        for (int g = 0; g < NG; g++) {
            if (view && view[g]) observe_device(g, view[g], feat[g]);
            if (actions && actions[g]) set_action_device(g, actions[g]);
        }
        step(done);
        for (int g = 0; g < NG; g++) if (rewards && rewards[g]) get_reward_device(g, rewards[g]);
        clear_dead();
        // env_cycle_many promises finished outputs at return (the two-launch form waits for its step record, published after
        // everything else): here the rewards and the compaction are still queued -- wait for them (microseconds against a
        // large world's cycle)
        HIP_OK(hipStreamSynchronize(stream));
        return;
    }
    {
        ProfScope p(*this, "render", true);
        launch_render_multi(stream, item.W, item.M);
    }
    {
        ProfScope p(*this, "step");
        launch_step_solo(stream, item.W, item.S);
    }
    HIP_OK(hipGetLastError());
    cycle_finish(done);
}

// many small environments, one pair of launches: every environment that can take the two-launch cycle is described in an
// item of a device array (one workgroup of k_step_solo_batch each); the others go one by one
// (others: called once the batch's launches are enqueued, with the list of environments that did NOT take the two-launch form --
// too large for the one-launch step, food_mode, rules on the host; they keep their own streams and the caller runs their
// ordinary cycles, on its host threads, while the batch is in flight)
void Env::cycle_many(Env **envs, int n_env, int n_group, float **view, float **feat, const int **actions, float **rewards, int *done,
                     const std::function<void(const std::vector<int> &)> &others) {
    // the batch shares the stream of its first eligible environment: launches need no cross-stream events.  Environments that
    // cannot join are not touched (they used to adopt the stream too and then ran one after the other)
    std::vector<char> eligible(n_env, 0);
    int lead_e = -1;
    for (int e = 0; e < n_env; e++) {
        const int o = e * n_group;
        envs[e]->batch_width = n_env;      // (solo_ok: the batch's limit; plan_render: the launch is shared)
        eligible[e] = envs[e]->cycle_eligible(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, nullptr);
        if (eligible[e] && lead_e < 0) lead_e = e;
    }
    std::vector<int> alone;
    if (lead_e < 0) {
        for (int e = 0; e < n_env; e++) { alone.push_back(e); envs[e]->batch_width = 1; }
        others(alone);
        return;
    }
    Env &lead = *envs[lead_e];
    lead.use_device();
    const auto t0 = std::chrono::steady_clock::now();
    for (int e = 0; e < n_env; e++) if (eligible[e] && e != lead_e) envs[e]->adopt_stream(lead);
    if ((size_t)n_env > lead.batch_cap) {
        HIP_OK(hipStreamSynchronize(lead.stream));
        if (lead.batch_h) HIP_OK(hipHostFree(lead.batch_h));
        dfree(lead.arena, lead.batch_d);
        lead.batch_cap = std::max<size_t>((size_t)n_env, lead.batch_cap * 2);
        HIP_OK(hipHostMalloc((void **)&lead.batch_h, sizeof(BatchItem) * lead.batch_cap, hipHostMallocDefault));
        HIP_OK(dev_malloc(lead.arena, &lead.batch_d, sizeof(BatchItem) * lead.batch_cap));
    }
    // item e describes environment e (an environment that cannot take the two-launch cycle leaves a skip marker and goes alone
    // below).  A description costs ~0.2 us of host time (measured: 28 us for 128 environments) -- sharing them out over threads
    // cost more than it saved.
    std::vector<char> in_batch(n_env, 0);
    for (int e = 0; e < n_env; e++) {
        const int o = e * n_group;
        BatchItem &it = lead.batch_h[e];
        in_batch[e] = eligible[e] && envs[e]->device_id == lead.device_id &&
                      envs[e]->cycle_prepare(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, actions ? actions + o : nullptr,
                                             rewards ? rewards + o : nullptr, it);
        if (!in_batch[e]) { it.M.n = 0; it.S.rec = nullptr; }
    }
    const auto t1 = std::chrono::steady_clock::now();
    int slots = 0, max_blocks = 0, n_in = 0;
    size_t render_lds = 0, step_lds = 0;
    for (int e = 0; e < n_env; e++) {
        if (!in_batch[e]) continue;
        n_in++;
        const BatchItem &it = lead.batch_h[e];
        slots = std::max(slots, it.M.n);
        for (int q = 0; q < it.M.n; q++) { max_blocks = std::max(max_blocks, it.M.blocks[q]); render_lds = std::max(render_lds, render_strip_lds(it.M.P[q])); }
        step_lds = std::max(step_lds, solo_step_lds(it.W, it.S));
    }
    if (n_in > 0) {
        lead.use_device();
        HIP_OK(hipMemcpyAsync(lead.batch_d, lead.batch_h, sizeof(BatchItem) * (size_t)n_env, hipMemcpyHostToDevice, lead.stream));
        launch_cycle_batch(lead.stream, lead.batch_d, n_env, slots, max_blocks, render_lds, step_lds);
        HIP_OK(hipGetLastError());
    }
    for (int e = 0; e < n_env; e++) if (!in_batch[e]) { alone.push_back(e); envs[e]->batch_width = 1; }
    if (!alone.empty()) others(alone);
    const auto t2 = std::chrono::steady_clock::now();
    auto t3 = t2;
    bool first = true;
    for (int e = 0; e < n_env; e++) {
        if (!in_batch[e]) continue;
        envs[e]->cycle_finish(&done[e]);
        envs[e]->batch_width = 1;          // (whatever is called on the environment next is called on it alone)
        if (first) { t3 = std::chrono::steady_clock::now(); first = false; }
    }
    const auto t4 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    lead.batch_us[0] += us(t0, t1); lead.batch_us[1] += us(t1, t2); lead.batch_us[2] += us(t2, t3); lead.batch_us[3] += us(t3, t4);
    lead.batch_rounds++;
}

// every environment of a batch shares the first one's stream (kept alive by whoever still uses it)
void Env::adopt_stream(Env &lead) {
    if (stream == lead.stream) return;
    if (!device_ready || !lead.device_ready) fatal("env_cycle_many called before reset");
    if (device_id != lead.device_id) return;
    use_device();
    HIP_OK(hipStreamSynchronize(stream));
    stream_owner = lead.stream_owner;
    stream = lead.stream;
}

}  // namespace magent_amd
