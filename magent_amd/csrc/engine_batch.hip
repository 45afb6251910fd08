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
// item of a device array (one workgroup of k_step_solo_batch each); worlds beyond the one-launch step whose game the pipeline of plain
// games takes share ONE chain of launches (pipe.hip); the others go one by one
// (others: called once the batches' launches are enqueued, with the list of environments that took neither form -- food_mode, rules on the
// host, generic bodies beyond the one-launch step ...; they keep their own streams and the caller runs their ordinary cycles, on its host
// threads, while the batches are in flight)
void Env::cycle_many(Env **envs, int n_env, int n_group, float **view, float **feat, const int **actions, float **rewards, int *done,
                     const std::function<void(const std::vector<int> &)> &others) {
    // Which form does every environment take?  kind 1: the two-launch cycle (one workgroup steps the world); kind 2: the batched pipeline.
    // A world that could take either goes to the pipeline from `batch_pipe_min` agents on (measured on the MI355X, profiles/r06_summary.md:
    // one workgroup steps 4000 agents in ~0.3 ms whatever else the chip does; 32 such worlds through the pipeline share ~0.2 ms)
    static const int pipe_min = std::max(1, tune("batch_pipe_min", 1537));
    static const bool pipe_on = tune("batch_pipe", 1) != 0;
    std::vector<char> kind(n_env, 0), solo_too(n_env, 0);
    std::vector<int> alone, totals(n_env, 0);
    int n_pipe = 0;
    for (int e = 0; e < n_env; e++) {
        const int o = e * n_group;
        envs[e]->batch_width = n_env;      // (solo_ok: the batch's limit; plan_render: the launch is shared)
        const bool solo = envs[e]->cycle_eligible(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, nullptr);
        int total = 0;
        const bool pipe = pipe_on && envs[e]->pipe_eligible(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, actions ? actions + o : nullptr, &total);
        kind[e] = pipe && (!solo || total >= pipe_min) ? 2 : solo ? 1 : 0;
        solo_too[e] = solo; totals[e] = total;
        n_pipe += kind[e] == 2;
    }
    if (n_pipe == 1)       // (a batch of one: the environment's own launches do the same with less ceremony -- the one-workgroup step only
        // below the limit an environment on its own has for it, Env::solo_ok: beyond it the ordinary launches are the faster ones)
        for (int e = 0; e < n_env; e++) if (kind[e] == 2) kind[e] = solo_too[e] && totals[e] <= envs[e]->solo_max_agents ? 1 : 0;
    int lead_e = -1, lead_p = -1;
    for (int e = 0; e < n_env; e++) {
        if (kind[e] == 1 && lead_e < 0) lead_e = e;
        if (kind[e] == 2 && lead_p < 0) lead_p = e;
    }
    // every batch shares the stream of its first environment: launches need no cross-stream events; the two batches overlap on the device.
    // Environments that join neither are not touched (they used to adopt the stream too and then ran one after the other)
    for (int e = 0; e < n_env; e++) {
        if (kind[e] == 1 && e != lead_e) { if (envs[e]->device_id == envs[lead_e]->device_id) envs[e]->adopt_stream(*envs[lead_e]); else kind[e] = 0; }
        if (kind[e] == 2 && e != lead_p) { if (envs[e]->device_id == envs[lead_p]->device_id) envs[e]->adopt_stream(*envs[lead_p]); else kind[e] = 0; }
    }
    const auto t0 = std::chrono::steady_clock::now();
    // ---------------- the two-launch cycle of small worlds, all of them in one pair of launches
    if (lead_e >= 0) {
        Env &lead = *envs[lead_e];
        lead.use_device();
        if ((size_t)n_env > lead.batch_cap) {
            HIP_OK(hipStreamSynchronize(lead.stream));
            if (lead.batch_h) HIP_OK(hipHostFree(lead.batch_h));
            dfree(lead.arena, lead.batch_d);
            lead.batch_cap = std::max<size_t>((size_t)n_env, lead.batch_cap * 2);
            HIP_OK(hipHostMalloc((void **)&lead.batch_h, sizeof(BatchItem) * lead.batch_cap, hipHostMallocDefault));
            HIP_OK(dev_malloc(lead.arena, &lead.batch_d, sizeof(BatchItem) * lead.batch_cap));
        }
        // item e describes environment e (an environment that does not take the two-launch cycle leaves a skip marker).  A description costs
        // ~0.2 us of host time (measured: 28 us for 128 environments) -- sharing them out over threads cost more than it saved.
        int slots = 0, max_blocks = 0, n_in = 0;
        size_t render_lds = 0, step_lds = 0;
        for (int e = 0; e < n_env; e++) {
            const int o = e * n_group;
            BatchItem &it = lead.batch_h[e];
            if (kind[e] == 1 && !envs[e]->cycle_prepare(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, actions ? actions + o : nullptr,
                                                       rewards ? rewards + o : nullptr, it)) kind[e] = 0;
            if (kind[e] != 1) { it.M.n = 0; it.S.rec = nullptr; continue; }
            n_in++;
            slots = std::max(slots, it.M.n);
            for (int q = 0; q < it.M.n; q++) { max_blocks = std::max(max_blocks, it.M.blocks[q]); render_lds = std::max(render_lds, render_strip_lds(it.M.P[q])); }
            step_lds = std::max(step_lds, solo_step_lds(it.W, it.S));
        }
        if (n_in > 0) {
            HIP_OK(hipMemcpyAsync(lead.batch_d, lead.batch_h, sizeof(BatchItem) * (size_t)n_env, hipMemcpyHostToDevice, lead.stream));
            launch_cycle_batch(lead.stream, lead.batch_d, n_env, slots, max_blocks, render_lds, step_lds);
            HIP_OK(hipGetLastError());
        }
    }
    const auto t1 = std::chrono::steady_clock::now();
    // ---------------- the pipeline, batched: one launch per phase for every world beyond the one-launch step (pipe.hip)
    PipeDims PD{};
    std::vector<int> piped;
    if (lead_p >= 0) {
        Env &lead = *envs[lead_p];
        lead.use_device();
        if ((size_t)n_env > lead.pipe_cap) {
            HIP_OK(hipStreamSynchronize(lead.stream));
            if (lead.pipe_h) HIP_OK(hipHostFree(lead.pipe_h));
            if (lead.reports_h) HIP_OK(hipHostFree(lead.reports_h));
            dfree(lead.arena, lead.pipe_d); dfree(lead.arena, lead.reports_d);
            lead.pipe_cap = std::max<size_t>((size_t)n_env, lead.pipe_cap * 2);
            HIP_OK(hipHostMalloc((void **)&lead.pipe_h, sizeof(PipeItem) * lead.pipe_cap, hipHostMallocDefault));
            HIP_OK(hipHostMalloc((void **)&lead.reports_h, sizeof(StepRecord) * lead.pipe_cap, hipHostMallocDefault));
            HIP_OK(dev_malloc(lead.arena, &lead.pipe_d, sizeof(PipeItem) * lead.pipe_cap));
            HIP_OK(dev_malloc(lead.arena, &lead.reports_d, sizeof(StepRecord) * lead.pipe_cap));
            if (!lead.pipe_ticket) {
                HIP_OK(dev_malloc(lead.arena, &lead.pipe_ticket, sizeof(int)));
                HIP_OK(hipMemsetAsync(lead.pipe_ticket, 0, sizeof(int), lead.stream));
                HIP_OK(hipHostMalloc((void **)&lead.pipe_flag, sizeof(int), hipHostMallocDefault));
                *lead.pipe_flag = 0;
            }
        }
        // every environment launches as many rounds as the one with the largest budget (a round that has nothing to do returns at once)
        // (MAGENT_TUNE attack_pairs=N fixes the budget for the process: 0 leaves every attack phase to the host)
        // THREE rounds, four while some environment's budget is raised (Env::step_end: for its window after a step of the episode ran out -- not for the first 64 steps behind a reset, the single environment's precaution: three rounds already are one more than its one pair): a round is a launch of ~5 us for the whole
        // batch whatever it finds to do, and the batch runs as many as its neediest environment -- so the single environment's 2-or-4 is the wrong
        // grain here.  Measured (32 x battle 200 x 200, 2 x 2000, 9920 environment steps, tools/many_envs_pipe.py): the last round that changes a
        // death rank is round 0 or 1 in 99.9 % of the steps and round 2 in the rest, never later; with two rounds every 34th round of the batch
        // had an environment run out (~0.27 ms of host each), with two-or-four one run-out kept all 32 environments at four for its window
        // (86 % of the steps).  Three rounds: no run-out seen, one launch less than four.
        int rounds = lead.opt_fixed ? 2 * lead.opt_attack_pairs : 3;
        for (int e = 0; e < n_env && !lead.opt_fixed; e++) if (kind[e] == 2 && envs[e]->boost_attack > 0 && envs[e]->boost_ran_out) rounds = 4;
        PD.G = n_group; PD.rounds = rounds;
        // the observations of the worlds that do not render for themselves (< 3.1 M window cells per group: pipe_own_cells), one launch: when every observed
        // group has the battle shape, the sweeping kernel with ~256 workgroups over all (environment, group) segments together -- its own
        // geometry, one workgroup per CU (measured on the MI355X, one box, in turn with the generic workgroups, profiles/r06_summary.md:
        // 32 x (2 x 2000) 0.226-0.236 ms per round against 0.233-0.240, 128 worlds 0.707-0.725 against 0.733-0.756, 8 worlds level) -- else,
        // or with MAGENT_TUNE pipe_sweep=0, the generic render's workgroups (pipe_sweep=N: N sweeping workgroups per segment)
        static const int sweep_tune = tune("pipe_sweep", -1);
        bool sweep_ok = sweep_tune != 0;
        for (int e = 0; e < n_env && sweep_ok; e++) if (kind[e] == 2) sweep_ok &= envs[e]->pipe_sweep_ok(view ? view + e * n_group : nullptr);
        int sweep_feat = 0, sweep_vhw = 0;
        long long sweep_steps = 0;
        for (int e = 0; e < n_env; e++) {
            if (kind[e] != 2) continue;
            const int o = e * n_group;
            PipeItem &it = lead.pipe_h[piped.size()];
            envs[e]->pipe_prepare(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, actions ? actions + o : nullptr, rewards ? rewards + o : nullptr, it, rounds, sweep_ok);
            it.rec = lead.reports_d + piped.size();
            piped.push_back(e);
            for (int g = 0; g < n_group; g++) PD.max_n = std::max(PD.max_n, it.W.grp[g].n);
            PD.max_total = std::max(PD.max_total, it.n_max);
            PD.kmax = std::max(PD.kmax, it.PW.kmax);
            PD.slots = std::max(PD.slots, it.M.n);
            PD.hist_cells = std::max(PD.hist_cells, it.Mi.vh * it.Mi.vw);
            for (int q = 0; q < it.M.n; q++) {
                PD.render_blocks = std::max(PD.render_blocks, sweep_ok ? it.M.P[q].feat_blocks : it.M.blocks[q]);
                PD.render_lds = std::max(PD.render_lds, sweep_ok ? render_sweep_lds(it.M.R[q].VH * it.M.R[q].VW, 7) : render_strip_lds(it.M.P[q]));
                sweep_feat = std::max(sweep_feat, it.M.P[q].feat_blocks);
                sweep_vhw = std::max(sweep_vhw, it.M.R[q].VH * it.M.R[q].VW);
                sweep_steps = std::max(sweep_steps, ((long long)it.M.R[q].n * it.M.R[q].VH * it.M.R[q].VW + 63) / 64);
            }
        }
        PD.n_env = (int)piped.size();
        if (sweep_ok && PD.slots > 0) {
            // `sweep` workgroups per (environment, group) segment: 256 over the whole launch (four times that while the segments are few:
            // 16 segments measured level at 16 and 64 each), at least 8 steps' worth each
            const int segs = PD.n_env * PD.slots;
            const int want = sweep_tune > 0 ? sweep_tune : segs <= 16 ? 1024 / segs : std::max(1, 256 / segs);
            PD.sweep = (int)std::max<long long>(1, std::min<long long>(want, (sweep_steps + 7) / 8));
        }
        PipeCtl C{lead.reports_d, lead.reports_h, lead.pipe_ticket, lead.pipe_flag, ++lead.pipe_flag_seq, PD.n_env};
        launch_pipe_upload(lead.stream, lead.pipe_h, lead.pipe_d, PD.n_env);
        launch_pipe_cycle(lead.stream, lead.pipe_d, PD, C);
        HIP_OK(hipGetLastError());
    }
    for (int e = 0; e < n_env; e++) if (kind[e] == 0) { alone.push_back(e); envs[e]->batch_width = 1; }
    if (!alone.empty()) others(alone);
    const auto t2 = std::chrono::steady_clock::now();
    auto t3 = t2;
    bool first = true;
    for (int e = 0; e < n_env; e++) {
        if (kind[e] != 1) continue;
        envs[e]->cycle_finish(&done[e]);
        envs[e]->batch_width = 1;          // (whatever is called on the environment next is called on it alone)
        if (first) { t3 = std::chrono::steady_clock::now(); first = false; }
    }
    if (lead_p >= 0) {
        Env &lead = *envs[lead_p];
        lead.use_device();
        // ONE word for the whole batch: k_pipe_finish's last workgroup sets it behind every environment's outputs and reports
        for (unsigned spins = 0; __atomic_load_n(lead.pipe_flag, __ATOMIC_ACQUIRE) != lead.pipe_flag_seq; spins++)
            if ((spins & 0x3FFF) == 0x3FFF) {
                const hipError_t q = hipStreamQuery(lead.stream);
                if (q == hipSuccess) { if (__atomic_load_n(lead.pipe_flag, __ATOMIC_ACQUIRE) == lead.pipe_flag_seq) break; fatal("the batched cycle finished without publishing its word"); }
                if (q != hipErrorNotReady) fatal("batched cycle failed: %s", hipGetErrorString(q));
            }
        if (first) { t3 = std::chrono::steady_clock::now(); first = false; }
        for (size_t k = 0; k < piped.size(); k++) {
            const int e = piped[k];
            envs[e]->pipe_after(rewards ? rewards + e * n_group : nullptr, lead.reports_h[k], &done[e]);
            envs[e]->batch_width = 1;
            envs[e]->pipe_rounds++;
        }
    }
    const auto t4 = std::chrono::steady_clock::now();
    auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    Env &acct = *envs[lead_e >= 0 ? lead_e : lead_p >= 0 ? lead_p : 0];
    acct.batch_us[0] += us(t0, t1); acct.batch_us[1] += us(t1, t2); acct.batch_us[2] += us(t2, t3); acct.batch_us[3] += us(t3, t4);
    acct.batch_rounds++;
}

// ------------------------------------------------------------------------------------------------ the pipeline, batched
// Can this environment's cycle go through the batched pipeline?  Plain games (Env::reset: plain_world) under the single-sync driver whose
// rules k_strike evaluates itself -- what battle and gather are -- with every observed group's buffers 16-byte aligned.  No device work.
bool Env::pipe_eligible(int n_group, float *const *view, float *const *feat, const int *const *actions, int *total_out) {
    if (!device_ready) fatal("cycle called before reset");
    const int NG = (int)groups.size();
    if (n_group != NG) fatal("env_cycle_many: n_group (%d) differs from the number of groups (%d)", n_group, NG);
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    if (total_out) *total_out = total_n;
    if (!plain_world || checked_step || host_shuffle || !first_render || serial_calls_on || step_pending || rules_on_host || stale_events || total_n == 0) return false;
    if (overlap_enabled) return false;
    if (!fused_rules(rule_args.data(), (int)rule_args.size())) return false;
    int n_obs = 0, first_obs = -1;
    for (int g = 0; g < NG; g++) {
        if (groups[g].acted) return false;                         // (given its actions by env_set_action_device already: the call sequence)
        if (!(view && view[g]) || groups[g].n == 0) continue;
        if (!feat || !feat[g] || (((uintptr_t)view[g]) & 15) || (((uintptr_t)feat[g]) & 15)) return false;
        if (first_obs < 0) first_obs = g;
        else if (minimap_mode && (groups[g].type->view.height != groups[first_obs].type->view.height || groups[g].type->view.width != groups[first_obs].type->view.width))
            return false;                                          // (one minimap per environment and cycle)
        n_obs++;
    }
    (void)actions;
    return n_obs <= RENDER_MULTI_MAX;
}

// are this environment's observed groups of the shape the sweeping render takes?  (Env::cycle_many: one launch form for the whole batch)
bool Env::pipe_sweep_ok(float *const *view) {
    if (pipe_sweep_shape < 0) {        // (a property of the game's configuration: decided once per reset, for every group)
        const WorldView W = this->view();
        pipe_sweep_shape = 0;
        for (int g = 0; g < (int)groups.size(); g++) {
            RenderArgs R; RenderPlan P;
            plan_render(g, R, P, nullptr, nullptr);
            if (render_sweep_mini_ok(W, R)) pipe_sweep_shape |= 1 << g;
        }
    }
    for (int g = 0; g < (int)groups.size(); g++)
        if (view && view[g] && groups[g].n > 0 && !((pipe_sweep_shape >> g) & 1)) return false;
    return true;
}

// window cells of a group from which an environment of a batch renders by launches of its own (MAGENT_TUNE pipe_own=N: N x 65536 cells).
// Measured on the MI355X, own launches / the batch's one launch (profiles/r06_summary.md): 16 worlds of 2 x 8000 (1.35 M cells per group) 0.413 / 0.349 ms
// per round, 8 worlds of 2 x 20,000 (3.4 M) 0.409 / 0.415, 8 gather worlds of 100k agents (4.9 M) 0.456 / 0.491 -- the line was at 1 M until then
static long long pipe_own_cells() {
    static const long long v = 65536ll * tune("pipe_own", 48);
    return v;
}

// Everything of one environment's cycle as an item of the batch: what observe_device, set_action_device, step_begin, get_reward_device and
// clear_dead would do on the host, with the launches left to the batch (launch_pipe_cycle).  Stale state that only the first cycle meets
// -- the painted map, the first minimap, tables, grown buffers -- is brought up to date by launches of the environment's own, on the
// batch's stream, ahead of the batch's.
void Env::pipe_prepare(int n_group, float *const *view, float *const *feat, const int *const *actions, float *const *rewards, PipeItem &it, int rounds, bool sweep_ok) {
    enter();
    const int NG = (int)groups.size();
    (void)n_group;
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    if (!tables_valid) ensure_tables();
    it.M = RenderMulti{};
    // ---- the observations: worlds whose renders fill the chip by themselves launch their own (launch_render picks the sweeping kernel
    // at that size); the others share the batch's render launch
    bool own_render = false;
    for (int g = 0; g < NG; g++)
        if (view && view[g] && groups[g].n > 0 && (long long)groups[g].n * groups[g].type->view.height * groups[g].type->view.width >= pipe_own_cells()) own_render = true;
    it.W = this->view();
    for (int g = 0; g < NG; g++) {
        if (!(view && view[g]) || groups[g].n == 0) continue;
        if (own_render) { observe_device(g, view[g], feat[g]); continue; }
        const int k = it.M.n++;
        prepare_render(g, it.W, it.M.R[k], it.M.P[k], view[g], feat[g]);
        it.M.blocks[k] = it.M.P[k].spans + it.M.P[k].feat_blocks;
    }
    if (sweep_ok && it.M.n > 0) pipe_sweep_rounds++;
    // ---- set_action: tile counts per call, in call order (Env::set_action_device)
    step_sa_tiled = true; sa_tiles = 0;
    step_calls.clear();
    for (int g = 0; g < MAXG; g++) { it.actions[g] = nullptr; it.call_base[g] = 0; it.P.off[g] = -1; it.rewards[g] = nullptr; it.group_reward[g] = 0; }
    for (int g = 0; g < NG; g++) {
        HostGroup &G = groups[g];
        if (!(actions && actions[g])) continue;
        G.acted = true;
        if (G.n == 0) continue;
        if ((long long)move_seq_base + G.n >= (1ll << 27)) fatal("more than 2^27 agents given actions in one step");
        it.actions[g] = actions[g]; it.call_base[g] = move_seq_base; it.P.off[g] = sa_tiles;
        move_seq_base += G.n;
        sa_tiles += (G.n + SCAN_TILE_HOST - 1) / SCAN_TILE_HOST;
    }
    if ((size_t)sa_tiles > asums_cap) {
        grow(arena, d_asums, asums_cap, (size_t)sa_tiles, stream);
        grow(arena, d_wpre, wpre_cap, asums_cap * (SCAN_TILE_HOST / 64), stream);
    }
    // ---- the step (Env::step_begin, single-sync driver, plain games, rules fused)
    step_live_paint = live_paint_now = paint_valid;
    step_pending = true;
    map_warm = false;
    step_was_fast = true; step_was_solo = false; step_was_plain = true; step_fused_rules = true;
    plain_steps++;
    scratch_for(1);
    shuffle_buffers(total_n);
    push_rng();
    attack_round = rounds;
    if (rounds >= 4) pairs_two_steps++; else if (rounds >= 2) pairs_one_steps++;
    it.W.live_paint = live_paint_now ? 1 : 0;       // (the one field of the view that the renders' bookkeeping above can change: Env::view)
    it.PW = plain_view();
    it.ptab = d_ptab; it.gtab = d_gtab; it.ttab = d_ttab;
    it.B = shuffle_bufs(); it.powtab = d_powtab;
    it.sums = d_asums; it.wpre = d_wpre;
    it.R = strike_rules(rule_args.data(), (int)rule_args.size());
    it.seq = ++step_seq;              // (it.rec: the batch's)
    it.n_max = total_n;
    alive_valid = true;
    stale_events = true;
    for (auto &g : groups) g.sa_off = -1;
    // ---- get_reward + clear_dead (Env::get_reward_device, Env::clear_dead: the form with k_strike's survivor counts; which groups compact
    // is decided on the device)
    it.A = ClearArgs{};
    for (int g = 0; g < NG; g++) {
        HostGroup &G = groups[g];
        if (rewards && rewards[g] && G.n > 0) { it.rewards[g] = rewards[g]; it.group_reward[g] = G.group_reward; }
        it.A.sums_off[g] = alive_off[g];
        it.A.dst[g] = {G.alt.x, G.alt.y, G.alt.id, G.alt.last_action, G.alt.hp, G.alt.next_reward, G.alt.last_reward, G.alt.absorbed, G.alt.dir};
    }
    it.A.sums_per_tile = SCAN_TILE_HOST / 256;
    it.alive_sums = d_alive;
    it.Mi = next_minimap();
    it.counts = fold_counts();
    pipe_folded = it.Mi.vh > 0;
    it.gtab_out = d_gtab; it.ttab_out = d_ttab;
    if (!d_newn) HIP_OK(dev_malloc(arena, &d_newn, sizeof(int) * MAXG));
    it.newn = d_newn;
    state_epoch++;
}

// The host's side of the end of a batched cycle: the step's report (Env::step_end: `done`, death counts, the generator -- and, when the
// optimistic rounds ran out, the continuation of the step), then the mirror of what k_pipe_clear / k_pipe_finish did on the device.  An
// environment whose step the host had to finish was skipped by those kernels: its rewards and its clear_dead are the ordinary calls.
void Env::pipe_after(float *const *rewards, const StepRecord &report, int *done) {
    static_assert(offsetof(StepRecord, marks) == PIPE_REPORT_BYTES && sizeof(StepRecord) % 16 == 0, "what k_pipe_finish sends of a report: everything ahead of the tuning marks");
    std::memcpy((void *)h_rec, (const void *)&report, PIPE_REPORT_BYTES);
    h_rec->n_marks = 0;
    h_rec->seq = step_seq;
    const int fallbacks = fallback_steps;
    step_end(done);
    const int NG = (int)groups.size();
    if (fallback_steps != fallbacks) {
        for (int g = 0; g < NG; g++) if (rewards && rewards[g]) get_reward_device(g, rewards[g]);
        clear_dead();
        HIP_OK(hipStreamSynchronize(stream));      // (env_cycle_many promises finished outputs at return)
        return;
    }
    enter();
    bool any = false;
    for (int g = 0; g < NG; g++) {
        HostGroup &G = groups[g];
        G.group_reward = 0;
        const int gone = G.h_dead + G.h_taken;
        if (gone > 0 && G.n > 0) {       // survivors: double-buffered arrays went to alt, the rest was reset in place
            std::swap(G.cur.x, G.alt.x); std::swap(G.cur.y, G.alt.y); std::swap(G.cur.id, G.alt.id);
            std::swap(G.cur.hp, G.alt.hp); std::swap(G.cur.last_action, G.alt.last_action);
            std::swap(G.cur.last_reward, G.alt.last_reward); std::swap(G.cur.next_reward, G.alt.next_reward);
            std::swap(G.cur.absorbed, G.alt.absorbed); std::swap(G.cur.dir, G.alt.dir);
            G.n -= gone;
            any = true;
        }
        G.h_dead = 0; G.h_taken = 0;
        G.indexed = G.n;
    }
    tables_valid = true;
    if (any) { h_occ_valid = false; mini_valid = false; }
    stale_events = false;
    alive_valid = false;
    if (pipe_folded) { mini_valid = true; mini_pop = mini_population(mini_skip); }
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
