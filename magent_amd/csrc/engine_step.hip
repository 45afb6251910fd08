// engine_step.hip -- set_action, the step drivers (one launch | single-sync pipeline | host-checked), get_reward, clear_dead of the host engine
// (GridWorld.cc:403-704)
#include "engine_impl.h"

namespace magent_amd {

// ------------------------------------------------------------------------------------------------ set_action
void Env::set_action_device(int g, const int *d_act) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in set_action : %d", g);
    use_device();
    HostGroup &G = groups[g];
    // a group is given actions again before the step: the reference appends (GridWorld.cc:403-454) -- or goals are given actions, which may
    // move them (Map::do_move treats a goal that has taken nobody in like any mover; the parallel move resolution rests on goals that
    // stand still): either way the step runs the reference's own loops on the device (k_step_serial)
    // ... unless every goal is told to stand still (the zero move): such a call is an ordinary one -- one small launch and one read-back
    // decide it, against about a microsecond per list entry of the whole world on the literal loop's single lane
    bool goals_act = G.type->can_absorb && G.n > 0 && !G.acted && !serial_calls_on;     // (an empty group of goals moves nobody)
    if (goals_act) {
        enter();
        int *flag = d_counters + CTR_GOALS_ACT;
        HIP_OK(hipMemsetAsync(flag, 0, sizeof(int), stream));
        launch_any_real_action(stream, d_act, G.n, G.tdev, d_delta, flag);
        int h_flag = 0;
        read_back(&h_flag, flag, sizeof(int));
        goals_act = h_flag != 0;
        static bool told = false;
        if (goals_act && !told) {
            told = true;
            std::fprintf(stderr, "magent-amd: a group of goals (can_absorb) was given actions that move, turn or attack: such steps run the reference's "
                                 "sequential loops on one lane of the device -- exact, about a microsecond per action of the whole world (INTEGRATION.md)\n");
        }
    }
    if (G.acted || serial_calls_on || goals_act) {
        serial_add_call(g, d_act);
        G.acted = true;
        return;
    }
    G.acted = true;
    if (step_calls.empty()) {           // the first call of a step fixes the form of all of them: worlds that step in one launch take the
        int total_n = 0;                // one-workgroup form (sequence numbers assigned at once), the others leave tile counts (SeqPlan)
        for (auto &q : groups) total_n += q.n;
        step_sa_tiled = !solo_ok(total_n);
        sa_tiles = 0;
    }
    step_calls.push_back(g);
    if (G.n == 0) return;
    int off = -1;
    if (step_sa_tiled) {
        const int nb = (G.n + SCAN_TILE_HOST - 1) / SCAN_TILE_HOST;
        off = sa_tiles;
        sa_tiles += nb;
        if ((size_t)sa_tiles > asums_cap) {      // (the counts of the step's earlier calls are kept)
            enter();
            grow(arena, d_asums, asums_cap, (size_t)sa_tiles, stream, true, (size_t)off);
            grow(arena, d_wpre, wpre_cap, asums_cap * (SCAN_TILE_HOST / 64), stream, true, (size_t)off * (SCAN_TILE_HOST / 64));
        }
    }
    G.sa_off = off;
    if ((long long)move_seq_base + G.n >= (1ll << 27)) fatal("more than 2^27 agents given actions in one step");   // (order keys: 27-bit insertion index, step.hip claim_word)
    hipStream_t s = action_stream();    // large worlds: beside the observation renders (see side_stream)
    ProfScope p(*this, "set_action", false, s);
    launch_set_action(s, view(), g, d_act, move_seq_base, d_asums, d_wpre, off);
    move_seq_base += G.n;
}

// Repeated set_action inside one step.  From the first repetition on every call of the step is kept as a list of (group, saved copy
// of the actions) in call order -- the earlier calls' actions are recovered from the pending actions they left -- and the step runs
// the reference's sequential loops on the device (k_step_serial).
void Env::serial_add_call(int g, const int *d_act) {
    enter();
    auto keep = [&](int gg, const int *src, bool from_pend) {
        HostGroup &G = groups[gg];
        int *buf = nullptr;
        if (G.n > 0) {
            HIP_OK(dev_malloc(arena, &buf, sizeof(int) * (size_t)G.n));
            GroupDev D = G.cur; D.n = G.n;
            if (from_pend) launch_pend_to_actions(stream, D, G.tdev, buf);
            else HIP_OK(hipMemcpyAsync(buf, src, sizeof(int) * (size_t)G.n, hipMemcpyDeviceToDevice, stream));
            // Agent::set_action stores last_action at once (GridWorld.h:176-178): an observation asked for before the step shows the latest call
            HIP_OK(hipMemcpyAsync(G.cur.last_action, buf, sizeof(int) * (size_t)G.n, hipMemcpyDeviceToDevice, stream));
        }
        serial_calls.push_back({gg, buf});
    };
    if (!serial_calls_on) {
        serial_calls_on = true;
        for (int gg : step_calls) keep(gg, nullptr, true);
    }
    keep(g, d_act, false);
}

void Env::serial_step() {
    WorldView W = view();
    size_t entries = 0;
    for (auto &c : serial_calls) entries += (size_t)groups[c.g].n;
    // (the lists' scratch is kept from one such step to the next; the attack events share the array of the checked driver's)
    grow(arena, d_events, events_cap, entries + 1, stream);
    grow(arena, serial_alist, serial_alist_cap, entries + 1, stream);
    grow(arena, serial_mlist, serial_mlist_cap, 2 * (entries + 1), stream);
    grow(arena, serial_dcalls, serial_dcalls_cap, serial_calls.size(), stream);
    int2 *alist = serial_alist; int4 *mlist = serial_mlist, *msorted = serial_mlist + (entries + 1), *events = d_events; SerialCall *d_calls = serial_dcalls;
    HIP_OK(hipMemcpyAsync(d_calls, serial_calls.data(), sizeof(SerialCall) * serial_calls.size(), hipMemcpyHostToDevice, stream));
    const int n_sep = large_map_mode ? (width + bandwidth - 1) / bandwidth : 0;
    if (n_sep >= 39) fatal("internal: too many move stripes for the serial step");
    push_rng();
    launch_step_serial(stream, W, d_calls, (int)serial_calls.size(), alist, mlist, msorted, n_sep, events);
    if (!rules_on_host) launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
    launch_step_report(stream, d_counters, h_rec, ++step_seq, (int)groups.size());
    HIP_OK(hipStreamSynchronize(stream));    // (the slow path: the scratch goes back at once)
    if (!first_render) {                     // attack events are recorded once rendering has started (GridWorld.cc:484,508)
        const int A = read_counters()[CTR_LAST_A];
        std::vector<int4> ev((size_t)std::max(A, 0));
        if (A > 0) read_back(ev.data(), events, sizeof(int4) * (size_t)A);
        attack_events.clear();
        for (const int4 &e : ev) if (e.w) attack_events.push_back({e.x, e.y, e.z});
    }
    for (auto &c : serial_calls) if (c.actions) { int *buf = const_cast<int *>(c.actions); dfree(arena, buf); }
    serial_calls.clear();
    serial_calls_on = false;
}

void Env::set_action_host(int g, const int *actions) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in set_action : %d", g);
    use_device();
    HostGroup &G = groups[g];
    hipStream_t s = stream;
    if (G.n > 0) {
        if ((size_t)G.n > actions_cap) { enter(); grow(arena, d_actions, actions_cap, (size_t)G.n, stream); }
        s = action_stream();
        HIP_OK(hipMemcpyAsync(d_actions, actions, sizeof(int) * G.n, hipMemcpyHostToDevice, s));
    }
    set_action_device(g, d_actions);
    HIP_OK(hipStreamSynchronize(s));   // d_actions is reused by the next call
}

// ------------------------------------------------------------------------------------------------ step
// GridWorld::step (GridWorld.cc:456-631).
//
// Two drivers over the same kernels:
//   * single-sync (default): every phase is enqueued without waiting for the device.  The attack list length and the
//     engine RNG state are read on the device; the fixed-point rounds of the attack and move phases are launched
//     optimistically (`opt_attack_pairs` pairs, `opt_move_batches` batches) and gated on the device -- a round
//     returns at once when its phase has converged.  ONE readback at the end returns `done`, the death counts, the
//     RNG state and whether a phase ran out of rounds; in that (rare) case everything after that phase has been
//     skipped on the device and the host continues from exactly that state with the checked driver.
//   * checked: the host reads the convergence flag after every pair / batch (also used while the text render is
//     recording attack events, and with MAGENT_TUNE host_shuffle=1 / checked_step=1 for A/B runs).
void Env::shuffle_buffers(int n_max) {
    grow(arena, d_rank, rank_cap, (size_t)n_max, stream);
    if ((size_t)n_max * 4 > shuf_cap) {   // four arrays: head | first | j | link
        grow(arena, d_shuf, shuf_cap, (size_t)n_max * 4, stream);
        shuf_cap -= shuf_cap % 4;
        // head and first are kept zero between steps (k_attack_rank clears what a step used)
        HIP_OK(hipMemsetAsync(d_shuf, 0, sizeof(int) * shuf_cap, stream));
    }
    int nb = (n_max + SCAN_TILE_HOST - 1) / SCAN_TILE_HOST;
    grow(arena, d_sums, sums_cap, (size_t)nb, stream);
    // powers of the minstd_rand0 multiplier for k_shuffle_draw: 16807^t (t < 256), then 16807^(256 h) up to h = n_max / 256 + 1
    const size_t need = 256 + (size_t)n_max / 256 + 2;
    if (need > powtab_cap) {
        grow(arena, d_powtab, powtab_cap, need, stream);
        std::vector<unsigned> tab(powtab_cap);
        const unsigned long long P = 2147483647ull;
        tab[0] = 1;
        for (int t = 1; t < 256; t++) tab[t] = (unsigned)(tab[t - 1] * 16807ull % P);
        const unsigned long long step = tab[255] * 16807ull % P;
        tab[256] = 1;
        for (size_t h = 257; h < powtab_cap; h++) tab[h] = (unsigned)(tab[h - 1] * step % P);
        HIP_OK(hipMemcpy(d_powtab, tab.data(), sizeof(unsigned) * powtab_cap, hipMemcpyHostToDevice));
    }
}

ShuffleBufs Env::shuffle_bufs() const {
    const size_t seg = shuf_cap / 4;
    return ShuffleBufs{d_shuf, d_shuf + seg, d_shuf + 2 * seg, d_shuf + 3 * seg};
}

void Env::push_rng() {
    if (rng_on_device) return;
    launch_set_rng(stream, d_counters, (unsigned)rng.x);
    rng_on_device = true;
}

// attack rounds, host-checked: pairs with ONE convergence check per pair (the flag of the second round)
void Env::attack_rounds_checked(const WorldView &W) {
    int iters = 0;
    while (true) {
        clear_changed();
        if (step_was_plain) {          // (the continuation of a step of the plain pipeline: its own rounds)
            const PlainWorld PW = plain_view();
            launch_plain_eval(stream, W, PW, d_ptab, d_gtab, d_ttab, ++attack_round, -1, shuffle_bufs());
            launch_plain_eval(stream, W, PW, d_ptab, d_gtab, d_ttab, ++attack_round, CTR_CHANGED, shuffle_bufs());
            iters += 2;
            if (!read_changed()) break;
            if (iters > 1000000) fatal("attack resolution did not converge");
            continue;
        }
        launch_attack_iter(stream, W, d_gtab, d_ttab, ++attack_round, attack_kmax, -1);
        launch_attack_iter(stream, W, d_gtab, d_ttab, ++attack_round, attack_kmax, CTR_CHANGED);
        iters += 2;
        if (!read_changed()) break;
        if (iters > 1000000) fatal("attack resolution did not converge");
    }
    last_attack_iters = iters;
}

// move rounds, host-checked: `move_jump_batch` rounds per convergence check (a resolved agent is a no-op later)
void Env::move_rounds_checked(const WorldView &W) {
    if (!any_multicell) { last_move_iters = 0; return; }   // one-cell bodies: the commit walks the dependency chains itself (move_resolve)
    int iters = 0;
    do {
        clear_changed();
        for (int k = 0; k < move_jump_batch; k++) {
            const int flag = k == move_jump_batch - 1 ? CTR_CHANGED : -1;
            launch_movg_sweep(stream, W, d_gtab, flag);
        }
        iters += move_jump_batch;
        if (iters > 1000000) fatal("move resolution did not converge");
    } while (read_changed());
    last_move_iters = iters;
}

void Env::phase_tail(const WorldView &W, int from /* 0 = after attack rounds, 1 = after move rounds */) {
    if (step_was_plain) {              // (only its attack rounds can run out: from == 0)
        launch_plain_tail(stream, W, plain_view(), d_ptab, d_gtab, d_ttab, step_fused_rules ? rule_args.data() : nullptr, (int)rule_args.size(), nullptr, 0);
        if (!step_fused_rules && !rules_on_host) launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
        return;
    }
    if (from == 0) {
        launch_attack_apply(stream, W, d_gtab, d_ttab, attack_kmax);
        if (any_multicell) launch_movg_prep(stream, W, true); else launch_move_prep(stream, W, d_gtab);   // (starve / recover first)
        move_rounds_checked(W);
    }
    if (any_multicell) launch_movg_apply(stream, W, d_gtab); else launch_move_apply(stream, W, d_gtab);
    if (!rules_on_host) launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
    if (any_multicell) launch_finish(stream, W);   // (the 1x1 move commit already consumed the pending actions)
}

// The per-cell scratch words (claim, hitbits) as the three step paths want them and leave them:
//   one-launch step / cycle (0): wants every claim word CLAIM_NONE and every hit word zero; keeps them so
//   step of plain games (1): does not use the hit words (its hits live in per-agent masks); its claim words carry the epoch of the step
//       that wrote them (step.hip: claim_word) and are never cleaned -- it wants every word either filled (all ones) or written by a
//       plain step of the current window of 63 epochs, so the array is refilled when a window begins and after any other path wrote it
//   everything else (2): wants nothing (fills what it needs) and leaves both arrays dirty
void Env::scratch_for(int path) {
    const size_t ncell = (size_t)width * height;
    if (path == 2) { claim_clean = claim_epochs = hit_clean = false; return; }
    if (path == 0 && !hit_clean) { HIP_OK(hipMemsetAsync(d_hit, 0, sizeof(unsigned) * ncell, stream)); hit_clean = true; }
    if (path == 1) {
        plain_epoch++;
        if (plain_epoch % 63u == 0) claim_epochs = false;      // a new window: the oldest words would look like this step's
    }
    if (!claim_clean && !(path == 1 && claim_epochs)) {
        HIP_OK(hipMemsetAsync(d_claim, 0xFF, sizeof(unsigned long long) * ncell, stream));
        claim_clean = true;
        if (path == 1) claim_refills++;
    }
    if (path == 0) claim_epochs = true;                // (filled is a special case of "filled or written in this window")
    else { claim_clean = false; claim_epochs = true; }
}

void Env::step(int *done) {
    step_begin();
    step_end(done);
}

// Worlds of up to `solo_max_agents` agents step in ONE launch (k_step_solo).  Not taken: food_mode (its per-cell food
// evaluation sweeps the map), hit lists that do not fit one workgroup's LDS, the A/B drivers, and steps that record attack
// events for the text render.
// Two limits (measured on the MI355X, profiles/r05_summary.md "one workgroup or a dozen launches"): an environment stepping on its own
// is faster through the multi-launch pipeline from ~1500 agents on in battle (2 x 1200: 0.093 ms per cycle against 0.141; 2 x 2000:
// 0.090 against 0.138; 2 x 8000: 0.109 against 0.389 -- one workgroup is one CU of 256) and level with it below; games with fewer
// fighters per agent (gather, pursuit) cross over later, at 2500-3000, and lose 0.015 ms per cycle to the lower limit there.  An
// environment that is one of a batch (env_cycle_many: one workgroup per environment, all in one launch) keeps the one-launch step up
// to 16384 agents -- the other CUs are busy with the other environments.
bool Env::solo_ok(int total_n) {
    return solo_enabled && !checked_step && !host_shuffle && !opt_fixed && first_render && !food_mode && !rules_on_host && total_n > 0 &&
           total_n <= (batch_width > 1 ? batch_solo_max : solo_max_agents) && solo_nt_eval >= 64;
}

// the host side of k_step_solo's report: spin on the sequence number in pinned memory (a stream synchronisation costs
// several times the PCIe write it waits for); the stream is polled now and then so that a failed launch cannot hang us
void Env::wait_record(int seq) {
    for (unsigned spins = 0;; spins++) {
        if (h_rec->seq == seq) break;
        if ((spins & 0x3FFF) == 0x3FFF) {
            hipError_t q = hipStreamQuery(stream);
            if (q == hipSuccess) {
                if (h_rec->seq == seq) break;
                fatal("the one-launch step finished without publishing its record");
            }
            if (q != hipErrorNotReady) fatal("step kernel failed: %s", hipGetErrorString(q));
        }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
}

void Env::enqueue_counters() {
    HIP_OK(hipMemcpyAsync(h_counters, d_counters, sizeof(int) * CTR_TOTAL, hipMemcpyDeviceToHost, stream));
}

// everything of the step that needs no answer from the device (single-sync driver), or the whole host-checked step
void Env::step_begin() {
    if (!device_ready) fatal("step called before reset");
    if (step_pending) fatal("step_begin called twice without step_end");
    use_device();
    if (!tables_valid) { ensure_tables(); state_epoch++; }   // (enqueued on `stream`: the side stream has to see it)
    step_live_paint = live_paint_now = paint_valid;   // the painted map is current: every driver of the step keeps it so
    move_nodes();
    WorldView W = view();
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    // (turn_mode with generic bodies: a third fixed point -- the turns -- between starvation and the moves; it runs under the
    // host-checked driver, or inside the one-launch step)
    const bool generic_turns = turn_mode && any_multicell;
    const bool fast = !checked_step && !host_shuffle && first_render && !generic_turns;
    step_pending = true;
    alive_valid = false;
    map_warm = false;
    step_was_fast = false;
    step_was_solo = false;
    step_was_plain = false;

    bool reported = false;      // (the plain pipeline with fused rules sends its report ahead of the moves)
    const bool beside = total_n > 0 && fast && side_wanted() && overlap_level >= 2 && !serial_calls_on;   // the read-only head of the step goes beside the renders
    if (!beside) join_side();
    step_calls.clear();
    if (total_n == 0) {
        enqueue_counters();
    } else if (serial_calls_on) {
        // ---------------- some group was given actions more than once: the reference's sequential loops, on the device
        scratch_for(2);
        step_was_fast = true;                    // (reports through the pinned record like the single-sync driver)
        step_live_paint = live_paint_now = false;   // the painted map is rebuilt by the next observation
        serial_step();
    } else if (solo_ok(total_n)) {
        // ---------------- one launch for the whole step
        step_was_solo = true;
        shuffle_buffers(total_n);
        push_rng();
        scratch_for(0);        // (fills after a multi-launch step or a reset: once)
        {                      // (given its actions in tiles, when the world was larger: the numbers, and the list's length, written out)
            bool first = true;
            for (size_t g = 0; g < groups.size(); g++)
                if (groups[g].sa_off >= 0) { launch_seq_assign(stream, W, (int)g, d_asums, d_wpre, groups[g].sa_off, first); first = false; }
        }
        const ShuffleBufs B = shuffle_bufs();
        SoloStep S{};
        S.shead = B.head; S.sfirst = B.first; S.sj = B.j; S.slink = B.link;
        S.rank = d_rank; S.powtab = d_powtab; S.hit = d_hit;
        S.rules = d_rule_args; S.progs = d_rule_progs; S.n_rules = (int)rule_args.size();
        S.kmax = attack_kmax; S.nt_eval = solo_nt_eval; S.max_rounds = 1 << 20;
        S.rec = h_rec; S.seq = ++step_seq;
        ProfScope p(*this, "step");
        launch_step_solo(stream, W, S);
    } else if (fast) {
        step_was_fast = true;
        // ---------------- single-sync driver
        const bool plain = W.plain != 0;       // plain games have a pipeline of their own behind the shuffle (step.hip: k_plain_rank ...)
        step_was_plain = plain;
        if (plain) plain_steps++;
        scratch_for(plain ? 1 : 2);
        PlainWorld PW{};
        if (plain) PW = plain_view();
        {
            const size_t caps = rank_cap + shuf_cap + sums_cap + powtab_cap;
            const bool rng_here = !rng_on_device;
            shuffle_buffers(total_n);
            push_rng();
            if (rng_here || caps != rank_cap + shuf_cap + sums_cap + powtab_cap) state_epoch++;   // (something was enqueued on `stream`)
        }
        // shuffle, hit gather and the death-rank fixed point only read the world (and write scratch no render looks at)
        hipStream_t a = beside ? side_stream() : stream;
        {
            ProfScope p(*this, "attack", false, a);
            // (plain games keep their hits in per-agent masks; otherwise the draw zero-fills the per-cell hit words)
            if (plain) launch_shuffle_draw(a, total_n, d_counters, shuffle_bufs(), d_powtab, step_sa_tiled);
            else launch_shuffle(a, total_n, d_counters, shuffle_bufs(), d_rank, d_hit, (size_t)width * height, d_powtab, step_sa_tiled);
            if (overlap_level == 2 && a != stream) { join_side(); a = stream; }
            attack_round = 0;
            const int pairs = opt_fixed ? opt_attack_pairs : (boost_attack > 0 ? 2 : 1);
            if (plain) { if (pairs >= 2) pairs_two_steps++; else if (pairs == 1) pairs_one_steps++; }
            // rounds after the first only touch agents whose inputs changed: they are launched back to back and the
            // LAST one reports whether anything still moved (one gate for all of them)
            if (plain) {
                launch_plain_rank(a, W, PW, d_ptab, shuffle_bufs(), d_asums, d_wpre, seq_plan());
                for (int r = 0; r < 2 * pairs; r++)
                    launch_plain_eval(a, W, PW, d_ptab, d_gtab, d_ttab, ++attack_round, r == 2 * pairs - 1 ? CTR_OPEN_ATTACK : -1, shuffle_bufs());
            } else {
                launch_attack_rank(a, W, d_gtab, d_rank, shuffle_bufs(), false, d_asums, d_wpre, seq_plan());
                for (int r = 0; r < 2 * pairs; r++)
                    launch_attack_iter(a, W, d_gtab, d_ttab, ++attack_round, attack_kmax, r == 2 * pairs - 1 ? CTR_OPEN_ATTACK : -1);
            }
            if (pairs == 0) launch_set_counter(a, d_counters, CTR_OPEN_ATTACK, 1, -1);   // tests: straight to the host
        }
        join_side();      // from here on the world changes: behind every render enqueued so far
        if (plain) {
            const bool fuse = step_fused_rules = !rules_on_host && !stale_events && fused_rules(rule_args.data(), (int)rule_args.size());
            {
                ProfScope p(*this, "move");
                // (with the rules fused -- or none -- nothing the report carries is decided behind k_strike: it goes out before the moves)
                static const bool early = tune("early_report", 1) != 0;          // (MAGENT_TUNE early_report=0: behind the moves, for A/B runs)
                reported = fuse && early;
                launch_plain_tail(stream, W, PW, d_ptab, d_gtab, d_ttab, fuse ? rule_args.data() : nullptr, (int)rule_args.size(),
                                  reported ? h_rec : nullptr, reported ? ++step_seq : 0);
            }
            if (!fuse && !rules_on_host) {
                ProfScope p(*this, "rules");
                launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
            }
            alive_valid = true;
        } else {
        {
            ProfScope p(*this, "attack");
            launch_attack_apply(stream, W, d_gtab, d_ttab, attack_kmax);
        }
        {
            ProfScope p(*this, "move");
            if (any_multicell) launch_movg_prep(stream, W, true); else launch_move_prep(stream, W, d_gtab);
            // (one-cell bodies need no rounds: the commit walks the dependency chains itself; the generic sweeps iterate)
            const int batches = opt_fixed ? opt_move_batches : (boost_move > 0 ? 2 : 1);
            for (int r = 0; any_multicell && r < batches * move_jump_batch; r++) {
                const int flag = r == batches * move_jump_batch - 1 ? CTR_OPEN_MOVE : -1;   // the last round reports
                launch_movg_sweep(stream, W, d_gtab, flag);
            }
            if (batches == 0) launch_set_counter(stream, d_counters, CTR_OPEN_MOVE, 1, CTR_OPEN_ATTACK);   // tests
            if (any_multicell) launch_movg_apply(stream, W, d_gtab); else launch_move_apply(stream, W, d_gtab);
        }
        {
            ProfScope p(*this, "rules");
            if (!rules_on_host) launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
            if (any_multicell) launch_finish(stream, W);
        }
        }
        if (!reported) launch_step_report(stream, d_counters, h_rec, ++step_seq, (int)groups.size());
    } else {
        // ---------------- checked driver
        scratch_for(2);
        HIP_OK(hipMemsetAsync(d_counters + CTR_OPEN_ATTACK, 0, 2 * sizeof(int), stream));
        int A = read_counters()[CTR_ATTACK];
        if (step_sa_tiled) for (int k = 0; k < ATT_SLOTS; k++) A += h_counters[att_slot(k)];   // (the tiled set_action's spread counters: k_shuffle_draw adds them up too)
        if (A > 0) {
            ProfScope p(*this, "attack");
            shuffle_buffers(std::max(A, total_n));
            if (host_shuffle) {   // the reference's literal loop on the host (MAGENT_TUNE host_shuffle=1, for A/B checks)
                if (rng_on_device) { rng.x = (unsigned)read_counters()[CTR_RNG]; }
                if ((size_t)A > hrank_cap) {
                    if (h_rank) HIP_OK(hipHostFree(h_rank));
                    hrank_cap = std::max<size_t>((size_t)A, hrank_cap * 2);
                    HIP_OK(hipHostMalloc((void **)&h_rank, sizeof(int) * hrank_cap, hipHostMallocDefault));
                }
                shuffle_perm.resize(A);
                for (int i = 0; i < A; i++) shuffle_perm[i] = i;
                for (int i = 0; i < A; i++) {
                    int j = (int)rng() % (i + 1);
                    std::swap(shuffle_perm[i], shuffle_perm[j]);
                }
                for (int pos = 0; pos < A; pos++) h_rank[shuffle_perm[pos]] = pos;
                HIP_OK(hipMemcpyAsync(d_rank, h_rank, sizeof(int) * A, hipMemcpyHostToDevice, stream));
                if (step_sa_tiled) launch_set_counter(stream, d_counters, CTR_ATTACK, A, -1);   // (k_shuffle_draw would have left the list's length there)
                rng_on_device = false;
            } else {              // exact parallel replay on the device
                push_rng();
                launch_shuffle(stream, total_n, d_counters, shuffle_bufs(), d_rank, d_hit, (size_t)width * height, d_powtab, step_sa_tiled);
            }
            launch_attack_rank(stream, W, d_gtab, d_rank, shuffle_bufs(), host_shuffle, d_asums, d_wpre, seq_plan());
            attack_round = 0;
            attack_rounds_checked(W);
            if (!first_render) {   // attack events are recorded once rendering has started (GridWorld.cc:484,508)
                grow(arena, d_events, events_cap, (size_t)A, stream);
                launch_attack_events(stream, W, d_events);
                std::vector<int4> ev(A);
                read_back(ev.data(), d_events, sizeof(int4) * A);
                attack_events.clear();
                for (const int4 &e : ev) if (e.w) attack_events.push_back({e.x, e.y, e.z});
            }
            launch_attack_apply(stream, W, d_gtab, d_ttab, attack_kmax);
        } else if (!first_render) attack_events.clear();
        if (generic_turns) {
            ProfScope p(*this, "turn");
            launch_turn_prep(stream, W);        // (starvation first)
            int iters = 0;
            do {
                clear_changed();
                for (int k = 0; k < move_jump_batch; k++) launch_turn_sweep(stream, W, d_gtab, k == move_jump_batch - 1 ? CTR_CHANGED : -1);
                iters += move_jump_batch;
                if (iters > 1000000) fatal("turn resolution did not converge");
            } while (read_changed());
            launch_turn_apply(stream, W);
        }
        {
            ProfScope p(*this, "move");
            if (any_multicell) launch_movg_prep(stream, W, !generic_turns); else launch_move_prep(stream, W, d_gtab);
            move_rounds_checked(W);
            if (any_multicell) launch_movg_apply(stream, W, d_gtab); else launch_move_apply(stream, W, d_gtab);
        }
        {
            ProfScope p(*this, "rules");
            if (!rules_on_host) launch_rules(stream, W, rule_args.data(), (int)rule_args.size(), rule_progs.data(), d_gtab);
            if (any_multicell) launch_finish(stream, W);
        }
        enqueue_counters();
    }
    stale_events = true;      // (last_op / op_obj hold this step's events until clear_dead resets them)
    for (auto &g : groups) g.sa_off = -1;
    state_epoch++;
}

PlainWorld Env::plain_view() {
    PlainWorld PW{};
    for (size_t g = 0; g < groups.size(); g++) PW.g[g] = groups[g].pl;
    PW.S = plain_slots; PW.kmax = attack_kmax;
    {   // where k_strike leaves its survivor counts (one per 256 agents, group after group)
        size_t total = 0;
        for (size_t g = 0; g < groups.size(); g++) { alive_off[g] = (int)total; alive_n[g] = groups[g].n; total += (size_t)(groups[g].n + 255) / 256; }
        grow(arena, d_alive, alive_cap, std::max<size_t>(total, 1), stream);
        PW.alive = d_alive;
        for (int g = 0; g < MAXG; g++) PW.alive_off[g] = g < (int)groups.size() ? alive_off[g] : 0;
    }
    PW.epoch = 62 - (int)(plain_epoch % 63u);
    PW.round_base = (int)(plain_epoch * 64u);            // (wraps after 2^26 steps: the stamps are compared modulo 2^32)
    if (!ptab_valid) {
        if (!d_ptab) HIP_OK(dev_malloc(arena, &d_ptab, sizeof(PlainGroup) * MAXG));
        HIP_OK(hipMemcpyAsync(d_ptab, PW.g, sizeof(PlainGroup) * MAXG, hipMemcpyHostToDevice, stream));   // (pageable source: the copy is done when the call returns)
        ptab_valid = true;
        state_epoch++;
    }
    return PW;
}

SeqPlan Env::seq_plan() const {
    SeqPlan P{};
    for (int g = 0; g < MAXG; g++) P.off[g] = g < (int)groups.size() ? groups[g].sa_off : -1;
    return P;
}

// the one host synchronisation of the step: `done`, death counts, RNG state, and the (rare) continuation when a
// phase ran out of optimistic rounds
void Env::step_end(int *done) {
    if (!step_pending) fatal("step_end without step_begin");
    step_pending = false;
    use_device();
    // the one-launch step and the single-sync driver both report through the pinned record
    if (step_was_solo || step_was_fast) {
        wait_record(step_seq);
        const StepRecord &r = *h_rec;
        if (!(r.open_attack | r.open_move)) {
            if (r.error) fatal("%s resolution did not converge", r.error == 1 ? "attack" : r.error == 2 ? "move" : "turn");
            if (r.unsupported) fatal("internal: a can_absorb agent moved on the parallel path (a set_action for goals switches the step to the literal loop)");
            if (r.pack_overflow) fatal("internal: hp / type.hp outside [0, 2) met the packed view-cell format");
            if (r.bad_action) fatal("set_action: an action outside [0, n_action) (the reference indexes its tables out of range here)");
            if (r.hit_overflow) fatal("a target collected more attack hits than the engine's hit lists hold (256)");
            if (rng_on_device) rng.x = r.rng;
            if (step_was_solo) { last_attack_iters = r.rounds_attack; last_move_iters = r.rounds_move; attack_round = r.rounds_attack; }
            else {
                if (boost_attack > 0) boost_attack--;
                if (boost_move > 0) boost_move--;
                if (rules_on_host) eval_rules_host();
                if (step_was_plain) { int hi = 0; for (int b = 1; b < 32; b++) if ((r.rounds_mask >> b) & 1u) hi = b; round_hist[std::min(hi, 8)]++; }
            }
            int live = 0;
            for (size_t g = 0; g < groups.size(); g++) {
                groups[g].h_dead = r.dead[g];
                groups[g].h_taken = r.taken[g];
                groups[g].acted = false;
                if (groups[g].n - groups[g].h_dead > 0) live++;
            }
            *done = live < (int)groups.size();   // GridWorld.cc:619-624
            for (size_t k = 0; k < rules.size(); k++)
                if ((rules_on_host ? host_triggers[k] != 0 : ((r.triggers >> k) & 1ull) != 0) && rules[k].terminal) *done = 1;
            move_seq_base = 0;
            h_occ_valid = false;
            paint_valid = step_live_paint; mini_valid = false;
            live_paint_now = false;
            return;
        }
        read_counters();     // a phase ran out of optimistic rounds: the whole counter block, for the continuation below
    }
    HIP_OK(hipStreamSynchronize(stream));
    const int *c = h_counters;
    if (step_was_fast) {
        if (c[CTR_OPEN_ATTACK] | c[CTR_OPEN_MOVE]) {   // continue from exactly the device state the open phase froze, host-checked
            WorldView W = view();
            const int phase = c[CTR_OPEN_ATTACK] ? 1 : 2;
            fallback_steps++;
            if (phase == 1) fallback_attack++; else fallback_move++;
            // deeper dependency chains around: one more batch.  A step that runs out while the budget is ONE pair doubles the time the budget
            // stays at two from then on (64, 128, ... steps): in a dense world -- the bench's 2 x 400k at random: 8 run-outs in the 70 one-pair
            // steps of a 550-step episode, profiles/r06_raw/c3_episode_550_steps.txt -- a run-out (a host round trip and the host-checked rounds)
            // costs ten times what the second pair does; a sparse one (config 5's formation: 486 one-pair steps, none ran out) keeps the saving
            if (phase == 1) { if (boost_attack == 0) boost_window = std::min(boost_window * 2, 4096); boost_attack = boost_window; boost_ran_out = true; } else boost_move = 64;
            HIP_OK(hipMemsetAsync(d_counters + CTR_OPEN_ATTACK, 0, 2 * sizeof(int), stream));   // both phase flags
            clear_changed();
            if (phase == 1) { attack_rounds_checked(W); phase_tail(W, 0); }
            else { move_rounds_checked(W); phase_tail(W, 1); }
            c = read_counters();
        }
        if (boost_attack > 0) boost_attack--;
        if (boost_move > 0) boost_move--;
    }
    if (rng_on_device) rng.x = (unsigned)c[CTR_RNG];   // the device advanced the engine state by A draws
    if (rules_on_host) eval_rules_host();

    int live = 0;
    for (size_t g = 0; g < groups.size(); g++) {
        groups[g].h_dead = 0;
        for (int k = 0; k < DEAD_SLOTS; k++) groups[g].h_dead += c[dead_slot((int)g, k)];
        groups[g].h_taken = c[CTR_TAKEN + g];
        groups[g].acted = false;
        if (groups[g].n - groups[g].h_dead > 0) live++;
    }
    if (c[CTR_UNSUPPORTED]) fatal("internal: a can_absorb agent moved on the parallel path (a set_action for goals switches the step to the literal loop)");
    if (c[CTR_PACK_OVERFLOW]) fatal("internal: hp / type.hp outside [0, 2) met the packed view-cell format");
    if (c[CTR_BAD_ACTION]) fatal("set_action: an action outside [0, n_action) (the reference indexes its tables out of range here)");
    if (c[CTR_HIT_OVERFLOW]) fatal("a target collected more attack hits than the engine's hit lists hold (256)");
    *done = live < (int)groups.size();   // GridWorld.cc:619-624
    for (size_t k = 0; k < rules.size(); k++)
        if ((rules_on_host ? host_triggers[k] != 0 : c[CTR_TRIGGER + k] != 0) && rules[k].terminal) *done = 1;
    // attack count and rule triggers are per step; dead_ct lives until clear_dead
    launch_step_reset(stream, d_counters);
    HIP_OK(hipGetLastError());
    move_seq_base = 0;
    h_occ_valid = false;
    paint_valid = step_live_paint; mini_valid = false;
    live_paint_now = false;
}

// ------------------------------------------------------------------------------------------------ reward / clear_dead
void Env::get_reward_device(int g, float *out) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_reward : %d", g);
    enter();
    GroupDev G = groups[g].cur; G.n = groups[g].n;
    launch_get_reward(stream, G, groups[g].group_reward, out);
}

void Env::get_reward_host(int g, float *out) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_reward : %d", g);
    int n = groups[g].n;
    if (n == 0) return;
    grow(arena, d_stage_small, stage_small_cap, (size_t)n * 8, stream);
    get_reward_device(g, (float *)d_stage_small);
    read_back(out, d_stage_small, sizeof(float) * n);
}

// GridWorld::clear_dead (GridWorld.cc:633-665)
void Env::clear_dead() {
    if (!device_ready) fatal("clear_dead called before reset");
    enter();
    ProfScope p(*this, "clear_dead");
    WorldView W = view();
    bool any = false, all_solo = true;
    for (auto &G : groups) { G.group_reward = 0; if (G.h_dead + G.h_taken > 0) { any = true; all_solo &= compact_is_solo(G.n); } }
    auto swap_buffers = [](HostGroup &G) {     // survivors: double-buffered arrays went to alt, the rest is reset in place
        std::swap(G.cur.x, G.alt.x); std::swap(G.cur.y, G.alt.y); std::swap(G.cur.id, G.alt.id);
        std::swap(G.cur.hp, G.alt.hp); std::swap(G.cur.last_action, G.alt.last_action);
        std::swap(G.cur.last_reward, G.alt.last_reward); std::swap(G.cur.next_reward, G.alt.next_reward);
        std::swap(G.cur.absorbed, G.alt.absorbed); std::swap(G.cur.dir, G.alt.dir);
        G.n -= G.h_dead + G.h_taken;
        G.h_dead = 0; G.h_taken = 0;
    };
    bool small_world = solo_enabled;
    for (auto &G : groups) small_world &= compact_is_solo(G.n);
    if (small_world) {           // one launch of one workgroup: compaction / init_reward of every group + the device tables
        ClearArgs A{};
        for (size_t g = 0; g < groups.size(); g++) {
            HostGroup &G = groups[g];
            A.mode[g] = G.n == 0 ? 0 : (G.h_dead + G.h_taken > 0 ? 2 : 1);
            A.dst[g] = {G.alt.x, G.alt.y, G.alt.id, G.alt.last_action, G.alt.hp, G.alt.next_reward, G.alt.last_reward, G.alt.absorbed, G.alt.dir};
        }
        // the observations that follow will want the minimap of the window they used last: made here, by the same launch
        const bool next_mini = minimap_mode && mini_vh > 0;
        MiniArgs M{};
        if (next_mini) M = mini_args(mini_vh, mini_vw, mini_skip);
        launch_clear_solo_all(stream, W, A, d_gtab, d_ttab, M);
        for (size_t g = 0; g < groups.size(); g++) if (A.mode[g] == 2) swap_buffers(groups[g]);
        tables_valid = true;
        solo_mini = next_mini;
    } else if (!any) {                  // Agent::init_reward for everybody: one launch (+ the normalisation of the next minimap)
        ClearArgs A{};
        for (size_t g = 0; g < groups.size(); g++) A.mode[g] = groups[g].n > 0 ? 1 : 0;
        const MiniArgs M = next_minimap();
        launch_clear_compact(stream, W, A, d_sums, M, fold_counts());
        if (M.vh > 0) { launch_mini_norm(stream, W, M, fold_counts()); solo_mini = true; }
    } else if (all_solo) {       // small worlds: one workgroup per group does everything for that group
        for (size_t g = 0; g < groups.size(); g++) {
            HostGroup &G = groups[g];
            if (G.h_dead + G.h_taken > 0) {
                GroupDev D = G.cur;
                D.x = G.alt.x; D.y = G.alt.y; D.id = G.alt.id; D.hp = G.alt.hp; D.last_action = G.alt.last_action;
                D.last_reward = G.alt.last_reward; D.next_reward = G.alt.next_reward; D.absorbed = G.alt.absorbed; D.dir = G.alt.dir;
                launch_compact(stream, W, (int)g, D, G.n - G.h_dead - G.h_taken, d_sums);
                swap_buffers(G);
            } else {
                launch_init_reward(stream, W, (int)g);
            }
        }
        if (any) tables_valid = false;
    } else {                     // three launches for all groups together
        ClearArgs A{};
        size_t nb_total = 0;
        for (size_t g = 0; g < groups.size(); g++) {
            HostGroup &G = groups[g];
            A.mode[g] = G.n == 0 ? 0 : (G.h_dead + G.h_taken > 0 ? 2 : 1);
            A.sums_off[g] = (int)nb_total;
            nb_total += (G.n + SCAN_TILE_HOST - 1) / SCAN_TILE_HOST;
            A.dst[g] = {G.alt.x, G.alt.y, G.alt.id, G.alt.last_action, G.alt.hp, G.alt.next_reward, G.alt.last_reward, G.alt.absorbed, G.alt.dir};
        }
        grow(arena, d_sums, sums_cap, nb_total, stream);
        const MiniArgs M = next_minimap();
        // (the last step was one of the plain pipeline and nothing was added since: k_strike has left the survivors of every 256 agents)
        bool counted = alive_valid;
        for (size_t g = 0; g < groups.size(); g++) counted &= groups[g].n == alive_n[g];
        if (counted) {
            for (size_t g = 0; g < groups.size(); g++) A.sums_off[g] = alive_off[g];
            A.sums_per_tile = SCAN_TILE_HOST / 256;
        }
        launch_clear_compact(stream, W, A, counted ? d_alive : d_sums, M, fold_counts());
        for (size_t g = 0; g < groups.size(); g++) if (A.mode[g] == 2) swap_buffers(groups[g]);
        launch_clear_finish(stream, view(), A, d_gtab, d_ttab, M, fold_counts());   // also refreshes the device tables
        tables_valid = true;
        if (M.vh > 0) solo_mini = true;
    }
    // (the death counters of the compacted groups were zeroed by the compaction kernels; the others were zero)
    if (any) { h_occ_valid = false; mini_valid = false; }
    for (auto &G : groups) G.indexed = G.n;   // Agent::set_index (GridWorld.cc:655)
    stale_events = false;
    alive_valid = false;                      // (k_strike's survivor counts describe the arrays as the step left them: consumed)
    if (solo_mini) { mini_valid = true; mini_pop = mini_population(mini_skip); solo_mini = false; }
}

}  // namespace magent_amd
