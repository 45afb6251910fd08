// engine_observe.hip -- get_observation (render planning, minimap upkeep), get_info and the text render dump of the host engine
// (GridWorld.cc:292-401, :709-894, RenderGenerator.cc)
#include "engine_impl.h"

namespace magent_amd {

// ------------------------------------------------------------------------------------------------ observation
void Env::plan_render(int g, RenderArgs &R, RenderPlan &P, float *view, float *feat) {
    const HostGroup &G = groups[g];
    const HostType &t = *G.type;
    const int NG = (int)groups.size();
    R = RenderArgs{};
    P = RenderPlan{};
    R.g = g; R.n = G.n;
    R.VH = t.view.height; R.VW = t.view.width; R.C = n_channel(); R.S = R.VH * R.VW * R.C;
    R.F = feature_size(g); R.E = embedding_size; R.NA = t.n_action;
    R.minimap = minimap_mode;
    R.turn = turn_mode ? 1 : 0;
    R.food = food_mode ? 1 : 0;
    R.scale_h = (height + R.VH - 1) / R.VH;   // GridWorld.cc:328-329
    R.scale_w = (width + R.VW - 1) / R.VW;
    // channel layout symmetric to every group (GridWorld.cc:897-913): block k belongs to group (g + k) % NG
    const int stride = minimap_mode ? 3 : 2;
    R.chan_desc[0] = (0 << 8) | (OCC_WALL & 0xff);
    for (int k = 0; k < NG; k++) {
        int j = (g + k) % NG, base = 1 + (food_mode ? 1 : 0) + k * stride;
        R.chan_desc[base] = (0 << 8) | j;
        R.chan_desc[base + 1] = (1 << 8) | j;
        if (minimap_mode) R.chan_desc[base + 2] = (2 << 8) | j;
    }
    for (int j = 0; j < NG; j++) R.totals[j] = groups[j].n;
    R.mini = d_minif;
    R.view = view; R.feat = feat;

    // flat decomposition of the n * VH * VW window cells into 64-cell wave steps, `steps_per_span` per workgroup
    if ((long long)R.n * R.VH * R.VW >= (1ll << 31) || (long long)R.n * R.F >= (1ll << 32))
        fatal("observation too large for 32-bit cell indexing");
    const long long steps = ((long long)R.n * R.VH * R.VW + 63) / 64;
    // 32 steps per workgroup at scale; a small observation is cut finer so that it still spreads over the chip (a wave's
    // steps run one after the other: a step is ~1 us of latency)
    // (`batch_width` environments share the launch under env_cycle_many)
    int per = (int)std::min<long long>(32, std::max<long long>(4, steps * batch_width / 2048));
    if (batch_width > 1) { static const int forced = tune("pipe_span", 0); if (forced > 0) per = forced; }     // (tuning: steps per workgroup of a batched render)
    P.steps_per_span = per;
    P.spans = (int)((steps + per - 1) / per);
    P.xcd_chunk = P.spans >= 64 ? P.spans / 8 : 0;
    P.strip_floats = 64 * R.C;
    P.unroll = 1;
    P.div_vhw = make_fastdiv(R.VH * R.VW); P.div_vw = make_fastdiv(R.VW); P.div_f = make_fastdiv(R.F);
    P.div_scale_w = make_fastdiv(R.scale_w); P.div_scale_h = make_fastdiv(R.scale_h);
}

// the minimap of a vh x vw window into d_minif (grown if needed)
// The minimap the next observations will ask for, to be made by clear_dead's own launches (large worlds): the window they used last.
// Not when the observing type skips absorbed agents (the histogram would need the `absorbed` flags: the ordinary path), nor before
// the first observation (no window known).  vh == 0: not folded.
MiniArgs Env::next_minimap() {
    MiniArgs M{};
    static const bool off = tune("fold_minimap", 1) == 0;
    if (off || !minimap_mode || mini_vh <= 0 || mini_skip) return M;
    const size_t need = MAXG + groups.size() * (size_t)mini_vh * mini_vw * (1 + MINI_COPIES);
    if (need > mini_cap) return M;       // (the histogram buffer of the first observation is not there yet)
    return mini_args(mini_vh, mini_vw, false);
}

int *Env::fold_counts() { return d_mini ? d_mini + MAXG + groups.size() * (size_t)mini_vh * mini_vw : nullptr; }

MiniArgs Env::mini_args(int vh, int vw, bool skip) {
    MiniArgs M{};
    M.vh = vh; M.vw = vw; M.skip = skip ? 1 : 0;
    M.scale_h = (height + vh - 1) / vh; M.scale_w = (width + vw - 1) / vw;   // GridWorld.cc:328-329
    grow(arena, d_minif, minif_cap, groups.size() * (size_t)vh * vw, stream);
    M.out = d_minif;
    return M;
}

long long Env::mini_population(bool skip) const {
    long long pop = 0;
    for (auto &gr : groups) pop = pop * 1000003ll + gr.n;
    return pop * 2 + (skip ? 1 : 0);   // the observing type decides whether absorbed agents count
}

// everything a render launch of group g needs: the painted map and the minimap brought up to date (launches only when they
// are stale), the launch plan.  Returns whether the view pointer allows 16-byte stores.
bool Env::prepare_render(int g, const WorldView &W, RenderArgs &R, RenderPlan &P, float *view, float *feat) {
    HostGroup &G = groups[g];
    if (!paint_valid) {
        ensure_tables();
        ProfScope p(*this, "paint");
        launch_paint(stream, W, d_gtab, d_ttab);
        paint_valid = true;
    }
    plan_render(g, R, P, view, feat);
    if (minimap_mode) {
        size_t need = (size_t)W.G * R.VH * R.VW;
        const size_t need_counts = MAXG + need * (1 + MINI_COPIES);   // left-out counters (k_minimap, skip mode) | histogram | clear_dead's copies
        if (need_counts > mini_cap) {   // the histogram buffer is kept zero between uses (k_minimap's last block zeroes what it reads)
            grow(arena, d_mini, mini_cap, need_counts, stream);
            HIP_OK(hipMemsetAsync(d_mini, 0, sizeof(int) * mini_cap, stream));
        }
        grow(arena, d_minif, minif_cap, need, stream);
        R.mini = d_minif;
        const long long pop = mini_population(G.type->can_absorb);
        if (!(mini_valid && mini_vh == R.VH && mini_vw == R.VW && mini_pop == pop)) {
            ProfScope p(*this, "minimap");
            launch_minimap(stream, W, R, d_mini, d_minif);
            mini_valid = true; mini_vh = R.VH; mini_vw = R.VW; mini_pop = pop; mini_skip = G.type->can_absorb;
        }
    }
    const bool aligned = (((uintptr_t)view) & 15) == 0, feat_aligned = (((uintptr_t)feat) & 15) == 0;
    // the feature rows ride in the render launch (its trailing workgroups) when both pointers have the same alignment
    const unsigned feat_q = (unsigned)R.n * (unsigned)R.F / 4;
    P.feat_blocks = aligned == feat_aligned ? (int)std::min<unsigned>((feat_q + 255) / 256 + 1, 16384) : 0;
    return aligned;
}

// GridWorld::get_observation (GridWorld.cc:292-401) into DEVICE buffers, asynchronous on the env stream
void Env::observe_device(int g, float *view, float *feat, bool cells16) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_observation : %d", g);
    if (cells16 && (n_channel() > 7 || (((uintptr_t)view) & 15)))
        fatal("get_observation (bf16 cells): needs at most 7 channels (this game has %d) and a 16-byte aligned buffer", n_channel());
    use_device();
    if (groups[g].n == 0) return;   // the reference dereferences agents[0] here (UB); nothing to write for n = 0
    if (groups[g].acted && !serial_calls_on) {   // set_action came first: the feature rows show the new last_action (GridWorld.cc:386-396)
        join_side();
        GroupDev G = groups[g].cur; G.n = groups[g].n;
        launch_commit_action(stream, G, groups[g].tdev);
    }
    mark_state();                   // (the side stream waits for the world as it is before this render, not for the render)
    WorldView W = this->view();
    RenderArgs R; RenderPlan P;
    const bool aligned = prepare_render(g, W, R, P, view, feat);
    const bool feat_aligned = (((uintptr_t)feat) & 15) == 0;
    R.cells16 = cells16 ? 1 : 0;
    {
        // A painted map that does not fit the L2s, looked at by agents whose order in the group says nothing about where they stand (random
        // placement): every window row is an L2 miss, served by the Infinity Cache -- if the map is still there.  Behind a step it is not
        // (the step's kernels have been through half a gigabyte of other arrays); streaming the map through once, ahead of the first render
        // of a cycle, puts it back: 80 MB in 13 us, and the two renders of the reference's 1M harness run 0.242 -> 0.215 ms each
        // (profiles/r05_summary.md; MAGENT_TUNE touch_map=0 / 1: never / before every such render).  Spatially ordered populations
        // (train_battle.py's formation) read the map once either way: nothing to warm.
        static const int touch = tune("touch_map", -1);
        const bool big_map = (size_t)width * height * (W.vc_packed ? 4 : 8) > (16u << 20);
        if (big_map && (touch > 0 || (touch < 0 && map_scattered && !map_warm))) launch_touch_map(stream, W);
        map_warm = true;                      // (a render walks the map itself)
        ProfScope p(*this, "render", true);
        last_render_kernel = launch_render(stream, W, R, P, aligned, aligned);
    }
    if (P.feat_blocks == 0) {
        ProfScope p(*this, "features", true);
        launch_features(stream, W, R, P, feat_aligned);
    }
    HIP_OK(hipGetLastError());
}

// host-buffer variant (the reference ABI): render into a staging buffer, then copy out
void Env::observe_host(int g, float *view, float *feat) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_observation : %d", g);
    use_device();
    HostGroup &G = groups[g];
    if (G.n == 0) return;
    const HostType &t = *G.type;
    size_t nv = (size_t)G.n * t.view.height * t.view.width * n_channel(), nf = (size_t)G.n * feature_size(g);
    grow(arena, d_stage_view, stage_view_cap, nv, stream);
    grow(arena, d_stage_feat, stage_feat_cap, nf, stream);
    observe_device(g, d_stage_view, d_stage_feat);
    copy_out(view, d_stage_view, sizeof(float) * nv);
    copy_out(feat, d_stage_feat, sizeof(float) * nf);
}

// ------------------------------------------------------------------------------------------------ info
void Env::info_device(int g, const char *name, void *out) {
    if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_info : %d", g);
    use_device();
    GroupDev G = groups[g].cur; G.n = groups[g].n;
    if (G.n == 0) return;
    std::string k(name);
    if (k == "id") HIP_OK(hipMemcpyAsync(out, G.id, sizeof(int) * G.n, hipMemcpyDeviceToDevice, stream));
    else if (k == "hp") HIP_OK(hipMemcpyAsync(out, G.hp, sizeof(float) * G.n, hipMemcpyDeviceToDevice, stream));
    else if (k == "pos") launch_get_pos(stream, G, (int *)out);
    else if (k == "alive") launch_get_alive(stream, G, (unsigned char *)out);
    else fatal("unsupported info name in get_info_device : %s", name);
}

// GridWorld::get_info (GridWorld.cc:709-894)
void Env::info_host(int g, const char *name, void *buf) {
    std::string k(name);
    int *ib = (int *)buf; float *fb = (float *)buf;
    auto need_group = [&]() { if (g < 0 || g >= (int)groups.size()) fatal("invalid group handle in get_info(%s) : %d", name, g); };
    if (k == "num") { need_group(); ib[0] = groups[g].n; return; }
    if (k == "engine_stats") {   // additive: steps whose optimistic rounds ran out (host continued), rounds of the last checked phases
        ib[0] = fallback_steps; ib[1] = last_attack_iters; ib[2] = last_move_iters; ib[3] = attack_round;
        ib[4] = fallback_attack; ib[5] = fallback_move;      // (which phase's optimistic rounds ran out)
        ib[6] = last_render_kernel;                          // 0 k_render, 1 k_render_fast, 4 k_render_sweep2
        ib[7] = plain_steps;                                 // steps that took the fused passes of plain games (k_strike ...)
        return;
    }
    if (k == "pipeline_stats") { // additive (tests): what only changes with the LENGTH of an episode of the plain pipeline
        ib[0] = plain_steps;                                 // steps through k_plain_rank ... k_plain_commit
        ib[1] = pairs_two_steps; ib[2] = pairs_one_steps;    // ... launched with two / with one optimistic pair of death-rank rounds
        ib[3] = claim_refills;                               // times the claim words were refilled for such a step (a new window of 63 epochs, or another path wrote them)
        ib[4] = (int)(plain_epoch % 63u);                    // where the current window stands
        ib[5] = fallback_attack;                             // steps whose optimistic rounds ran out
        ib[6] = pipe_rounds;                                 // cycles that went through the batched pipeline (env_cycle_many, pipe.hip)
        ib[7] = pipe_sweep_rounds;                           // ... of which the batch's render launch was the sweeping kernel
        return;
    }
    if (k == "round_hist") {     // additive (tuning): plain steps since the last read by the last round of the death-rank fixed point that
        // still changed something (0: none did; one more round than that was needed to see it converge), steps that ran out not counted
        for (int q = 0; q < 9; q++) { ib[q] = round_hist[q]; round_hist[q] = 0; }
        return;
    }
    if (k == "batch_host_us") {  // additive (tuning): host microseconds per env_cycle_many round since the last read:
        // prepare | copy + launches | wait for the first record | the other records
        for (int q = 0; q < 4; q++) { fb[q] = batch_rounds ? (float)(batch_us[q] / batch_rounds) : 0.f; batch_us[q] = 0; }
        batch_rounds = 0;
        return;
    }
    if (k == "step_marks") {     // additive (tuning): ns since the first mark at every phase boundary of the last one-launch step
        const int n = h_rec ? h_rec->n_marks : 0;
        ib[0] = n;
        for (int q = 0; q < n; q++) ib[1 + q] = (int)((h_rec->marks[q] - h_rec->marks[0]) * 10ull);
        return;
    }
    if (k == "action_space") { need_group(); ib[0] = groups[g].type->n_action; return; }
    if (k == "view_space") { need_group(); ib[0] = groups[g].type->view.height; ib[1] = groups[g].type->view.width; ib[2] = n_channel(); return; }
    if (k == "feature_space") { need_group(); ib[0] = feature_size(g); return; }
    if (k == "attack_base") { need_group(); ib[0] = groups[g].type->attack_base; return; }
    if (k == "view2attack") {  // GridWorld.cc:853-870
        need_group();
        const HostType &t = *groups[g].type;
        std::fill(ib, ib + t.view.height * t.view.width, -1);
        for (int i = 0; i < t.attack.count; i++) {   // (an offset outside the view window has no cell in the table: the reference writes out of bounds there)
            const int vy = t.attack.dy[i] - t.view.y1, vx = t.attack.dx[i] - t.view.x1;
            if (vy >= 0 && vy < t.view.height && vx >= 0 && vx < t.view.width) ib[vy * t.view.width + vx] = i;
        }
        return;
    }
    if (k == "groups_info") {
        const int colors[][3] = {{192, 64, 64}, {64, 64, 192}, {64, 192, 64}, {64, 64, 64}};
        for (size_t i = 0; i < groups.size(); i++) {
            ib[5 * i] = groups[i].type->width; ib[5 * i + 1] = groups[i].type->length;
            for (int c = 0; c < 3; c++) ib[5 * i + 2 + c] = colors[i % 4][c];
        }
        return;
    }
    if (k == "both_attack") { ib[0] = 0; return; }
    if (!device_ready) fatal("get_info(%s) called before reset", name);
    use_device();
    if (k == "mean_info") {      // GridWorld.cc:765-786 ("deprecated" there; a cold path here: the arrays are fetched to the host)
        // [mean x, mean y, share of every action]: float sums in agent order (the reference's loop under one OpenMP thread), dead agents that
        // have not been cleared included, the last action of every agent counted.  An agent that has never been given an action holds
        // n_action (GridWorld.h:140): the reference counts it one past the end of its `new int[n_action]`; here it is counted nowhere.
        need_group();
        HostGroup &G = groups[g];
        const int n = G.n, na = G.type->n_action;
        if (n == 0) fatal("get_info(mean_info) of an empty group (the reference asserts agent_size != 0 here, GridWorld.cc:782)");
        if (G.acted && !serial_calls_on && n > 0) {    // set_action came first: Agent::get_action shows the new action (as in observe_device)
            join_side();
            GroupDev D = G.cur; D.n = n;
            launch_commit_action(stream, D, G.tdev);
        }
        HIP_OK(hipStreamSynchronize(stream));
        std::vector<int> xs(n), ys(n), la(n);
        if (n) {
            HIP_OK(hipMemcpy(xs.data(), G.cur.x, sizeof(int) * n, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(ys.data(), G.cur.y, sizeof(int) * n, hipMemcpyDeviceToHost));
            HIP_OK(hipMemcpy(la.data(), G.cur.last_action, sizeof(int) * n, hipMemcpyDeviceToHost));
        }
        float sum_x = 0, sum_y = 0;
        std::vector<int> counter(na, 0);
        for (int i = 0; i < n; i++) {
            sum_x += xs[i]; sum_y += ys[i];
            if (la[i] >= 0 && la[i] < na) counter[la[i]]++;
        }
        const size_t agent_size = (size_t)n;
        fb[0] = sum_x / agent_size; fb[1] = sum_y / agent_size;
        for (int i = 0; i < na; i++) fb[2 + i] = (float)(1.0 * counter[i] / agent_size);
        return;
    }
    if (k == "id" || k == "pos" || k == "alive") {
        need_group();
        int n = groups[g].n;
        if (n == 0) return;
        size_t bytes = k == "pos" ? sizeof(int) * 2 * n : k == "alive" ? (size_t)n : sizeof(int) * n;
        grow(arena, d_stage_small, stage_small_cap, (size_t)n * 8, stream);
        info_device(g, name, d_stage_small);
        read_back(buf, d_stage_small, bytes);
        return;
    }
    if (k == "walls_info") {
        download_occ();
        int ct = 0;
        for (size_t c = 0; c < h_occ.size(); c++) if (h_occ[c] == OCC_WALL) { ct++; ib[2 * ct] = (int)(c % width); ib[2 * ct + 1] = (int)(c / width); }
        ib[0] = ct;
        return;
    }
    if (k == "global_minimap") {  // GridWorld.cc:738-764 (cold path: positions are fetched to the host)
        int vh = (int)std::lround(fb[0]), vw = (int)std::lround(fb[1]), NG = (int)groups.size();
        std::memset(fb, 0, sizeof(float) * vh * vw * NG);
        int sh = (height + vh - 1) / vh, sw = (width + vw - 1) / vw;
        HIP_OK(hipStreamSynchronize(stream));
        for (int i = 0; i < NG; i++) {
            int ch = (i - g + NG) % NG, n = groups[i].n;
            std::vector<int> xs(n), ys(n);
            if (n) {
                HIP_OK(hipMemcpy(xs.data(), groups[i].cur.x, sizeof(int) * n, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(ys.data(), groups[i].cur.y, sizeof(int) * n, hipMemcpyDeviceToHost));
            }
            for (int j = 0; j < n; j++) fb[((ys[j] / sh) * vw + xs[j] / sw) * NG + ch]++;
            for (int c = 0; c < vh * vw; c++) fb[c * NG + ch] /= (size_t)n;
        }
        return;
    }
    if (k == "render_window_info") {  // GridWorld.cc:797-834
        first_render = false;
        int x1 = ib[0], y1 = ib[1], x2 = ib[2], y2 = ib[3], ct = 1;
        HIP_OK(hipStreamSynchronize(stream));
        for (size_t i = 0; i < groups.size(); i++) {
            int n = groups[i].n;
            std::vector<int> xs(n), ys(n), ids(n);
            std::vector<unsigned char> taken(n, 1);
            if (n) {
                HIP_OK(hipMemcpy(xs.data(), groups[i].cur.x, sizeof(int) * n, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(ys.data(), groups[i].cur.y, sizeof(int) * n, hipMemcpyDeviceToHost));
                HIP_OK(hipMemcpy(ids.data(), groups[i].cur.id, sizeof(int) * n, hipMemcpyDeviceToHost));
                if (groups[i].type->can_absorb) HIP_OK(hipMemcpy(taken.data(), groups[i].cur.absorbed, n, hipMemcpyDeviceToHost));
            }
            for (int j = 0; j < n; j++) {
                if (xs[j] < x1 || xs[j] > x2 || ys[j] < y1 || ys[j] > y2) continue;
                if (!taken[j]) continue;   // a goal shows once it has taken a mover in (GridWorld.cc:821-822)
                ib[4 * ct] = ids[j]; ib[4 * ct + 1] = xs[j]; ib[4 * ct + 2] = ys[j]; ib[4 * ct + 3] = (int)i;
                ct++;
            }
        }
        ib[0] = ct - 1; ib[1] = (int)attack_events.size();
        return;
    }
    if (k == "attack_event") {
        for (size_t i = 0; i < attack_events.size(); i++) { ib[3 * i] = attack_events[i].id; ib[3 * i + 1] = attack_events[i].x; ib[3 * i + 2] = attack_events[i].y; }
        return;
    }
    fatal("unsupported info name in GridWorld::get_info : %s", name);
}

// ------------------------------------------------------------------------------------------------ render (text dump)
// RenderGenerator::gen_config (RenderGenerator.cc:57-105)
void Env::gen_render_config() {
    std::ofstream f(render_dir + "/config.json");
    const int colors[][3] = {{192, 64, 64}, {64, 64, 192}, {64, 192, 64}, {64, 64, 64}};
    auto rgba = [](int r, int g, int b, float a) { std::stringstream ss; ss << "\"rgba(" << r << "," << g << "," << b << "," << a << ")\""; return ss.str(); };
    auto kv = [&](const char *key, auto value, bool last = false) { f << "\"" << key << "\": " << value; f << (last ? "" : ",") << std::endl; };
    f << "{" << std::endl;
    kv("width", width); kv("height", height); kv("static-file", "\"static.map\"");
    kv("obstacle-style", rgba(127, 127, 127, 1)); kv("dynamic-file-directory", "\".\"");
    kv("attack-style", rgba(63, 63, 63, 0.8f)); kv("minimap-width", 300); kv("minimap-height", 250);
    f << "\"group\" : [" << std::endl;
    for (size_t i = 0; i < groups.size(); i++) {
        const HostType &t = *groups[i].type;
        const int *c = colors[i % 4];
        f << "{" << std::endl;
        kv("height", t.length); kv("width", t.width); kv("style", rgba(c[0], c[1], c[2], 1)); kv("anchor", "[0, 0]");
        kv("max-speed", (int)t.speed); kv("speed-style", rgba(c[0], c[1], c[2], 0.01f));
        kv("vision-radius", t.view_radius); kv("vision-angle", t.view_angle); kv("vision-style", rgba(c[0], c[1], c[2], 0.2f));
        kv("attack-radius", t.attack_radius); kv("attack-angle", t.attack_angle); kv("attack-style", rgba(c[0], c[1], c[2], 0.1f));
        kv("broadcast-radius", 1, true);
        f << (i + 1 == groups.size() ? "}" : "},") << std::endl;
    }
    f << "]" << std::endl << "}" << std::endl;
}

// GridWorld::render (GridWorld.cc:939-949) + RenderGenerator::render_a_frame (RenderGenerator.cc:108-185)
void Env::render() {
    if (!device_ready) fatal("render called before reset");
    enter();
    if (first_render) {
        first_render = false;
        if (!render_dir.empty()) gen_render_config();
    }
    if (render_dir.empty()) return;
    HIP_OK(hipStreamSynchronize(stream));
    std::ofstream fout(render_dir + "/video_" + std::to_string(file_ct) + ".txt", frame_ct == 0 ? std::ios::out : std::ios::app);
    if (frame_ct == 0) {
        download_occ();
        size_t n_wall = 0;
        for (int c : h_occ) n_wall += c == OCC_WALL;
        fout << "W " << n_wall << std::endl;
        for (size_t c = 0; c < h_occ.size(); c++) if (h_occ[c] == OCC_WALL) fout << (c % width) << " " << (c / width) << std::endl;
    }
    size_t n_agents = 0;
    std::vector<std::vector<unsigned char>> taken(groups.size());
    for (size_t i = 0; i < groups.size(); i++) {   // goals are drawn once they have taken a mover in (RenderGenerator.cc:128-141)
        taken[i].assign(groups[i].n, 1);
        if (groups[i].type->can_absorb && groups[i].n)
            HIP_OK(hipMemcpy(taken[i].data(), groups[i].cur.absorbed, groups[i].n, hipMemcpyDeviceToHost));
        for (unsigned char t : taken[i]) n_agents += t;
    }
    fout << "F " << n_agents << " " << attack_events.size() << " " << 0 << std::endl;
    for (size_t i = 0; i < groups.size(); i++) {
        const int n = groups[i].n;
        if (n == 0) continue;
        std::vector<int> xs(n), ys(n), ids(n);
        std::vector<float> hp(n);
        HIP_OK(hipMemcpy(xs.data(), groups[i].cur.x, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ys.data(), groups[i].cur.y, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(ids.data(), groups[i].cur.id, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(hp.data(), groups[i].cur.hp, sizeof(float) * n, hipMemcpyDeviceToHost));
        std::vector<int> dirs(n, DIR_NORTH);
        if (turn_mode) HIP_OK(hipMemcpy(dirs.data(), groups[i].cur.dir, sizeof(int) * n, hipMemcpyDeviceToHost));
        const float type_hp = groups[i].type->hp;
        for (int j = 0; j < n; j++) {
            if (!taken[i][j]) continue;
            int pct = std::min(std::max(0, int(100 * hp[j] / type_hp)), 100);
            fout << ids[j] << " " << pct << " " << 90 * dirs[j] << " " << xs[j] << " " << ys[j] << " " << i << std::endl;  // dir2angle (RenderGenerator.cc:148)
        }
    }
    for (const AttackEvent &e : attack_events) fout << 0 << " " << e.id << " " << e.x << " " << e.y << std::endl;
    if (frame_ct++ > frame_per_file) { frame_ct = 0; file_ct++; }
}

void Env::sync() {
    if (!device_ready) return;
    enter();
    HIP_OK(hipStreamSynchronize(stream));
}

}  // namespace magent_amd
