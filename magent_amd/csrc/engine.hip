// engine.hip -- host side of the MI355X grid-world engine: configuration, agent types, placement, per-call
// orchestration of the kernels in render.hip / step.hip / cycle.hip on one HIP stream per environment.
//
// Mirrors the behaviour of the reference's GridWorld class (src/gridworld/GridWorld.{h,cc}) for the hot-path scope
// of SURVEY.md section 8.  The product never falls back to a CPU engine: every state-changing operation after
// placement runs on the GPU.  The only host-side algorithmic pieces are the cold-path placement (add_agents, which
// the reference defines as sequential rejection sampling on the engine RNG) and, in this round, the attack
// shuffle's permutation (a function of the RNG state and the attack count only).
#include "engine_impl.h"

namespace magent_amd {

[[noreturn]] void fatal(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    std::fprintf(stderr, "magent-amd FATAL: ");
    std::vfprintf(stderr, fmt, ap);
    std::fprintf(stderr, "\n");
    va_end(ap);
    std::abort();
}

// ------------------------------------------------------------------------------------------------ device memory
// Small device arrays are carved from a few large blocks per environment instead of one hipMalloc each: an environment has
// ~100 of them (30 per group, the map arrays, tables, scratch) -- one allocation call instead of a hundred when a world is
// created, and the state of a small world contiguous in memory.  (Requests above 1 MiB keep their own hipMalloc.)
void *DevArena::take(size_t bytes) {
    bytes = (bytes + 255) & ~(size_t)255;
    if (used + bytes > BLOCK) {
        char *blk = nullptr;
        HIP_OK(hipMalloc(&blk, BLOCK));
        blocks.push_back(blk);
        used = 0;
    }
    void *p = blocks.back() + used;
    used += bytes;
    return p;
}
bool DevArena::owns(const void *p) const {
    for (char *b : blocks) if ((const char *)p >= b && (const char *)p < b + BLOCK) return true;
    return false;
}
void DevArena::release() {
    for (char *b : blocks) (void)hipFree(b);
    blocks.clear();
    used = BLOCK;
}

// ------------------------------------------------------------------------------------------------ ranges / types
// CircleRange of the reference (Range.h:149-190): double arithmetic and eps constants reproduced exactly
void HostRange::circle(float radius, float inner_radius, int parity) {
    const double eps = 1e-8;
    width = 2 * int(radius + eps) + parity;
    const int center = (int)radius;
    if (width % 2 != parity) width++;
    height = width;
    in.assign((size_t)width * width, 0);
    dx.clear(); dy.clear(); count = 0;
    const double delta = (parity == 0 ? 0.5 : 0);
    for (int i = 0; i < width; i++)
        for (int j = 0; j < width; j++) {
            double ax = std::fabs(j - center + delta), ay = std::fabs(i - center + delta);
            double dis = std::sqrt(ax * ax + ay * ay);
            if (dis < radius + eps && dis > inner_radius - eps) {
                in[(size_t)i * width + j] = 1;
                dx.push_back(j - center); dy.push_back(i - center); count++;
            }
        }
    x1 = y1 = -center;
    x2 = y2 = width - center - 1;
}

// SectorRange of the reference (Range.h:104-144): the rectangle in front of the agent -- rows -height .. -1 of its frame -- under
// a sector mask; float / double mix and constants as there (the window shape and the in-range test are part of the observable result)
void HostRange::sector(float angle, float radius, int parity) {
    static const double PI = 3.1415926536;
    height = (int)(radius + 0.5);
    width = (int)(2 * radius * std::sin(angle / 2 * (PI / 180)) + 0.5);
    if (width % 2 != parity) width--;
    // (radius 0 -- a type without attack range: height 0, width -1 after the parity step; no cell, no action)
    if (height > 0 && width <= 0) fatal("sector range (angle %g, radius %g) too narrow: the reference allocates a non-positive array here", angle, radius);
    in.assign(height > 0 ? (size_t)width * height : 0, 0);
    dx.clear(); dy.clear(); count = 0;
    const double eps = 0.00001;
    for (int i = 0; i < height; i++)
        for (int j = 0; j < width; j++) {
            const double dis_x = std::fabs(j - (width - 1) / 2.0), dis_y = std::fabs(height - i);
            const double dis = std::sqrt(dis_x * dis_x + dis_y * dis_y);
            if (dis < radius + 0.2 + eps && dis_x / dis_y < std::tan(angle / 2 * PI / 180) + eps) {
                in[(size_t)i * width + j] = 1;
                dx.push_back(j - width / 2); dy.push_back(i - height); count++;
            }
        }
    x1 = -width / 2; y1 = -height;
    x2 = (width - 1) / 2; y2 = -1;
}

// ------------------------------------------------------------------------------------------------ profiling
hipEvent_t Env::prof_event() {
    if (!prof_pool.empty()) { hipEvent_t ev = prof_pool.back(); prof_pool.pop_back(); return ev; }
    hipEvent_t ev;
    HIP_OK(hipEventCreate(&ev));
    return ev;
}

void Env::profile_read(const char *name, int *n, float *ms) {
    enter();
    HIP_OK(hipStreamSynchronize(stream));
    ProfSlot &s = prof[name];
    *n = (int)s.pending.size();
    *ms = 0;
    for (auto &p : s.pending) {
        float t = 0;
        HIP_OK(hipEventElapsedTime(&t, p.first, p.second));
        *ms += t;
        prof_pool.push_back(p.first); prof_pool.push_back(p.second);
    }
    s.pending.clear();
}

// ------------------------------------------------------------------------------------------------ host copy-out
CopyPool::CopyPool(int n_threads) {
    for (int i = 0; i < n_threads; i++) threads.emplace_back([this, i] { worker(i); });
}
CopyPool::~CopyPool() {
    { std::lock_guard<std::mutex> l(mu); stop = true; generation++; }
    cv_go.notify_all();
    for (auto &t : threads) t.join();
}
void CopyPool::worker(int id) {
    unsigned long long seen = 0;
    while (true) {
        char *d; const char *s; size_t b; size_t parts;
        {
            std::unique_lock<std::mutex> l(mu);
            cv_go.wait(l, [&] { return generation != seen; });
            seen = generation;
            if (stop) return;
            d = dst; s = src; b = bytes; parts = threads.size();
        }
        size_t per = ((b + parts - 1) / parts + 4095) & ~(size_t)4095;
        size_t lo = per * (size_t)id, hi = std::min(b, lo + per);
        if (lo < hi) std::memcpy(d + lo, s + lo, hi - lo);
        {
            std::lock_guard<std::mutex> l(mu);
            if (--pending == 0) cv_done.notify_one();
        }
    }
}
void CopyPool::copy(void *d, const void *s, size_t b) {
    std::unique_lock<std::mutex> l(mu);
    dst = (char *)d; src = (const char *)s; bytes = b;
    pending = (int)threads.size();
    generation++;
    cv_go.notify_all();
    cv_done.wait(l, [&] { return pending == 0; });
}

// Device buffer -> the caller's pageable host buffer (the reference ABI hands numpy arrays).  A plain hipMemcpy to
// pageable memory measured 11 GB/s; here 32 MiB chunks are DMA'd into a ring of pinned buffers on a second stream
// while worker threads drain the previous chunk into the destination.
// Small read-backs go through a pinned bounce buffer of the engine's own and are handed to the caller's (pageable)
// memory by the calling thread, after the stream has been waited for: nothing outside this function ever writes into
// the caller's buffer, and nothing writes into it after the call has returned.
void Env::read_back(void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes == 0) return;
    if (bytes > h_small_cap) {
        if (h_small) HIP_OK(hipHostFree(h_small));
        h_small_cap = std::max<size_t>(bytes, std::max<size_t>(h_small_cap * 2, 1u << 16));
        HIP_OK(hipHostMalloc((void **)&h_small, h_small_cap, hipHostMallocDefault));
    }
    HIP_OK(hipMemcpyAsync(h_small, dev_src, bytes, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::memcpy(host_dst, h_small, bytes);
}

void Env::copy_out(void *host_dst, const void *dev_src, size_t bytes) {
    if (bytes < (8u << 20)) { read_back(host_dst, dev_src, bytes); return; }
    if (!pool) {
        unsigned hw = std::thread::hardware_concurrency();
        int nt = (int)std::max(2u, std::min(16u, hw / 4));
        if (const char *v = std::getenv("MAGENT_COPY_THREADS")) nt = std::max(1, std::atoi(v));
        pool = new CopyPool(nt);
        HIP_OK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
        for (int i = 0; i < COPY_RING; i++) {
            HIP_OK(hipHostMalloc((void **)&h_ring[i], COPY_CHUNK, hipHostMallocDefault));
            HIP_OK(hipEventCreateWithFlags(&ring_ev[i], hipEventDisableTiming));
        }
    }
    HIP_OK(hipStreamSynchronize(stream));   // the producer kernels have finished
    const size_t n_chunks = (bytes + COPY_CHUNK - 1) / COPY_CHUNK;
    auto len = [&](size_t k) { return std::min(COPY_CHUNK, bytes - k * COPY_CHUNK); };
    // software pipeline: up to COPY_RING - 1 DMA chunks in flight ahead of the CPU drain
    size_t issued = 0;
    for (size_t k = 0; k < n_chunks; k++) {
        while (issued < n_chunks && issued < k + COPY_RING) {   // chunk k + COPY_RING reuses chunk k's buffer
            int b = (int)(issued % COPY_RING);
            HIP_OK(hipMemcpyAsync(h_ring[b], (const char *)dev_src + issued * COPY_CHUNK, len(issued), hipMemcpyDeviceToHost, copy_stream));
            HIP_OK(hipEventRecord(ring_ev[b], copy_stream));
            issued++;
        }
        int b = (int)(k % COPY_RING);
        HIP_OK(hipEventSynchronize(ring_ev[b]));
        pool->copy((char *)host_dst + k * COPY_CHUNK, h_ring[b], len(k));
    }
}

// ------------------------------------------------------------------------------------------------ lifecycle
Env::Env() {
    const char *d = std::getenv("MAGENT_DEVICE");
    if (!d) d = std::getenv("LOCAL_RANK");
    device_id = d ? std::atoi(d) : 0;
    rng.seed(0);  // GridWorld.cc:29
    // what tests and tuning runs force through MAGENT_TUNE (tune.h); the defaults are the measured best
    host_shuffle = tune("host_shuffle", 0) != 0;
    checked_step = tune("checked_step", 0) != 0;
    if (tune_set("attack_pairs")) { opt_attack_pairs = std::max(0, tune("attack_pairs", 1)); opt_fixed = true; }
    if (tune_set("move_batches")) { opt_move_batches = std::max(0, tune("move_batches", 1)); opt_fixed = true; }
    solo_enabled = tune("solo_step", 1) != 0;
    if (tune_set("overlap")) { overlap_level = tune("overlap", 0); overlap_enabled = overlap_level != 0; }
    solo_max_agents = std::max(0, tune("solo_max", solo_max_agents));
    batch_solo_max = std::max(0, tune("batch_solo_max", batch_solo_max));
}


Env::~Env() {
    if (!device_ready) return;
    use_device();
    if (side) { (void)hipStreamSynchronize(side); (void)hipStreamDestroy(side); (void)hipEventDestroy(ev_state); (void)hipEventDestroy(ev_side); }
    (void)hipStreamSynchronize(stream);
    for (auto &g : groups) free_group(g);
    dfree(arena, d_occ); dfree(arena, d_viewcell); dfree(arena, d_claim); dfree(arena, d_food); dfree(arena, d_powtab); dfree(arena, d_counters); dfree(arena, d_gtab); dfree(arena, d_ttab);
    dfree(arena, d_delta); dfree(arena, d_mask); dfree(arena, d_mini); dfree(arena, d_minif); dfree(arena, d_sums); dfree(arena, d_rank); dfree(arena, d_shuf); dfree(arena, d_events); dfree(arena, serial_alist); dfree(arena, serial_mlist); dfree(arena, serial_dcalls); dfree(arena, d_actions);
    dfree(arena, d_stage_view); dfree(arena, d_stage_feat); dfree(arena, d_stage_small);
    dfree(arena, d_mvnodes); dfree(arena, d_hit); dfree(arena, d_rule_args); dfree(arena, d_rule_progs); dfree(arena, batch_d); dfree(arena, d_asums); dfree(arena, d_wpre); dfree(arena, d_ptab); dfree(arena, d_alive);
    if (batch_h) (void)hipHostFree(batch_h);
    if (h_rec) (void)hipHostFree(h_rec);
    dfree(arena, pipe_d); dfree(arena, reports_d); dfree(arena, pipe_ticket); dfree(arena, d_newn);
    if (pipe_h) (void)hipHostFree(pipe_h);
    if (reports_h) (void)hipHostFree(reports_h);
    if (pipe_flag) (void)hipHostFree(pipe_flag);
    if (pool) {
        delete pool;
        for (int i = 0; i < COPY_RING; i++) { (void)hipHostFree(h_ring[i]); (void)hipEventDestroy(ring_ev[i]); }
        (void)hipStreamDestroy(copy_stream);
    }
    if (h_counters) (void)hipHostFree(h_counters);
    if (h_small) (void)hipHostFree(h_small);
    if (h_rank) (void)hipHostFree(h_rank);
    for (auto &kv : prof) for (auto &p : kv.second.pending) { (void)hipEventDestroy(p.first); (void)hipEventDestroy(p.second); }
    for (auto ev : prof_pool) (void)hipEventDestroy(ev);
    stream_owner.reset();   // (the stream goes when its last user does)
    arena.release();
}

void Env::use_device() { HIP_OK(hipSetDevice(device_id)); }

// ------------------------------------------------------------------------------------------------ the side stream
// At 800k agents a step + observation is ~1 ms of which the two observation renders are 0.64 ms of pure HBM writing, while
// set_action, the attack shuffle, the hit gather and the death-rank fixed point (~0.18 ms of latency-bound launches) only READ
// the world the renders read, and write scratch the renders never look at (pend, key, the hit words, ranks; last_action is
// stored later, see k_set_action_a).  They run on a second stream, under the renders:
//   stream : ... clear_dead | render g0 | render g1 ............| (waits for side) attack_apply, move, rules, finish
//   side   :   (waits for the state)   set_action g0, g1, shuffle, rank, eval rounds |
// MEASURED (MI355X, bench workload, profiles/r02_overlap.txt): 0.952 -> 0.896 ms per step (+6 %), but the renders stretch from
// 0.303 to 0.347 ms each -- the side work is random 4-byte traffic that costs whole HBM transactions, so it takes back more
// than half of what it hides.  OFF by default (MAGENT_TUNE overlap=3 turns all of it on, 2 the shuffle only, 1 set_action only):
// the render stays at its roofline fraction and the step's head stays the thing to make cheaper.  The GPU suite runs the
// dense scenarios both ways (tests/test_gpu_fullsize.py: multi_launch_step / multi_launch_one_stream).
// Rules that keep this exact whatever the caller does:
//   * every call that changes the world, or must see all of it, starts with enter(): `stream` waits for what `side` still
//     has in flight, and the state epoch moves on;
//   * `side` waits for an event recorded on `stream` behind the last state-changing call (mark_state: recorded lazily, by the
//     first observation or side-stream use after such a call, i.e. BEFORE any render is enqueued behind it);
//   * an observation of a group whose actions are already set joins and commits last_action first (the feature rows show it).
bool Env::side_wanted() {
    if (!overlap_enabled || checked_step || host_shuffle || !first_render || (turn_mode && any_multicell)) return false;
    int total_n = 0;
    for (auto &g : groups) total_n += g.n;
    return total_n > 0 && !solo_ok(total_n);
}
void Env::mark_state() {
    if (!side || marked_epoch == state_epoch) return;
    HIP_OK(hipEventRecord(ev_state, stream));
    marked_epoch = state_epoch;
}
hipStream_t Env::side_stream() {
    use_device();
    if (!side) {
        HIP_OK(hipStreamCreateWithFlags(&side, hipStreamNonBlocking));
        HIP_OK(hipEventCreateWithFlags(&ev_state, hipEventDisableTiming));
        HIP_OK(hipEventCreateWithFlags(&ev_side, hipEventDisableTiming));
    }
    mark_state();
    if (side_epoch != marked_epoch) {
        HIP_OK(hipStreamWaitEvent(side, ev_state, 0));
        side_epoch = marked_epoch;
    }
    side_dirty = true;
    return side;
}
void Env::join_side() {
    if (!side_dirty) return;
    HIP_OK(hipEventRecord(ev_side, side));
    HIP_OK(hipStreamWaitEvent(stream, ev_side, 0));
    side_dirty = false;
}
void Env::enter() {
    use_device();
    join_side();
    state_epoch++;
}
hipStream_t Env::action_stream() { return device_ready && side_wanted() ? side_stream() : stream; }

void Env::init_device() {
    if (device_ready) return;
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count == 0)
        fatal("no HIP device available (%s). This engine has no CPU fallback.", hipGetErrorString(e));
    if (device_id >= count) fatal("device_id %d out of range (%d devices)", device_id, count);
    use_device();
    HIP_OK(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
    stream_owner = std::shared_ptr<void>((void *)stream, [](void *p) { (void)hipStreamDestroy((hipStream_t)p); });
    HIP_OK(dev_malloc(arena, &d_counters, sizeof(int) * CTR_TOTAL));
    HIP_OK(hipMemset(d_counters, 0, sizeof(int) * CTR_TOTAL));
    HIP_OK(dev_malloc(arena, &d_gtab, sizeof(GroupDev) * MAXG));
    HIP_OK(dev_malloc(arena, &d_ttab, sizeof(TypeDev) * MAXG));
    HIP_OK(hipHostMalloc((void **)&h_counters, sizeof(int) * CTR_TOTAL, hipHostMallocDefault));
    HIP_OK(hipHostMalloc((void **)&h_rec, sizeof(StepRecord), hipHostMallocDefault));   // (default = coherent, device-visible)
    std::memset(h_rec, 0, sizeof(StepRecord));
    device_ready = true;
}

// ------------------------------------------------------------------------------------------------ configuration
// GridWorld::set_config (GridWorld.cc:120-149)
void Env::set_config(const char *key, void *p) {
    std::string k(key);
    if (k == "map_width") width = *(int *)p;
    else if (k == "map_height") height = *(int *)p;
    else if (k == "minimap_mode") minimap_mode = *(bool *)p;
    else if (k == "embedding_size") embedding_size = *(int *)p;
    else if (k == "seed") { rng.seed((unsigned long)*(int *)p); rng_on_device = false; }
    else if (k == "device_id") { if (device_ready && *(int *)p != device_id) fatal("device_id must be set before env_reset"); device_id = *(int *)p; }
    else if (k == "render_dir") render_dir = (const char *)p;
    else if (k == "food_mode") food_mode = *(bool *)p;
    else if (k == "turn_mode") {
        if (!types.empty() && *(bool *)p != turn_mode) fatal("turn_mode must be configured before the agent types are registered (it changes their action layout)");
        turn_mode = *(bool *)p;
    } else if (k == "goal_mode") goal_mode = *(bool *)p;   // two more feature slots, never written (GridWorld.cc:137-138, :929-930)
    else fatal("invalid argument in GridWorld::set_config : %s", key);
}

// AgentType::AgentType (AgentType.cc:30-123)
void Env::register_agent_type(const char *name, int n, const char **keys, float *values) {
    if (types.count(name)) fatal("duplicated name of agent type in GridWorld::register_agent_type : %s", name);
    HostType t;
    t.name = name;
    for (int i = 0; i < n; i++) {
        std::string k(keys[i]);
        float v = values[i];
        if (k == "width") t.width = (int)(v + 0.5);
        else if (k == "length") t.length = (int)(v + 0.5);
        else if (k == "speed") t.speed = v;
        else if (k == "hp") t.hp = v;
        else if (k == "view_radius") t.view_radius = v;
        else if (k == "view_angle") t.view_angle = v;
        else if (k == "attack_radius") t.attack_radius = v;
        else if (k == "attack_angle") t.attack_angle = v;
        else if (k == "damage") t.damage = v;
        else if (k == "step_recover") t.step_recover = v;
        else if (k == "kill_supply") t.kill_supply = v;
        else if (k == "attack_in_group") t.attack_in_group = bool(int(v + 0.5));
        else if (k == "can_absorb") t.can_absorb = bool(int(v + 0.5));
        else if (k == "step_reward") t.step_reward = v;
        else if (k == "kill_reward") t.kill_reward = v;
        else if (k == "dead_penalty") t.dead_penalty = v;
        else if (k == "attack_penalty") t.attack_penalty = v;
        else if (k == "food_supply") t.food_supply = v;
        else if (k == "eat_ability") t.eat_ability = v;
        else if (k == "hear_radius" || k == "speak_radius" || k == "speak_ability" || k == "trace" || k == "view_x_offset" || k == "view_y_offset" || k == "att_x_offset" ||
                 k == "att_y_offset" || k == "turn_x_offset" || k == "turn_y_offset") {
            // accepted like the reference; never read on this path (offsets are recomputed, AgentType.cc:106-108)
        } else fatal("invalid agent config in AgentType::AgentType : %s", keys[i]);
    }
    if (t.width < 1 || t.length < 1 || t.width > 16 || t.length > 16) fatal("agent type %s: body %dx%d out of range", name, t.width, t.length);
    // AgentType.cc:84-104: a circle for angle 360, a sector below 180.  (A type registered without an attack range keeps the
    // defaults attack_radius = 0, attack_angle = 0 and gets SectorRange(0, 0): 0 rows, no attack action -- examples/train_trans.py)
    const int parity = t.width % 2;
    if (t.view_angle >= 180) {
        if (std::fabs(t.view_angle - 360) > 1e-5) fatal("only supports ranges with angle = 360, when angle > 180.");
        t.view.circle(t.view_radius, 0, parity);
    } else t.view.sector(t.view_angle, t.view_radius, parity);
    if (t.attack_angle >= 180) {
        if (std::fabs(t.attack_angle - 360) > 1e-5) fatal("only supports ranges with angle = 360, when angle > 180.");
        t.attack.circle(t.attack_radius, t.width / 2.0f, parity);
    } else t.attack.sector(t.attack_angle, t.attack_radius, parity);
    if (t.view.width < 1 || t.view.height < 1) fatal("agent type %s: empty view range", name);
    t.move.circle(t.speed, 0, 1);
    t.view_x_offset = t.att_x_offset = t.width / 2;
    t.view_y_offset = t.att_y_offset = t.length / 2;
    t.attack_base = t.move.count + (turn_mode ? 2 : 0);  // move | (turn_mode: turn left, turn right) | attack (AgentType.cc:110-118)
    t.n_action = t.attack_base + t.attack.count;
    if (t.n_action > PEND_ARG) fatal("action space too large");
    types[name] = t;
}

void Env::new_group(const char *type_name, int *handle) {
    auto it = types.find(type_name);
    if (it == types.end()) fatal("invalid name of agent type in new_group : %s", type_name);
    if ((int)groups.size() >= MAXG) fatal("at most %d groups are supported", MAXG);
    *handle = (int)groups.size();
    HostGroup g;
    g.type = &it->second;
    groups.push_back(g);
}

int Env::n_channel() const { return 1 + (food_mode ? 1 : 0) + (int)groups.size() * (minimap_mode ? 3 : 2); }  // GridWorld.cc:915-924
int Env::feature_size(int g) const { return embedding_size + groups[g].type->n_action + 1 + (goal_mode ? 2 : 0) + (minimap_mode ? 2 : 0); }  // GridWorld.cc:926-934

// RewardEngine.cc:28-69
void Env::define_agent_symbol(int no, int group, int index) {
    if (no >= (int)symbols.size()) symbols.resize(no + 1);
    symbols[no] = {group, index};
}
void Env::define_event_node(int no, int op, int *inputs, int n) {
    if (no >= (int)nodes.size()) nodes.resize(no + 1);
    nodes[no].op = op;
    for (int i = 0; i < n; i++) nodes[no].raw.push_back(inputs[i]);
}
void Env::add_reward_rule(int on, int *recv, float *val, int n, bool terminal) {
    HostRule r;
    r.on = on; r.terminal = terminal;
    for (int i = 0; i < n; i++) { r.recv.push_back(recv[i]); r.val.push_back(val[i]); }
    rules.push_back(r);
}

// ------------------------------------------------------------------------------------------------ device buffers

void Env::free_group(HostGroup &g) {
    GroupDev &c = g.cur, &a = g.alt;
    dfree(arena, c.x); dfree(arena, c.y); dfree(arena, c.id); dfree(arena, c.last_action); dfree(arena, c.op_obj); dfree(arena, c.pend); dfree(arena, c.hp);
    dfree(arena, c.next_reward); dfree(arena, c.last_reward); dfree(arena, c.dead); dfree(arena, c.last_op); dfree(arena, c.key); dfree(arena, c.drank_a);
    dfree(arena, c.drank_b); dfree(arena, c.mv); dfree(arena, c.hits); dfree(arena, c.hitf); dfree(arena, c.absorbed); dfree(arena, a.absorbed); dfree(arena, c.dir); dfree(arena, a.dir);
    dfree(arena, c.eat); dfree(arena, c.fleft); dfree(arena, c.fcell);
    dfree(arena, g.pl.rec); dfree(arena, g.pl.atk); dfree(arena, g.pl.hmask); dfree(arena, g.pl.hlist);
    ptab_valid = false;
    dfree(arena, a.x); dfree(arena, a.y); dfree(arena, a.id); dfree(arena, a.last_action); dfree(arena, a.hp); dfree(arena, a.next_reward); dfree(arena, a.last_reward);
    g.cap = 0; g.n = 0;
}


// the scratch of the step of plain games (launch.h: PlainGroup), for a group of capacity `cap` whose first n records are kept
void Env::plain_arrays(HostGroup &g, size_t n, size_t cap) {
    (void)n;
    regrow(arena, g.pl.rec, 0, cap);                                     // (every record is written by k_plain_rank before anybody reads it)
    regrow(arena, g.pl.atk, 0, cap);
    regrow(arena, g.pl.hmask, 0, cap);
    HIP_OK(hipMemset(g.pl.hmask, 0, sizeof(unsigned) * cap));            // (every mask is zero between steps: k_strike leaves them so)
    regrow(arena, g.pl.hlist, 0, cap * (size_t)std::max(1, plain_slots));
    ptab_valid = false;
}

void Env::ensure_capacity(HostGroup &g, int need) {
    if (need <= g.cap) {
        if (plain_world && !g.pl.rec && g.cap > 0) { HIP_OK(hipStreamSynchronize(stream)); plain_arrays(g, 0, (size_t)g.cap); }
        return;
    }
    HIP_OK(hipStreamSynchronize(stream));
    size_t ncap = std::max<size_t>(std::max<size_t>(need, (size_t)g.cap * 2), 1024);
    size_t n = g.n;
    GroupDev &c = g.cur, &a = g.alt;
    regrow(arena, c.x, n, ncap); regrow(arena, c.y, n, ncap); regrow(arena, c.id, n, ncap); regrow(arena, c.last_action, n, ncap);
    regrow(arena, c.op_obj, n, ncap); regrow(arena, c.pend, n, ncap); regrow(arena, c.hp, n, ncap); regrow(arena, c.next_reward, n, ncap);
    regrow(arena, c.last_reward, n, ncap); regrow(arena, c.dead, n, ncap); regrow(arena, c.last_op, n, ncap); regrow(arena, c.key, n, ncap);
    regrow(arena, c.drank_a, n, ncap); regrow(arena, c.drank_b, n, ncap); regrow(arena, c.mv, n, ncap); regrow(arena, c.hits, n, ncap);
    HIP_OK(hipMemset(c.hits, 0, sizeof(int) * ncap));
    regrow(arena, c.hitf, 0, ncap);
    HIP_OK(hipMemset(c.hitf, 0, ncap));            // (zero between steps: attack_apply_body leaves them so)
    regrow(arena, c.absorbed, n, ncap); regrow(arena, a.absorbed, 0, ncap);
    if (turn_mode) { regrow(arena, c.dir, n, ncap); regrow(arena, a.dir, 0, ncap); }
    regrow(arena, c.eat, 0, ncap); regrow(arena, c.fleft, 0, ncap); regrow(arena, c.fcell, 0, ncap);   // attack-phase scratch (food_mode)
    if (plain_world) plain_arrays(g, g.pl.rec ? n : 0, ncap);
    regrow(arena, a.x, 0, ncap); regrow(arena, a.y, 0, ncap); regrow(arena, a.id, 0, ncap); regrow(arena, a.last_action, 0, ncap);
    regrow(arena, a.hp, 0, ncap); regrow(arena, a.next_reward, 0, ncap); regrow(arena, a.last_reward, 0, ncap);
    g.cap = (int)ncap;
    tables_valid = false;
}

WorldView Env::view() const {
    WorldView W{};
    W.w = width; W.h = height; W.G = (int)groups.size();
    W.occ = d_occ; W.viewcell = d_viewcell; W.claim = d_claim; W.hitbits = d_hit; W.delta = d_delta; W.mask = d_mask; W.counters = d_counters;
    W.any_kill_supply = any_kill_supply;
    W.any_multicell = any_multicell;
    W.any_absorb = any_absorb;
    W.food_mode = food_mode ? 1 : 0;
    W.food = d_food; W.food_next = d_food ? d_food + (size_t)width * height : nullptr;
    W.large_map = large_map_mode; W.bandwidth = bandwidth;
    W.turn_mode = turn_mode ? 1 : 0;
    W.reach = map_reach;
    W.vc_packed = (groups.size() <= 3 && !any_absorb) ? 1 : 0;
    W.live_paint = live_paint_now ? 1 : 0;   // (set for the length of a step whose painted map was current at its start)
    W.plain = plain_world ? 1 : 0;
    int nodes = 0;
    for (int g = 0; g < W.G; g++) {
        W.type[g] = groups[g].tdev;
        W.grp[g] = groups[g].cur;
        W.grp[g].n = groups[g].n;
        W.node_base[g] = nodes;
        nodes += groups[g].cap * groups[g].tdev.bw * groups[g].tdev.bl;
    }
    W.mv_nodes = d_mvnodes;      // (sized by move_nodes() before a step of the generic move resolution)
    return W;
}

// the node array of the generic move resolution's candidate lists (WorldView::mv_nodes): one node per agent and body cell, by capacity --
// only worlds whose moves take the generic path ever allocate it
void Env::move_nodes() {
    if (!any_multicell) return;
    size_t nodes = 0;
    for (auto &G : groups) nodes += (size_t)G.cap * G.tdev.bw * G.tdev.bl;
    if (nodes >= (1u << 31)) fatal("too many body cells for the move phase's candidate lists");
    if (nodes > mvnodes_cap) { enter(); grow(arena, d_mvnodes, mvnodes_cap, nodes, stream); }
}

void Env::ensure_tables() {
    if (tables_valid) return;
    launch_set_tables(stream, view(), d_gtab, d_ttab);
    tables_valid = true;
}

int *Env::read_counters() {
    HIP_OK(hipMemcpyAsync(h_counters, d_counters, sizeof(int) * CTR_TOTAL, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return h_counters;
}

bool Env::read_changed() {
    HIP_OK(hipMemcpyAsync(h_counters, d_counters, sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    return h_counters[CTR_CHANGED] != 0;
}

void Env::clear_changed() { HIP_OK(hipMemsetAsync(d_counters + CTR_CHANGED, 0, sizeof(int), stream)); }

// ------------------------------------------------------------------------------------------------ reset / placement
// GridWorld::reset (GridWorld.cc:72-118) + Map::reset (Map.cc:23-47).  Does not reseed the RNG.
void Env::reset() {
    if (width <= 2 || height <= 2) fatal("map_width / map_height must be configured before reset");
    if ((long long)width * height > (1ll << 30)) fatal("map too large");
    init_device();
    enter();
    HIP_OK(hipStreamSynchronize(stream));
    id_counter = 0;
    for (auto &c : serial_calls) if (c.actions) { int *buf = const_cast<int *>(c.actions); dfree(arena, buf); }
    serial_calls.clear(); step_calls.clear(); serial_calls_on = false;
    alive_valid = false;
    map_scattered = map_warm = false;
    placed_random = placed_total = 0;
    pipe_sweep_shape = -1;
    // a fresh episode starts with two pairs of optimistic attack rounds: the first steps of a dense placement hold the deepest
    // dependency chains (measured at 2 x 400k: one pair runs out once in the first few steps, two never did), and a step that runs
    // out costs a host round trip; the budget falls back to one pair after 64 steps that did not need the second
    boost_window = 64;
    boost_attack = boost_window;
    boost_ran_out = false;
    file_ct++; frame_ct = 0;   // RenderGenerator::next_file (GridWorld.cc:97)
    large_map_mode = width * height > 99 * 99;
    const int n_sep = large_map_mode ? (width * height > 1000 * 1000 ? 16 : 8) : 1;
    bandwidth = (width + n_sep - 1) / n_sep;
    const size_t ncell = (size_t)width * height;
    if (ncell != map_cells) {
        dfree(arena, d_occ); dfree(arena, d_viewcell); dfree(arena, d_claim); dfree(arena, d_hit);
        HIP_OK(dev_malloc(arena, &d_occ, sizeof(int) * ncell));
        HIP_OK(dev_malloc(arena, &d_viewcell, sizeof(int2) * ncell));
        HIP_OK(dev_malloc(arena, &d_claim, sizeof(unsigned long long) * ncell));
        HIP_OK(dev_malloc(arena, &d_hit, sizeof(unsigned) * ncell));
        dfree(arena, d_food);
        map_cells = ncell;
    }
    HIP_OK(hipMemset(d_hit, 0, sizeof(unsigned) * ncell));
    claim_clean = claim_epochs = false; hit_clean = true;
    if (food_mode && !d_food) HIP_OK(dev_malloc(arena, &d_food, sizeof(float) * 2 * ncell));   // amounts, then the attack phase's scratch
    if (d_food) HIP_OK(hipMemset(d_food, 0, sizeof(float) * 2 * ncell));
    h_occ.assign(ncell, OCC_EMPTY);
    for (int i = 0; i < width; i++) { h_occ[i] = OCC_WALL; h_occ[(size_t)(height - 1) * width + i] = OCC_WALL; }
    for (int i = 0; i < height; i++) { h_occ[(size_t)i * width] = OCC_WALL; h_occ[(size_t)i * width + width - 1] = OCC_WALL; }
    h_occ_valid = true;
    upload_occ();

    // per-type constant tables (action deltas, view masks) for the groups of this game
    std::vector<int2> delta;
    std::vector<unsigned char> mask;
    any_kill_supply = 0; any_multicell = 0; any_absorb = 0;
    int total_attack = 0;
    for (auto &g : groups) {
        HostType &t = *g.type;
        TypeDev d{};
        d.hp = t.hp; d.damage = t.damage; d.step_recover = t.step_recover; d.kill_supply = t.kill_supply;
        d.kill_reward = t.kill_reward; d.dead_penalty = t.dead_penalty; d.attack_penalty = t.attack_penalty;
        d.step_reward = t.step_reward; d.attack_in_group = t.attack_in_group;
        d.food_supply = t.food_supply; d.eat_ability = t.eat_ability;
        d.bw = t.width; d.bl = t.length;
        d.can_absorb = t.can_absorb;
        if (t.width * t.length > 1) any_multicell = 1;
        if (t.can_absorb) any_absorb = any_multicell = 1;   // goals: the generic move resolution knows how movers are taken in
        d.n_move = t.move.count; d.n_attack = t.attack.count; d.n_turn = turn_mode ? 2 : 0;
        d.move_off = (int)delta.size();
        for (int k = 0; k < t.move.count; k++) delta.push_back(make_int2(t.move.dx[k], t.move.dy[k]));
        d.attack_bit = total_attack;
        d.attack_off = (int)delta.size();
        for (int k = 0; k < t.attack.count; k++) delta.push_back(make_int2(t.attack.dx[k] + t.att_x_offset, t.attack.dy[k] + t.att_y_offset));
        d.view_w = t.view.width; d.view_h = t.view.height;
        d.view_x1 = t.view.x1 + t.view_x_offset; d.view_y1 = t.view.y1 + t.view_y_offset;
        d.mask_off = (int)mask.size();
        mask.insert(mask.end(), t.view.in.begin(), t.view.in.end());
        g.tdev = d;
        if (t.kill_supply != 0) any_kill_supply = 1;
        total_attack += t.attack.count;
        g.n = 0; g.group_reward = 0; g.acted = false; g.h_dead = 0; g.h_taken = 0; g.indexed = 0; g.sa_off = -1;
    }
    // most hits one target can receive: attack offsets of every group allowed to attack it
    attack_kmax = 1;
    for (size_t t = 0; t < groups.size(); t++) {
        int k = 0;
        for (size_t a = 0; a < groups.size(); a++)
            if (a != t || groups[a].type->attack_in_group || food_mode) k += groups[a].type->attack.count;
        k *= groups[t].type->width * groups[t].type->length;   // every body cell can be hit with every offset
        attack_kmax = std::max(attack_kmax, k);
    }
    if (food_mode) attack_kmax = std::max(attack_kmax, total_attack);   // a food cell is hit by every group
    map_reach = 0;
    if (turn_mode) {
        // an attack bit may stand for one attacker per direction.  The lists hold at most 256 hits: beyond that the worst case is not
        // covered by construction any more, and an overflow is reported at the end of the step (CTR_HIT_OVERFLOW) instead
        attack_kmax = std::min(attack_kmax * DIR_NUM, 256);
        // how far the top-left cell of a body can be from a cell its move or its turn enters (neighbourhood scans, kernels_dev.h)
        for (auto &g : groups) {
            const HostType &t = *g.type;
            int far = 0;
            for (int k = 0; k < t.move.count; k++) far = std::max(far, std::max(std::abs(t.move.dx[k]), std::abs(t.move.dy[k])));
            // a move shifts the top-left cell by `far`, a turn by up to M - 1 (the body is re-laid about its reference corner);
            // the entered cell lies up to M - 1 further inside the new rectangle
            const int M = std::max(t.width, t.length);
            map_reach = std::max(map_reach, std::max(far, M - 1) + M - 1);
        }
    }
    if (attack_kmax > 256) fatal("attack ranges x body size too large for the LDS hit lists (%d > 256)", attack_kmax);
    if (!attack_lds_ok(attack_kmax)) fatal("attack ranges x body size (%d hits per target) need more LDS per workgroup than this device grants", attack_kmax);
    if (total_attack > ATTACK_KMAX_HOST) fatal("sum of attack-range sizes (%d) exceeds the engine limit %d", total_attack, ATTACK_KMAX_HOST);
    if (n_channel() > 32) fatal("too many observation channels");
    // the step of plain games (step.hip) keeps scratch of its own per agent, sized by the attack offsets of all groups
    {
        const bool plain = !any_multicell && !turn_mode && !food_mode && !any_absorb && !any_kill_supply && plain_eval_lds_ok(attack_kmax);
        const int slots = std::max(1, total_attack);
        for (auto &g : groups) {
            if (g.pl.rec && (!plain || slots != plain_slots)) {      // (the configuration changed between two resets: built anew when agents are added)
                dfree(arena, g.pl.rec); dfree(arena, g.pl.atk); dfree(arena, g.pl.hmask); dfree(arena, g.pl.hlist);
            }
            if (g.pl.rec) HIP_OK(hipMemset(g.pl.hmask, 0, sizeof(unsigned) * g.cap));
        }
        plain_world = plain; plain_slots = slots;
        ptab_valid = false;
    }
    dfree(arena, d_delta); dfree(arena, d_mask);
    HIP_OK(dev_malloc(arena, &d_delta, sizeof(int2) * std::max<size_t>(delta.size(), 1)));
    HIP_OK(dev_malloc(arena, &d_mask, std::max<size_t>(mask.size(), 1)));
    if (!delta.empty()) HIP_OK(hipMemcpy(d_delta, delta.data(), sizeof(int2) * delta.size(), hipMemcpyHostToDevice));
    if (!mask.empty()) HIP_OK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
    HIP_OK(hipMemset(d_counters, 0, sizeof(int) * CTR_TOTAL));
    rng_on_device = false;
    move_seq_base = 0;
    if (!rules_compiled) {   // once, like init_reward_description
        compile_rules();
        rules_compiled = true;
        dfree(arena, d_rule_args); dfree(arena, d_rule_progs);
        HIP_OK(dev_malloc(arena, &d_rule_args, sizeof(RuleArgs) * std::max<size_t>(rule_args.size(), 1)));
        HIP_OK(dev_malloc(arena, &d_rule_progs, sizeof(RuleProg) * std::max<size_t>(rule_progs.size(), 1)));
        if (!rule_args.empty()) HIP_OK(hipMemcpy(d_rule_args, rule_args.data(), sizeof(RuleArgs) * rule_args.size(), hipMemcpyHostToDevice));
        if (!rule_progs.empty()) HIP_OK(hipMemcpy(d_rule_progs, rule_progs.data(), sizeof(RuleProg) * rule_progs.size(), hipMemcpyHostToDevice));
    }
    // threads of the one-launch step that evaluate hit lists: as many as kmax x threads x 8 B of LDS allow
    {
        int dev_lds = 0;
        HIP_OK(hipDeviceGetAttribute(&dev_lds, hipDeviceAttributeMaxSharedMemoryPerBlock, device_id));
        const int budget = std::max(0, dev_lds - solo_step_static_lds() - 256);
        solo_nt_eval = std::min(1024, budget / (attack_kmax * 8) / 64 * 64);
        if (solo_nt_eval >= 64 && (size_t)attack_kmax * solo_nt_eval * 8 > (48u << 10) && !solo_step_allow_lds((size_t)attack_kmax * solo_nt_eval * 8)) {
            solo_nt_eval = std::min(1024, (48 << 10) / (attack_kmax * 8) / 64 * 64);   // stay under the default limit
        }
    }
    tables_valid = false;
    paint_valid = false; mini_valid = false;
}

void Env::download_occ() {
    if (h_occ_valid) return;
    HIP_OK(hipStreamSynchronize(stream));
    HIP_OK(hipMemcpy(h_occ.data(), d_occ, sizeof(int) * h_occ.size(), hipMemcpyDeviceToHost));
    h_occ_valid = true;
}

void Env::upload_occ() {
    HIP_OK(hipMemcpy(d_occ, h_occ.data(), sizeof(int) * h_occ.size(), hipMemcpyHostToDevice));
    paint_valid = false; mini_valid = false;
}

// Map::is_blank_area (Map.cc:454-470)
bool Env::host_blank(int x, int y, int bw, int bl) const {
    if (x < 0 || y < 0 || x + bw >= width || y + bl >= height) return false;
    for (int i = 0; i < bw; i++)
        for (int j = 0; j < bl; j++)
            if (h_occ[(size_t)(y + j) * width + x + i] != OCC_EMPTY) return false;
    return true;
}

// Map::get_random_blank (Map.cc:49-63): two RNG draws per try
void Env::host_random_blank(int bw, int bl, int &ox, int &oy) {
    int tries = 0;
    while (true) {
        rng_on_device = false;   // the host draws: its copy of the engine state is the current one again
        int x = (int)rng() % (width - bw);
        int y = (int)rng() % (height - bl);
        if (host_blank(x, y, bw, bl)) { ox = x; oy = y; return; }
        if (tries++ > width * height) fatal("cannot find a blank position in a filled map");
    }
}

// GridWorld::add_agents (GridWorld.cc:180-290).  Cold path: placement is defined sequentially by the reference.
// GridWorld::set_goal (GridWorld.cc:667-679; "deprecated" there): "random" draws a goal position for every agent of the group -- dead ones
// that clear_dead has not removed yet included -- with two draws of the engine's generator each.  Nothing in the reference ever reads a
// goal back (Agent::get_goal has no caller), so what the call leaves behind is the generator, 2 n draws further on.
void Env::set_goal(int group, const char *method) {
    if (!device_ready) fatal("set_goal called before reset");
    if (group < 0 || group >= (int)groups.size()) fatal("invalid group handle in GridWorld::set_goal : %d", group);
    if (std::string(method) != "random") fatal("invalid goal type in GridWorld::set_goal");
    enter();
    rng_on_device = false;
    rng.skip(2u * (unsigned)groups[group].n);
}

void Env::add_agents(int group, int n, const char *method, const int *px, const int *py, const int *pdir) {
    if (!device_ready) fatal("add_agents called before reset");
    enter();
    alive_valid = false;            // (the groups change: k_strike's survivor counts no longer describe them)
    download_occ();
    std::string m(method);
    auto add_wall = [&](int x, int y) {  // Map::add_wall (Map.cc:108-115)
        if (x < 0 || x >= width || y < 0 || y >= height) fatal("wall position (%d, %d) out of the map", x, y);
        int &c = h_occ[(size_t)y * width + x];
        if (c >= 0 || c == OCC_FOOD) return;   // occupied by an agent (or by food): ignored
        c = OCC_WALL;
    };
    if (group == -1) {
        if (m == "random") { for (int i = 0; i < n; i++) { int x, y; host_random_blank(1, 1, x, y); add_wall(x, y); } }
        else if (m == "custom") { for (int i = 0; i < n; i++) add_wall(px[i], py[i]); }
        else if (m == "fill") { for (int x = px[0]; x < px[0] + px[2]; x++) for (int y = px[1]; y < px[1] + px[3]; y++) add_wall(x, y); }
        else fatal("unsupported method in GridWorld::add_agents : %s", method);
        upload_occ();
        return;
    }
    if (group < 0 || group >= (int)groups.size()) fatal("invalid group handle in GridWorld::add_agents : %d", group);
    HostGroup &G = groups[group];
    std::vector<int> sx, sy, sid, sdir;
    // turn_mode: every agent faces a direction of its own; a body lying east-west has its footprint transposed (Map.cc:589-599)
    auto place = [&](int x, int y, int dir) {     // add_or_error: occupied positions are silently skipped, the id is reused
        const bool upright = dir == DIR_NORTH || dir == DIR_SOUTH;
        const int bw = upright ? G.type->width : G.type->length, bl = upright ? G.type->length : G.type->width;
        if (!host_blank(x, y, bw, bl)) return;
        const int ref = ref_pack(group, G.n + (int)sx.size());
        for (int i = 0; i < bw; i++) for (int j = 0; j < bl; j++) h_occ[(size_t)(y + j) * width + x + i] = ref;
        sx.push_back(x); sy.push_back(y); sid.push_back(id_counter++); sdir.push_back(dir);
    };
    if (m == "random") {
        placed_random += n;                   // agents that stand next to each other in the group stand anywhere on the map (observe_device)
        for (int i = 0; i < n; i++) {
            rng_on_device = false;
            const int dir = turn_mode ? (int)(rng() % DIR_NUM) : DIR_NORTH;   // drawn before the position (GridWorld.cc:230)
            const bool upright = dir == DIR_NORTH || dir == DIR_SOUTH;
            int x, y;
            host_random_blank(upright ? G.type->width : G.type->length, upright ? G.type->length : G.type->width, x, y);
            place(x, y, dir);
        }
    } else if (m == "custom") {
        for (int i = 0; i < n; i++) {
            if (pdir && pdir[i] >= DIR_NUM) fatal("invalid direction in GridWorld::add_agent");
            place(px[i], py[i], turn_mode && pdir ? pdir[i] : DIR_NORTH);
        }
    } else if (m == "fill") {
        const int dir = turn_mode ? px[4] : DIR_NORTH;
        if (dir < 0 || dir >= DIR_NUM) fatal("invalid direction in GridWorld::add_agent");
        const bool upright = dir == DIR_NORTH || dir == DIR_SOUTH;
        const int bw = upright ? G.type->width : G.type->length, bl = upright ? G.type->length : G.type->width;
        for (int x = px[0]; x < px[0] + px[2]; x += bw) for (int y = px[1]; y < px[1] + px[3]; y += bl) place(x, y, dir);
    } else fatal("unsupported method in GridWorld::add_agents : %s", method);

    const int k = (int)sx.size();
    placed_total += k;
    // (the map counts as scattered while at least a quarter of everything placed since the reset was placed at random: a few random
    // agents beside a large formation do not make every step pay the stream-through -- ADVICE round 5)
    map_scattered = placed_random > 0 && 4ll * placed_random >= placed_total;
    if (G.n + k > REF_MASK) fatal("too many agents in one group");
    if (k > 0) {
        ensure_capacity(G, G.n + k);
        HIP_OK(hipStreamSynchronize(stream));
        const HostType &t = *G.type;
        GroupDev &c = G.cur;
        const size_t o = G.n;
        auto up = [&](auto *dst, const auto &vec) {
            HIP_OK(hipMemcpy(dst + o, vec.data(), sizeof(vec[0]) * vec.size(), hipMemcpyHostToDevice));
        };
        up(c.x, sx); up(c.y, sy); up(c.id, sid);
        up(c.hp, std::vector<float>(k, t.hp));
        up(c.last_action, std::vector<int>(k, t.n_action));          // GridWorld.h:140 "dangerous here !"
        up(c.next_reward, std::vector<float>(k, t.step_reward));     // Agent ctor -> init_reward (GridWorld.h:168-174)
        up(c.last_reward, std::vector<float>(k, 0.0f));
        up(c.op_obj, std::vector<int>(k, -1));
        up(c.pend, std::vector<int>(k, PEND_NONE));
        up(c.dead, std::vector<unsigned char>(k, 0));
        up(c.absorbed, std::vector<unsigned char>(k, 0));
        if (turn_mode) up(c.dir, sdir);
        up(c.last_op, std::vector<unsigned char>(k, (unsigned char)OP_NULL));
        G.n += k;
        tables_valid = false;
    }
    upload_occ();
}

}  // namespace magent_amd
