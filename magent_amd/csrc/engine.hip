// engine.hip -- host side of the MI355X grid-world engine: configuration, agent types, placement, per-call
// orchestration of the kernels in render.hip / step.hip / cycle.hip on one HIP stream per environment.
//
// Mirrors the behaviour of the reference's GridWorld class (src/gridworld/GridWorld.{h,cc}) for the hot-path scope
// of SURVEY.md section 8.  The product never falls back to a CPU engine: every state-changing operation after
// placement runs on the GPU.  The only host-side algorithmic pieces are the cold-path placement (add_agents, which
// the reference defines as sequential rejection sampling on the engine RNG) and, in this round, the attack
// shuffle's permutation (a function of the RNG state and the attack count only).
#include <algorithm>
#include <chrono>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <functional>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "engine_host.h"

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

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) fatal("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

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
// (every allocation names the arena of the environment it belongs to: an environment only ever frees what it allocated)
template <class T>
static hipError_t dev_malloc(DevArena &arena, T **p, size_t bytes) {
    if (bytes <= DevArena::SMALL) { *p = (T *)arena.take(bytes ? bytes : 1); return hipSuccess; }
    return hipMalloc(p, bytes);
}
static void dev_free(DevArena &arena, void *p) {
    if (arena.owns(p)) return;      // (arena memory goes back with the environment)
    (void)hipFree(p);
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
struct Env::ProfScope {
    Env &e; Env::ProfSlot *slot = nullptr; hipEvent_t a{}, b{}; hipStream_t s{};
    ProfScope(Env &env, const char *name, bool dominant = false, hipStream_t on = nullptr) : e(env), s(on ? on : env.stream) {
        if (!e.prof_level || (e.prof_level == 2 && !dominant)) return;   // an event pair costs ~10 us of stream time
        slot = &e.prof[name];
        a = e.prof_event(); b = e.prof_event();
        HIP_OK(hipEventRecord(a, s));
    }
    ~ProfScope() {
        if (!slot) return;
        HIP_OK(hipEventRecord(b, s));
        slot->pending.emplace_back(a, b);
    }
};

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

template <class T>
static void dfree(DevArena &arena, T *&p) {
    if (p) { dev_free(arena, p); p = nullptr; }
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

// A rule shape the GPU kernels do not take (several iterated symbols, 'all' / fixed-index symbols, in_a_line, receivers in
// the subject's and the object's group at once ...) is not refused: ALL rules of such a game are evaluated on the host by
// the reference's recursive search (Env::eval_rules_host) -- all of them, because the float adds of different rules on
// one agent have to keep their order.
namespace { struct RuleGoesToHost { char why[256]; }; }
[[noreturn]] static void to_host(const char *fmt, ...) {
    RuleGoesToHost e;
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(e.why, sizeof(e.why), fmt, ap);
    va_end(ap);
    throw e;
}

// translate the rule shapes the kernels take into kernel arguments; the other shapes send the game's rules to the host
void Env::compile_rules() {
    rule_args.clear();
    rule_progs.clear();
    rules_on_host = false;
    try {
        compile_rules_gpu();
    } catch (const RuleGoesToHost &e) {
        rules_on_host = true;
        rule_args.clear();
        rule_progs.clear();
        if (std::getenv("MAGENT_VERBOSE")) std::fprintf(stderr, "magent-amd: reward rules evaluated on the host (%s)\n", e.why);
    }
    if (rules_on_host) plan_host_rules();
}

void Env::compile_rules_gpu() {
    if ((int)rules.size() > CTR_TRIGGER_END - CTR_TRIGGER) to_host("too many reward rules");
    for (size_t k = 0; k < rules.size(); k++) {
        const HostRule &r = rules[k];
        if (r.on < 0 || r.on >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
        const HostNode &on = nodes[r.on];
        auto binary = [&](const HostNode &n) { return (n.op == OP_ATTACK || n.op == OP_KILL || n.op == OP_COLLIDE) && n.raw.size() == 2; };
        auto any_sym = [&](int no) {
            if (no < 0 || no >= (int)symbols.size()) fatal("reward rule %zu refers to an undefined agent symbol", k);
            const HostSymbol &sy = symbols[no];
            if (sy.index != -1) to_host("reward rule %zu: only 'any' agent symbols are on the GPU path", k);
            if (sy.group < 0 || sy.group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
            return sy.group;
        };
        if (on.op == 0 /* and */ && on.raw.size() == 2 && on.raw[0] >= 0 && on.raw[1] >= 0 && on.raw[0] < (int)nodes.size() &&
            on.raw[1] < (int)nodes.size() && binary(nodes[on.raw[0]]) && binary(nodes[on.raw[1]])) {
            // Event(a, p, c) & Event(b, q, c): "two agents act on the same third" (builtin/config/double_attack.py:33-40)
            const HostNode *e1 = &nodes[on.raw[0]], *e2 = &nodes[on.raw[1]];
            if (e1->raw[1] != e2->raw[1] || e1->raw[0] == e2->raw[0] || e1->raw[0] == e1->raw[1] || e2->raw[0] == e2->raw[1]) {
                compile_rule_program(k);   // not "two agents on one object": a general expression, if its search iterates one symbol
                continue;
            }
            if (e2->raw[0] < e1->raw[0]) std::swap(e1, e2);   // the search binds symbols in ascending number (RewardEngine.cc:155-189)
            RuleArgs a{};
            a.prog = -1;
            a.pair = 1; a.rule_no = (int)k;
            a.ga = any_sym(e1->raw[0]); a.op = e1->op;
            a.gy = any_sym(e2->raw[0]); a.op_y = e2->op;
            a.gb = any_sym(e1->raw[1]);
            for (size_t i = 0; i < r.recv.size(); i++) {
                int *cnt; float *val;
                if (r.recv[i] == e1->raw[0]) { cnt = &a.n_subj; val = a.v_subj; }
                else if (r.recv[i] == e2->raw[0]) { cnt = &a.n_y; val = a.v_y; }
                else if (r.recv[i] == e1->raw[1]) { cnt = &a.n_obj; val = a.v_obj; }
                else to_host("reward rule %zu: a receiver must be a symbol of the event", k);
                if (*cnt == 4) to_host("too many receivers");
                val[(*cnt)++] = r.val[i];
            }
            if (a.n_obj && (a.gb == a.ga || a.gb == a.gy))
                to_host("reward rule %zu: paying the shared object inside a subject's group interleaves float adds; not on the GPU path", k);
            rule_args.push_back(a);
            continue;
        }
        if (!binary(on)) {   // a general expression: on the GPU path when its search iterates a single symbol
            compile_rule_program(k);
            continue;
        }
        const int group_a = any_sym(on.raw[0]), group_b = any_sym(on.raw[1]);
        // one symbol as subject AND object: the reference binds the object over the subject's entity (RewardEngine.cc:17-24,
        // 405-408) and then tests the TARGET against itself -- it fires for an agent whose target hit itself (bodies whose
        // in-group attack range covers their own cells): the program form knows that shape
        if (on.raw[0] == on.raw[1]) { compile_rule_program(k); continue; }
        RuleArgs a{};
        a.prog = -1;
        a.ga = group_a; a.gb = group_b; a.op = on.op; a.rule_no = (int)k;
        for (size_t i = 0; i < r.recv.size(); i++) {
            if (r.recv[i] == on.raw[0]) { if (a.n_subj == 4) to_host("too many receivers"); a.v_subj[a.n_subj++] = r.val[i]; }
            else if (r.recv[i] == on.raw[1]) { if (a.n_obj == 4) to_host("too many receivers"); a.v_obj[a.n_obj++] = r.val[i]; }
            else to_host("reward rule %zu: a receiver must be the subject or the object of the event", k);
        }
        if (a.n_subj && a.n_obj && a.ga == a.gb)
            to_host("reward rule %zu: subject and object receivers in the same group interleave float adds; not on the GPU path", k);
        rule_args.push_back(a);
    }
}

// A rule whose event is a general expression (and / or / not over attack, kill, collide, die, at, in).  The reference
// plans its search per rule (GridWorld::init_reward_description, RewardEngine.cc:105-214): the symbols of the expression
// in ascending number; a symbol that is the subject of a binary event brings that event's object along ("inferred":
// bound to the subject's op_obj instead of being iterated).  The GPU path takes the rules whose plan iterates ONE
// symbol -- every other symbol is its inferred object -- and evaluates the expression per agent (k_rule_prog).
void Env::compile_rule_program(size_t k) {
    const HostRule &r = rules[k];
    struct Info { std::vector<int> related; std::vector<std::pair<int, int>> infer; };
    std::function<Info(int)> collect = [&](int no) -> Info {
        if (no < 0 || no >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
        const HostNode &n = nodes[no];
        Info I;
        auto add_sym = [&](int s2) { if (std::find(I.related.begin(), I.related.end(), s2) == I.related.end()) I.related.push_back(s2); };
        auto add_inf = [&](std::pair<int, int> p) { for (auto &q : I.infer) if (q.first == p.first) return; I.infer.push_back(p); };
        if (n.op == 0 || n.op == 1 || n.op == 2) {
            const size_t kids = n.op == 2 ? 1 : 2;
            if (n.raw.size() < kids) fatal("reward rule %zu: malformed event node", k);
            for (size_t c = 0; c < kids; c++) {
                Info C = collect(n.raw[c]);
                for (int s2 : C.related) add_sym(s2);
                for (auto &p : C.infer) add_inf(p);
            }
        } else if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) {
            add_sym(n.raw[0]); add_sym(n.raw[1]); add_inf({n.raw[0], n.raw[1]});
        } else if (n.op == 4 || n.op == 5 || n.op == 8) {   // at, in, die
            add_sym(n.raw[0]);
        } else to_host("reward rule %zu: event predicate %d (in_a_line / align) is not on the GPU path", k, n.op);
        std::sort(I.related.begin(), I.related.end());
        std::sort(I.infer.begin(), I.infer.end());
        return I;
    };
    const Info I = collect(r.on);
    std::vector<int> iterated, inferred, added;
    auto has = [&](int s2) { return std::find(added.begin(), added.end(), s2) != added.end(); };
    for (int s2 : I.related) {
        if (has(s2)) continue;
        for (auto &p : I.infer) if (p.first == s2) { iterated.push_back(s2); inferred.push_back(p.second); added.push_back(s2); added.push_back(p.second); break; }
    }
    for (int s2 : I.related) if (!has(s2)) { iterated.push_back(s2); inferred.push_back(-1); }
    if (iterated.size() != 1)
        to_host("reward rule %zu: its search iterates %zu agent symbols; the GPU path takes rules that iterate one symbol "
              "(plus Event(a, p, c) & Event(b, q, c))", k, iterated.size());
    // (sy == sx: an event whose subject is its own object.  The search iterates the symbol and then re-binds it to the
    // iterated agent's op_obj: every leaf and every receiver then means that target -- slot 1)
    const int sx = iterated[0], sy = inferred[0];
    const bool self = sy == sx;
    auto group_of = [&](int no) {
        if (no < 0 || no >= (int)symbols.size()) to_host("reward rule %zu refers to an undefined agent symbol", k);
        if (symbols[no].index != -1) to_host("reward rule %zu: only 'any' agent symbols are on the GPU path", k);
        if (symbols[no].group < 0 || symbols[no].group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
        return symbols[no].group;
    };
    RuleProg P{};
    P.ga = group_of(sx); P.has_obj = sy >= 0; P.gb = sy >= 0 ? group_of(sy) : 0; P.rule_no = (int)k;
    auto slot = [&](int no) { if (no == sy && (self || no != sx)) return 1; if (no == sx) return 0; fatal("reward rule %zu: internal: unplanned symbol", k); return 0; };
    std::function<void(int)> emit = [&](int no) {
        const HostNode &n = nodes[no];
        if (n.op == 0 || n.op == 1) { emit(n.raw[0]); emit(n.raw[1]); }
        else if (n.op == 2) emit(n.raw[0]);
        if (P.n == 24) to_host("reward rule %zu: expression too long", k);
        P.op[P.n] = n.op;
        if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) { P.a[P.n][0] = slot(n.raw[0]); P.a[P.n][1] = slot(n.raw[1]); }
        else if (n.op == 4 || n.op == 5 || n.op == 8) {
            P.a[P.n][0] = slot(n.raw[0]);
            const size_t want = n.op == 4 ? 3 : n.op == 5 ? 5 : 1;
            if (n.raw.size() < want) fatal("reward rule %zu: malformed event node", k);
            for (size_t q = 1; q < want; q++) P.a[P.n][q] = n.raw[q];
        }
        P.n++;
    };
    emit(r.on);
    RuleArgs a{};
    a.prog = (int)rule_progs.size(); a.rule_no = (int)k; a.ga = P.ga; a.gb = P.gb;
    for (size_t i = 0; i < r.recv.size(); i++) {
        if (r.recv[i] == sx && !self) { if (P.n_subj == 4) to_host("too many receivers"); P.v_subj[P.n_subj++] = r.val[i]; }
        else if (r.recv[i] == sy && sy >= 0) { if (P.n_obj == 4) to_host("too many receivers"); P.v_obj[P.n_obj++] = r.val[i]; }
        else to_host("reward rule %zu: a receiver must be a symbol of the event", k);
    }
    if (P.n_subj && P.n_obj && P.ga == P.gb)
        to_host("reward rule %zu: subject and object receivers in the same group interleave float adds; not on the GPU path", k);
    a.n_obj = P.n_obj;
    for (int q = 0; q < P.n_obj; q++) a.v_obj[q] = P.v_obj[q];
    rule_progs.push_back(P);
    rule_args.push_back(a);
}

// ------------------------------------------------------------------------------------------------ rules on the host
// GridWorld::init_reward_description (RewardEngine.cc:105-214): per rule, the order in which the recursive search binds
// the symbols of its event expression -- ascending symbol number; a symbol that is the subject of a binary event brings
// that event's object along (bound to the subject's op_obj instead of being iterated; the first such pair per subject,
// children left to right).
void Env::plan_host_rules() {
    struct Info { std::vector<int> related; std::vector<std::pair<int, int>> infer; };
    host_plans.assign(rules.size(), HostRulePlan{});
    for (size_t k = 0; k < rules.size(); k++) {
        std::function<Info(int)> collect = [&](int no) -> Info {
            if (no < 0 || no >= (int)nodes.size()) fatal("reward rule %zu refers to an undefined event", k);
            const HostNode &n = nodes[no];
            Info I;
            auto sym = [&](int s2) {
                if (s2 < 0 || s2 >= (int)symbols.size()) fatal("reward rule %zu refers to an undefined agent symbol", k);
                if (symbols[s2].group < 0 || symbols[s2].group >= (int)groups.size()) fatal("reward rule %zu: invalid group in agent symbol", k);
                if (std::find(I.related.begin(), I.related.end(), s2) == I.related.end()) I.related.push_back(s2);
            };
            auto inf = [&](std::pair<int, int> p) { for (auto &q : I.infer) if (q.first == p.first) return; I.infer.push_back(p); };
            const size_t want = (n.op == 0 || n.op == 1) ? 2 : n.op == 4 ? 3 : n.op == 5 ? 5 : (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) ? 2 : 1;
            if (n.raw.size() < want) fatal("reward rule %zu: malformed event node", k);
            switch (n.op) {
                case 0: case 1: case 2:
                    for (size_t c = 0; c < (n.op == 2 ? 1u : 2u); c++) {
                        Info C = collect(n.raw[c]);
                        for (int s2 : C.related) sym(s2);
                        for (auto &q : C.infer) inf(q);
                    }
                    break;
                case OP_KILL: case OP_COLLIDE: case OP_ATTACK:
                    sym(n.raw[0]); sym(n.raw[1]); inf({n.raw[0], n.raw[1]});
                    if (symbols[n.raw[1]].index == -2) fatal("reward rule %zu: the object of attack / kill / collide cannot be a whole group (the reference asserts)", k);
                    break;
                case 4: case 5: case 8: sym(n.raw[0]); break;
                case 9:    // in_a_line: a statement about a whole group (the reference asserts is_all)
                    sym(n.raw[0]);
                    if (symbols[n.raw[0]].index != -2) fatal("reward rule %zu: in_a_line takes an 'all' symbol (the reference asserts)", k);
                    break;
                case 10:
                    fatal("reward rule %zu: 'align' reads two counters the reference allocates and never fills (GridWorld.cc:94-95, 955-968): "
                          "it has no defined result to reproduce", k);
                default: fatal("reward rule %zu: invalid event predicate %d", k, n.op);
            }
            std::sort(I.related.begin(), I.related.end());
            std::sort(I.infer.begin(), I.infer.end());
            return I;
        };
        const Info I = collect(rules[k].on);
        HostRulePlan &P = host_plans[k];
        std::vector<int> added;
        auto has = [&](int s2) { return std::find(added.begin(), added.end(), s2) != added.end(); };
        for (int s2 : I.related) {
            if (has(s2)) continue;
            for (auto &q : I.infer)
                if (q.first == s2) { P.order.push_back(s2); P.brings.push_back(q.second); added.push_back(s2); added.push_back(q.second); break; }
        }
        for (int s2 : I.related) if (!has(s2)) { P.order.push_back(s2); P.brings.push_back(-1); }
        for (int rc : rules[k].recv) if (rc < 0 || rc >= (int)symbols.size()) fatal("reward rule %zu: undefined receiver", k);
    }
}

// GridWorld::calc_reward with calc_rule / calc_event_node (GridWorld.cc:681-692, RewardEngine.cc:216-443) on host copies
// of what the rules read -- last_op, op_obj, positions, dead flags -- in the reference's binding order, so that every
// float add lands in the reference's order.  Rewards go back to the device; triggers stay here.
void Env::eval_rules_host() {
    const int NG = (int)groups.size();
    struct Copy { std::vector<unsigned char> last_op, dead, busy; std::vector<int> op_obj, x, y; std::vector<float> reward; bool dirty = false; };
    std::vector<Copy> C(NG);
    HIP_OK(hipStreamSynchronize(stream));
    for (int g = 0; g < NG; g++) {
        const int n = groups[g].n;
        Copy &c = C[g];
        c.last_op.resize(n); c.dead.resize(n); c.busy.assign(n, 0); c.op_obj.resize(n); c.x.resize(n); c.y.resize(n); c.reward.resize(n);
        if (!n) continue;
        const GroupDev &D = groups[g].cur;
        HIP_OK(hipMemcpy(c.last_op.data(), D.last_op, n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.dead.data(), D.dead, n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.op_obj.data(), D.op_obj, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.x.data(), D.x, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.y.data(), D.y, sizeof(int) * n, hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(c.reward.data(), D.next_reward, sizeof(float) * n, hipMemcpyDeviceToHost));
    }
    auto bound = [&](const HostSymbol &sy, size_t k) {
        if (sy.ent_g < 0 || sy.ent_g >= NG || sy.ent_i < 0 || sy.ent_i >= groups[sy.ent_g].n)
            fatal("reward rule %zu reads an agent symbol that no search has bound (the reference follows a dangling pointer here)", k);
    };
    // AgentSymbol::bind_with_check (RewardEngine.cc:14-24)
    // (Agent::index is 0 from the constructor and only clear_dead sets it, GridWorld.h:136 / GridWorld.cc:655: an agent added
    // since the last clear_dead answers 0 here, whatever its position in the group -- HostGroup::indexed)
    auto bind = [&](HostSymbol &sy, int ref) {
        if (sy.group != ref_group(ref)) return false;
        const int stored = ref_index(ref) < groups[ref_group(ref)].indexed ? ref_index(ref) : 0;
        if (sy.index != -1 && sy.index != stored) return false;
        sy.ent_g = ref_group(ref); sy.ent_i = ref_index(ref);
        return true;
    };
    host_triggers.assign(rules.size(), 0);
    for (size_t k = 0; k < rules.size(); k++) {
        const HostRule &R = rules[k];
        const HostRulePlan &P = host_plans[k];
        std::function<bool(int)> holds = [&](int no) -> bool {
            const HostNode &n = nodes[no];
            if (n.op == 0) return holds(n.raw[0]) && holds(n.raw[1]);
            if (n.op == 1) return holds(n.raw[0]) || holds(n.raw[1]);
            if (n.op == 2) return !holds(n.raw[0]);
            const HostSymbol &s0 = symbols[n.raw[0]];
            const Copy &c0 = C[s0.group];
            const int n0 = groups[s0.group].n;
            if (n.op == 9) {   // in_a_line: one column (or one row) of consecutive cells, in any order (RewardEngine.cc:263-296)
                if (n0 < 2) return true;
                const int dx = c0.x[0] - c0.x[1], dy = c0.y[0] - c0.y[1];
                if ((dx == 0) == (dy == 0)) return false;
                const std::vector<int> &fixed = dx == 0 ? c0.x : c0.y, &runs = dx == 0 ? c0.y : c0.x;
                int lo = runs[0], hi = runs[0];
                bool in_line = true;
                for (int i = 1; i < n0 && in_line; i++) { lo = std::min(lo, runs[i]); hi = std::max(hi, runs[i]); in_line = fixed[i] == fixed[0]; }
                return in_line && hi - lo + 1 == n0;
            }
            std::function<bool(int, int)> leaf;
            if (n.op == OP_KILL || n.op == OP_COLLIDE || n.op == OP_ATTACK) {
                const HostSymbol &s1 = symbols[n.raw[1]];
                bound(s1, k);
                const int obj = ref_pack(s1.ent_g, s1.ent_i);
                leaf = [&C, &n, obj](int g, int i) { return C[g].last_op[i] == n.op && C[g].op_obj[i] == obj; };
            } else if (n.op == 8) leaf = [&C](int g, int i) { return C[g].dead[i] != 0; };
            else if (n.op == 4) leaf = [&C, &n](int g, int i) { return C[g].x[i] == n.raw[1] && C[g].y[i] == n.raw[2]; };
            else leaf = [&C, &n](int g, int i) { return C[g].x[i] > n.raw[1] && C[g].x[i] < n.raw[3] && C[g].y[i] > n.raw[2] && C[g].y[i] < n.raw[4]; };
            if (s0.index == -2) {      // 'all': every agent of the group
                for (int i = 0; i < n0; i++) if (!leaf(s0.group, i)) return false;
                return true;
            }
            bound(s0, k);
            return leaf(s0.ent_g, s0.ent_i);
        };
        std::function<void(size_t)> search = [&](size_t depth) {
            if (depth == P.order.size()) {
                if (!holds(R.on)) return;
                host_triggers[k] = 1;
                for (size_t q = 0; q < R.recv.size(); q++) {
                    const HostSymbol &sy = symbols[R.recv[q]];
                    if (sy.index == -2) groups[sy.group].group_reward += R.val[q];        // Group::add_reward
                    else { bound(sy, k); C[sy.ent_g].reward[sy.ent_i] += R.val[q]; C[sy.ent_g].dirty = true; }
                }
                return;
            }
            HostSymbol &sy = symbols[P.order[depth]];
            const int brings = P.brings[depth];
            Copy &c = C[sy.group];
            const int n = groups[sy.group].n;
            if (sy.index == -1) {          // 'any': every agent of the group that no outer level of this search holds
                for (int i = 0; i < n; i++) {
                    sy.ent_g = sy.group; sy.ent_i = i;
                    if (c.busy[i]) continue;
                    c.busy[i] = 1;
                    if (brings < 0) search(depth + 1);
                    else if (c.op_obj[i] >= 0 && bind(symbols[brings], c.op_obj[i])) search(depth + 1);
                    c.busy[i] = 0;
                }
            } else if (sy.index == -2) {   // 'all': nothing to bind; an object is inferred from the FIRST agent
                if (brings < 0) search(depth + 1);
                else if (n > 0 && c.op_obj[0] >= 0 && bind(symbols[brings], c.op_obj[0])) search(depth + 1);
            } else if (sy.index < n) {     // a fixed agent: the reference only goes on when it can infer an object (RewardEngine.cc:426-438)
                sy.ent_g = sy.group; sy.ent_i = sy.index;
                if (brings >= 0 && c.op_obj[sy.index] >= 0 && bind(symbols[brings], c.op_obj[sy.index])) search(depth + 1);
            }
        };
        search(0);
    }
    for (int g = 0; g < NG; g++)
        if (C[g].dirty) HIP_OK(hipMemcpy(groups[g].cur.next_reward, C[g].reward.data(), sizeof(float) * groups[g].n, hipMemcpyHostToDevice));
}

// ------------------------------------------------------------------------------------------------ device buffers
template <class T>
static void grow(DevArena &arena, T *&p, size_t &cap, size_t need, hipStream_t stream, bool keep = false, size_t keep_n = 0) {
    if (need <= cap) return;
    size_t ncap = std::max(need, cap * 2);
    T *q = nullptr;
    HIP_OK(dev_malloc(arena, &q, sizeof(T) * ncap));
    if (p) {
        HIP_OK(hipStreamSynchronize(stream));
        if (keep && keep_n) HIP_OK(hipMemcpy(q, p, sizeof(T) * keep_n, hipMemcpyDeviceToDevice));
        dev_free(arena, p);
    }
    p = q; cap = ncap;
}

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

template <class T>
static void regrow(DevArena &arena, T *&p, size_t old_n, size_t ncap) {
    T *q = nullptr;
    HIP_OK(dev_malloc(arena, &q, sizeof(T) * ncap));
    if (p && old_n) HIP_OK(hipMemcpy(q, p, sizeof(T) * old_n, hipMemcpyDeviceToDevice));
    if (p) dev_free(arena, p);
    p = q;
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
    // a fresh episode starts with two pairs of optimistic attack rounds: the first steps of a dense placement hold the deepest
    // dependency chains (measured at 2 x 400k: one pair runs out once in the first few steps, two never did), and a step that runs
    // out costs a host round trip; the budget falls back to one pair after 64 steps that did not need the second
    boost_attack = 64;
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
        if (n > 0) map_scattered = true;      // agents that stand next to each other in the group stand anywhere on the map (observe_device)
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
            if (phase == 1) boost_attack = 64; else boost_move = 64;   // deeper dependency chains around: one more batch
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
