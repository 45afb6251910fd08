// runtime_api.hip -- the C-ABI of libmagent.so (include/magent_runtime_api.h): thin trampolines onto Env.
// Replaces reference src/runtime_api.cc:15-163 symbol for symbol; PART 2 adds the device-resident calls.
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

#include "../../include/magent_runtime_api.h"
#include "engine_host.h"

using magent_amd::Env;
using magent_amd::fatal;

static inline Env *E(EnvHandle h) {
    if (!h) fatal("null environment handle");
    return (Env *)h;
}

// Worker threads of env_cycle_many: started once, parked on a condition variable between rounds (creating threads
// per call cost more than a small world's whole step).  One round at a time (rounds are serialised by `round_mutex`).
namespace {
class CyclePool {
public:
    void run(int n_threads, int n_items, const std::function<void(int)> &fn) {
        std::lock_guard<std::mutex> round(round_mutex);
        {
            std::unique_lock<std::mutex> lk(m);
            while ((int)workers.size() < n_threads - 1) workers.emplace_back([this] { loop(); });
            job = &fn; total = n_items; next = 0; pending = std::min(n_threads - 1, (int)workers.size()); active = pending; epoch++;
        }
        cv.notify_all();
        for (int e; (e = next.fetch_add(1)) < n_items;) fn(e);     // the calling thread works too
        std::unique_lock<std::mutex> lk(m);
        done_cv.wait(lk, [this] { return pending == 0; });
        job = nullptr;
    }
    ~CyclePool() {
        { std::unique_lock<std::mutex> lk(m); quit = true; }
        cv.notify_all();
        for (auto &t : workers) t.join();
    }
private:
    void loop() {
        unsigned seen = 0;
        std::unique_lock<std::mutex> lk(m);
        while (true) {
            cv.wait(lk, [&] { return quit || (epoch != seen && active > 0); });
            if (quit) return;
            seen = epoch; active--;
            const std::function<void(int)> *fn = job;
            const int n = total;
            lk.unlock();
            for (int e; (e = next.fetch_add(1)) < n;) (*fn)(e);
            lk.lock();
            if (--pending == 0) done_cv.notify_one();
        }
    }
    std::mutex m, round_mutex;
    std::condition_variable cv, done_cv;
    std::vector<std::thread> workers;
    const std::function<void(int)> *job = nullptr;
    std::atomic<int> next{0};
    int total = 0, pending = 0, active = 0;
    unsigned epoch = 0;
    bool quit = false;
};
CyclePool &cycle_pool() { static CyclePool *p = new CyclePool(); return *p; }   // never destroyed: no join at process exit
}  // namespace

extern "C" {

int env_new_game(EnvHandle *game, const char *name) {
    if (std::strcmp(name, "GridWorld") != 0)   // reference throws std::invalid_argument (runtime_api.cc:23)
        fatal("invalid name of game '%s': this engine provides GridWorld only (DiscreteSnake is a different game)", name);
    *game = new Env();
    return 0;
}
int env_delete_game(EnvHandle game) { delete E(game); return 0; }
int env_config_game(EnvHandle game, const char *name, void *p_value) { E(game)->set_config(name, p_value); return 0; }
int env_reset(EnvHandle game) { E(game)->reset(); return 0; }
int env_get_observation(EnvHandle game, GroupHandle group, float **buffer) { E(game)->observe_host(group, buffer[0], buffer[1]); return 0; }
int env_set_action(EnvHandle game, GroupHandle group, const int *actions) { E(game)->set_action_host(group, actions); return 0; }
int env_step(EnvHandle game, int *done) { E(game)->step(done); return 0; }
int env_get_reward(EnvHandle game, GroupHandle group, float *buffer) { E(game)->get_reward_host(group, buffer); return 0; }
int env_get_info(EnvHandle game, GroupHandle group, const char *name, void *buffer) { E(game)->info_host(group, name, buffer); return 0; }
int env_render(EnvHandle game) { E(game)->render(); return 0; }   // text video dump (RenderGenerator): host-side, off the hot path
int env_render_next_file(EnvHandle) { return 0; }                 // the reference casts the handle to DiscreteSnake here (runtime_api.cc:85-88)

int gridworld_register_agent_type(EnvHandle game, const char *name, int n, const char **keys, float *values) {
    E(game)->register_agent_type(name, n, keys, values); return 0;
}
int gridworld_new_group(EnvHandle game, const char *agent_type_name, GroupHandle *group) { E(game)->new_group(agent_type_name, group); return 0; }
int gridworld_add_agents(EnvHandle game, GroupHandle group, int n, const char *method, const int *pos_x, const int *pos_y, const int *dir) {
    E(game)->add_agents(group, n, method, pos_x, pos_y, dir); return 0;
}
int gridworld_clear_dead(EnvHandle game) { E(game)->clear_dead(); return 0; }
int gridworld_set_goal(EnvHandle game, GroupHandle group, const char *method, const int *) { E(game)->set_goal(group, method); return 0; }
int gridworld_define_agent_symbol(EnvHandle game, int no, int group, int index) { E(game)->define_agent_symbol(no, group, index); return 0; }
int gridworld_define_event_node(EnvHandle game, int no, int op, int *inputs, int n_inputs) { E(game)->define_event_node(no, op, inputs, n_inputs); return 0; }
int gridworld_add_reward_rule(EnvHandle game, int on, int *receiver, float *value, int n_receiver, bool is_terminal, bool /*auto_value*/) {
    E(game)->add_reward_rule(on, receiver, value, n_receiver, is_terminal); return 0;
}
int discrete_snake_clear_dead(EnvHandle) { fatal("DiscreteSnake is a different game; not provided by this engine"); }
int discrete_snake_add_object(EnvHandle, int, int, const char *, const int *) { fatal("DiscreteSnake is a different game; not provided by this engine"); }

// ---- PART 2: device-resident extensions
int env_get_observation_device(EnvHandle game, GroupHandle group, float **device_buffer) {
    E(game)->observe_device(group, device_buffer[0], device_buffer[1]); return 0;
}
int env_get_observation_device_bf16(EnvHandle game, GroupHandle group, void **device_buffer) {
    E(game)->observe_device(group, (float *)device_buffer[0], (float *)device_buffer[1], true); return 0;
}
int env_set_action_device(EnvHandle game, GroupHandle group, const int *device_actions) { E(game)->set_action_device(group, device_actions); return 0; }
int env_get_reward_device(EnvHandle game, GroupHandle group, float *device_buffer) { E(game)->get_reward_device(group, device_buffer); return 0; }
int env_get_info_device(EnvHandle game, GroupHandle group, const char *name, void *device_buffer) { E(game)->info_device(group, name, device_buffer); return 0; }
// step several environments with overlapped device work: every step is enqueued on its environment's own stream
// before the first one is waited for (small worlds are launch-latency bound; concurrent environments fill the gaps)
int env_step_many(EnvHandle *games, int n, int *done) {
    for (int i = 0; i < n; i++) E(games[i])->step_begin();
    for (int i = 0; i < n; i++) E(games[i])->step_end(&done[i]);
    return 0;
}
// One full environment cycle -- for each group: observe into device buffers, set device actions; step; rewards;
// clear_dead (the loop body of examples/train_battle.py:61-109) -- for n_env independent environments, driven by
// n_threads host threads inside the library (no Python GIL, one HIP stream per environment).  Arrays are indexed
// [e * n_group + g]; a NULL view / actions / rewards entry skips that call for that group.
int env_cycle_many(EnvHandle *games, int n_env, int n_group, float **view, float **feat, const int **actions,
                   float **rewards, int *done, int n_threads) {
    auto one = [&](int e) {
        const int o = e * n_group;
        E(games[e])->cycle(n_group, view ? view + o : nullptr, feat ? feat + o : nullptr, actions ? actions + o : nullptr,
                           rewards ? rewards + o : nullptr, &done[e]);
    };
    // several environments: one pair of launches for all that are small enough (MAGENT_TUNE batch_cycle=0: one by one, by threads)
    static const bool batch = magent_amd::tune("batch_cycle", 1) != 0;
    if (n_env >= 2 && batch) {
        std::vector<Env *> envs(n_env);
        for (int e = 0; e < n_env; e++) envs[e] = E(games[e]);
        Env::cycle_many(envs.data(), n_env, n_group, view, feat, actions, rewards, done, [&](const std::vector<int> &alone) {
            // environments outside the batch: ordinary cycles on their own streams, spread over the host threads
            if (n_threads <= 1 || alone.size() <= 1) { for (int e : alone) one(e); return; }
            cycle_pool().run(n_threads < (int)alone.size() ? n_threads : (int)alone.size(), (int)alone.size(), [&](int k) { one(alone[k]); });
        });
        return 0;
    }
    if (n_threads <= 1 || n_env <= 1) { for (int e = 0; e < n_env; e++) one(e); return 0; }
    cycle_pool().run(n_threads < n_env ? n_threads : n_env, n_env, one);
    return 0;
}
// agent counts of n_env environments x n_group groups in one call (the host mirror: no device work)
int env_num_many(EnvHandle *games, int n_env, int n_group, int *out) {
    for (int e = 0; e < n_env; e++) {
        Env *env = E(games[e]);
        for (int g = 0; g < n_group; g++) out[e * n_group + g] = env->group_count(g);
    }
    return 0;
}
// the streams of n_env environments in one call: out[2 e] = env_get_stream's, out[2 e + 1] = env_get_action_stream's
int env_streams_many(EnvHandle *games, int n_env, void **out) {
    for (int e = 0; e < n_env; e++) { out[2 * e] = (void *)E(games[e])->stream; out[2 * e + 1] = (void *)E(games[e])->action_stream(); }
    return 0;
}
int env_sync(EnvHandle game) { E(game)->sync(); return 0; }
int env_get_stream(EnvHandle game, void **stream) { *stream = (void *)E(game)->stream; return 0; }
int env_get_action_stream(EnvHandle game, void **stream) { *stream = (void *)E(game)->action_stream(); return 0; }
int env_profile_enable(EnvHandle game, int on) { E(game)->prof_level = on < 0 ? 0 : on > 2 ? 1 : on; return 0; }
int env_profile_read(EnvHandle game, const char *name, int *n_launches, float *total_ms) { E(game)->profile_read(name, n_launches, total_ms); return 0; }

}  // extern "C"
