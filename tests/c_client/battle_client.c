/* A plain-C client of the C-ABI (include/magent_runtime_api.h): no Python, no torch, only pointers and ints.
 * Builds the battle game by hand (the calls python/magent/gridworld.py:41-115 makes), plays a scripted episode with a
 * tiny LCG as the action source and prints an FNV-1a checksum of every output buffer after every step.
 * tests/test_c_client.py runs it against the HIP library (GPU) and against the CPU oracle and compares the lines. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "magent_runtime_api.h"

static uint64_t fnv(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; }
    return h;
}

int main(int argc, char **argv) {
    int map_size = argc > 1 ? atoi(argv[1]) : 40, n_agents = argc > 2 ? atoi(argv[2]) : 300, steps = argc > 3 ? atoi(argv[3]) : 8;
    EnvHandle game;
    env_new_game(&game, "GridWorld");
    int w = map_size, emb = 10, seed = 777;
    bool yes = true;
    env_config_game(game, "map_width", &w);
    env_config_game(game, "map_height", &w);
    env_config_game(game, "minimap_mode", &yes);
    env_config_game(game, "embedding_size", &emb);
    const char *keys[] = {"width", "length", "hp", "speed", "view_radius", "view_angle", "attack_radius", "attack_angle",
                          "damage", "step_recover", "step_reward", "kill_reward", "dead_penalty", "attack_penalty"};
    float values[] = {1, 1, 4, 2, 6, 360, 1.5f, 360, 3, 0.1f, -0.005f, 5, -0.1f, -0.1f};
    gridworld_register_agent_type(game, "small", 14, keys, values);
    /* reward rules: Event(g0 attack g1) -> g0 += 0.2 and the mirror image */
    gridworld_define_agent_symbol(game, 0, 0, -1);
    gridworld_define_agent_symbol(game, 1, 1, -1);
    int in01[2] = {0, 1}, in10[2] = {1, 0}, r0[1] = {0}, r1[1] = {1};
    float v[1] = {0.2f};
    gridworld_define_event_node(game, 0, 7, in01, 2);
    gridworld_define_event_node(game, 1, 7, in10, 2);
    gridworld_add_reward_rule(game, 0, r0, v, 1, false, false);
    gridworld_add_reward_rule(game, 1, r1, v, 1, false, false);
    GroupHandle g[2];
    gridworld_new_group(game, "small", &g[0]);
    gridworld_new_group(game, "small", &g[1]);
    env_config_game(game, "seed", &seed);
    env_reset(game);
    for (int k = 0; k < 2; k++) gridworld_add_agents(game, g[k], n_agents, "random", NULL, NULL, NULL);

    int space[3], fsz, n_action;
    env_get_info(game, g[0], "view_space", space);
    env_get_info(game, g[0], "feature_space", &fsz);
    env_get_info(game, g[0], "action_space", &n_action);
    size_t vsz = (size_t)space[0] * space[1] * space[2];
    float *view = malloc(sizeof(float) * vsz * n_agents), *feat = malloc(sizeof(float) * fsz * n_agents);
    float *reward = malloc(sizeof(float) * n_agents);
    int *act = malloc(sizeof(int) * n_agents), *pos = malloc(sizeof(int) * 2 * n_agents);
    bool *alive = malloc(n_agents);
    uint32_t lcg = 12345;
    for (int s = 0; s < steps; s++) {
        uint64_t h = 1469598103934665603ull;
        for (int k = 0; k < 2; k++) {
            int n;
            env_get_info(game, g[k], "num", &n);
            float *bufs[2] = {view, feat};
            env_get_observation(game, g[k], bufs);
            h = fnv(h, view, sizeof(float) * vsz * n);
            h = fnv(h, feat, sizeof(float) * fsz * n);
            for (int i = 0; i < n; i++) { lcg = lcg * 1664525u + 1013904223u; act[i] = (int)((lcg >> 8) % (uint32_t)n_action); }
            env_set_action(game, g[k], act);
        }
        int done;
        env_step(game, &done);
        for (int k = 0; k < 2; k++) {
            int n;
            env_get_info(game, g[k], "num", &n);
            env_get_reward(game, g[k], reward);
            env_get_info(game, g[k], "pos", pos);
            env_get_info(game, g[k], "alive", alive);
            h = fnv(h, reward, sizeof(float) * n);
            h = fnv(h, pos, sizeof(int) * 2 * n);
            h = fnv(h, alive, n);
        }
        gridworld_clear_dead(game);
        int n0, n1;
        env_get_info(game, g[0], "num", &n0);
        env_get_info(game, g[1], "num", &n1);
        printf("step %d done %d num %d %d checksum %016llx\n", s, done, n0, n1, (unsigned long long)h);
    }
    env_delete_game(game);
    return 0;
}
