/* TEST INFRASTRUCTURE.  A libmagent.so that stands between the reference's UNMODIFIED Python wrapper and a real engine library and writes down
 * every C-ABI call the wrapper makes -- the arguments exactly as they cross the boundary (src/runtime_api.h:20-62 of the reference), the bytes of
 * every input buffer and the bytes the engine wrote back.  tests/golden/make_abi_trace.py runs an episode of the reference wrapper through it
 * against the compiled reference (oracle/_ref) in the build container; the transcript is committed, and tests/test_abi_trace.py replays it call by
 * call against the HIP engine on the GPU box (where the reference tree does not exist) and compares every returned byte.
 *
 * Transcript: records of  u32 func | u32 n_fields | fields...;  a field = u32 kind (0 value / input bytes, 1 output bytes) | u32 nbytes | bytes.
 * The replayer knows the same per-function field lists (tests/test_abi_trace.py: SCHEMA). */
#include <dlfcn.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef void *EnvHandle;
typedef int GroupHandle;
static void *target;
static FILE *out;

static void *sym(const char *name) {
    if (!target) {
        const char *path = getenv("MAGENT_TRACE_TARGET");
        target = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!target) { fprintf(stderr, "trace_shim: cannot open %s: %s\n", path ? path : "(MAGENT_TRACE_TARGET unset)", dlerror()); abort(); }
        out = fopen(getenv("MAGENT_TRACE_OUT"), "wb");
        if (!out) { perror("trace_shim: MAGENT_TRACE_OUT"); abort(); }
    }
    void *f = dlsym(target, name);
    if (!f) { fprintf(stderr, "trace_shim: %s missing in the target\n", name); abort(); }
    return f;
}
static void begin(uint32_t func, uint32_t n_fields) { fwrite(&func, 4, 1, out); fwrite(&n_fields, 4, 1, out); }
static void field(uint32_t kind, const void *p, size_t n) { uint32_t k = kind, b = (uint32_t)n; fwrite(&k, 4, 1, out); fwrite(&b, 4, 1, out); if (n) fwrite(p, 1, n, out); }
static void field_int(int v) { field(0, &v, 4); }
static void field_str(const char *s) { field(0, s, strlen(s)); }
static void end(void) { fflush(out); }

enum { F_NEW_GAME, F_DELETE_GAME, F_CONFIG_GAME, F_RESET, F_GET_OBSERVATION, F_SET_ACTION, F_STEP, F_GET_REWARD, F_GET_INFO, F_RENDER, F_RENDER_NEXT_FILE,
       F_REGISTER_AGENT_TYPE, F_NEW_GROUP, F_ADD_AGENTS, F_CLEAR_DEAD, F_SET_GOAL, F_DEFINE_AGENT_SYMBOL, F_DEFINE_EVENT_NODE, F_ADD_REWARD_RULE };

/* sizes the wrapper's buffers are made for (the shim asks the engine, the way the wrapper did before it allocated them) */
static int info_int(EnvHandle g, GroupHandle h, const char *name, int k) {
    int buf[8] = {0};
    ((int (*)(EnvHandle, GroupHandle, const char *, void *))sym("env_get_info"))(g, h, name, buf);
    return buf[k];
}
static int num(EnvHandle g, GroupHandle h) { return info_int(g, h, "num", 0); }

int env_new_game(EnvHandle *game, const char *name) {
    int r = ((int (*)(EnvHandle *, const char *))sym("env_new_game"))(game, name);
    begin(F_NEW_GAME, 1); field_str(name); end();
    return r;
}
int env_delete_game(EnvHandle game) { begin(F_DELETE_GAME, 0); end(); return ((int (*)(EnvHandle))sym("env_delete_game"))(game); }
int env_config_game(EnvHandle game, const char *name, void *p_value) {
    /* value widths by key as the wrapper passes them (gridworld.py:52-63): bool 1 byte, int / float 4 bytes, render_dir a C string */
    size_t n = 4;
    if (!strcmp(name, "render_dir")) n = strlen((const char *)p_value);
    else if (!strcmp(name, "turn_mode") || !strcmp(name, "minimap_mode") || !strcmp(name, "goal_mode") || !strcmp(name, "revive_mode") ||
             !strcmp(name, "food_mode")) n = 1;
    begin(F_CONFIG_GAME, 2); field_str(name); field(0, p_value, n); end();
    return ((int (*)(EnvHandle, const char *, void *))sym("env_config_game"))(game, name, p_value);
}
int env_reset(EnvHandle game) { begin(F_RESET, 0); end(); return ((int (*)(EnvHandle))sym("env_reset"))(game); }
int env_get_observation(EnvHandle game, GroupHandle group, float **buffer) {
    int n = num(game, group), vs[3], fs;
    vs[0] = info_int(game, group, "view_space", 0); vs[1] = info_int(game, group, "view_space", 1); vs[2] = info_int(game, group, "view_space", 2);
    fs = info_int(game, group, "feature_space", 0);
    int r = ((int (*)(EnvHandle, GroupHandle, float **))sym("env_get_observation"))(game, group, buffer);
    begin(F_GET_OBSERVATION, 3); field_int(group);
    field(1, buffer[0], (size_t)n * vs[0] * vs[1] * vs[2] * 4); field(1, buffer[1], (size_t)n * fs * 4); end();
    return r;
}
int env_set_action(EnvHandle game, GroupHandle group, const int *actions) {
    begin(F_SET_ACTION, 2); field_int(group); field(0, actions, (size_t)num(game, group) * 4); end();
    return ((int (*)(EnvHandle, GroupHandle, const int *))sym("env_set_action"))(game, group, actions);
}
int env_step(EnvHandle game, int *done) {
    int r = ((int (*)(EnvHandle, int *))sym("env_step"))(game, done);
    begin(F_STEP, 1); field(1, done, 4); end();
    return r;
}
int env_get_reward(EnvHandle game, GroupHandle group, float *buffer) {
    int n = num(game, group);
    int r = ((int (*)(EnvHandle, GroupHandle, float *))sym("env_get_reward"))(game, group, buffer);
    begin(F_GET_REWARD, 2); field_int(group); field(1, buffer, (size_t)n * 4); end();
    return r;
}
int env_get_info(EnvHandle game, GroupHandle group, const char *name, void *buffer) {
    size_t n_out = 0;
    if (!strcmp(name, "num") || !strcmp(name, "feature_space") || !strcmp(name, "action_space") || !strcmp(name, "attack_base")) n_out = 4;
    else if (!strcmp(name, "view_space")) n_out = 12;
    else if (!strcmp(name, "id")) n_out = (size_t)num(game, group) * 4;
    else if (!strcmp(name, "pos")) n_out = (size_t)num(game, group) * 8;
    else if (!strcmp(name, "alive")) n_out = (size_t)num(game, group);
    else if (!strcmp(name, "view2attack")) n_out = (size_t)info_int(game, group, "view_space", 0) * info_int(game, group, "view_space", 1) * 4;
    else { fprintf(stderr, "trace_shim: env_get_info(%s) is not in the traced set\n", name); abort(); }
    int r = ((int (*)(EnvHandle, GroupHandle, const char *, void *))sym("env_get_info"))(game, group, name, buffer);
    begin(F_GET_INFO, 3); field_int(group); field_str(name); field(1, buffer, n_out); end();
    return r;
}
int env_render(EnvHandle game) { begin(F_RENDER, 0); end(); return ((int (*)(EnvHandle))sym("env_render"))(game); }
int env_render_next_file(EnvHandle game) { begin(F_RENDER_NEXT_FILE, 0); end(); return ((int (*)(EnvHandle))sym("env_render_next_file"))(game); }
int gridworld_register_agent_type(EnvHandle game, const char *name, int n, const char **keys, float *values) {
    begin(F_REGISTER_AGENT_TYPE, 2 + 2 * (uint32_t)n); field_str(name); field_int(n);
    for (int k = 0; k < n; k++) { field_str(keys[k]); field(0, &values[k], 4); }
    end();
    return ((int (*)(EnvHandle, const char *, int, const char **, float *))sym("gridworld_register_agent_type"))(game, name, n, keys, values);
}
int gridworld_new_group(EnvHandle game, const char *agent_type_name, GroupHandle *group) {
    int r = ((int (*)(EnvHandle, const char *, GroupHandle *))sym("gridworld_new_group"))(game, agent_type_name, group);
    begin(F_NEW_GROUP, 2); field_str(agent_type_name); field(1, group, 4); end();
    return r;
}
int gridworld_add_agents(EnvHandle game, GroupHandle group, int n, const char *method, const int *pos_x, const int *pos_y, const int *dir) {
    begin(F_ADD_AGENTS, 6); field_int(group); field_int(n); field_str(method);
    if (!strcmp(method, "custom")) { field(0, pos_x, (size_t)n * 4); field(0, pos_y, (size_t)n * 4); field(0, dir, (size_t)n * 4); }
    else if (!strcmp(method, "fill") || !strcmp(method, "maze")) { field(0, pos_x, 20); field(0, NULL, 0); field(0, NULL, 0); }   /* five ints behind pos_x (gridworld.py:185-190) */
    else { field(0, NULL, 0); field(0, NULL, 0); field(0, NULL, 0); }                                                             /* random: the wrapper passes three zeros */
    end();
    return ((int (*)(EnvHandle, GroupHandle, int, const char *, const int *, const int *, const int *))sym("gridworld_add_agents"))(game, group, n, method, pos_x, pos_y, dir);
}
int gridworld_clear_dead(EnvHandle game) { begin(F_CLEAR_DEAD, 0); end(); return ((int (*)(EnvHandle))sym("gridworld_clear_dead"))(game); }
int gridworld_set_goal(EnvHandle game, GroupHandle group, const char *method, const int *linear_buffer, int n) {
    begin(F_SET_GOAL, 0); end();
    return ((int (*)(EnvHandle, GroupHandle, const char *, const int *, int))sym("gridworld_set_goal"))(game, group, method, linear_buffer, n);
}
int gridworld_define_agent_symbol(EnvHandle game, int no, int group, int index) {
    begin(F_DEFINE_AGENT_SYMBOL, 3); field_int(no); field_int(group); field_int(index); end();
    return ((int (*)(EnvHandle, int, int, int))sym("gridworld_define_agent_symbol"))(game, no, group, index);
}
int gridworld_define_event_node(EnvHandle game, int no, int op, int *inputs, int n_inputs) {
    begin(F_DEFINE_EVENT_NODE, 3); field_int(no); field_int(op); field(0, inputs, (size_t)n_inputs * 4); end();
    return ((int (*)(EnvHandle, int, int, int *, int))sym("gridworld_define_event_node"))(game, no, op, inputs, n_inputs);
}
/* (the wrapper passes SIX of the seven arguments, gridworld.py:564-565: auto_value is whatever the register holds -- not recorded, not forwarded) */
int gridworld_add_reward_rule(EnvHandle game, int on, int *receiver, float *value, int n_receiver, unsigned char is_terminal, unsigned char auto_value) {
    (void)auto_value;
    int term = is_terminal ? 1 : 0;
    begin(F_ADD_REWARD_RULE, 4); field_int(on); field(0, receiver, (size_t)n_receiver * 4); field(0, value, (size_t)n_receiver * 4); field_int(term); end();
    return ((int (*)(EnvHandle, int, int *, float *, int, unsigned char, unsigned char))sym("gridworld_add_reward_rule"))(game, on, receiver, value, n_receiver, is_terminal, 0);
}
int discrete_snake_clear_dead(EnvHandle game) { (void)game; return 0; }
int discrete_snake_add_object(EnvHandle game, int a, int b, const char *c, const int *d) { (void)game; (void)a; (void)b; (void)c; (void)d; return 0; }
