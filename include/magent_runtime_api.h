/*
 * magent_runtime_api.h -- C-ABI of the MI355X-native grid-world step engine (libmagent.so).
 *
 * This is the drop-in boundary.  Every symbol in PART 1 replaces, with the same name, argument order,
 * argument meaning and return value (always 0), the function the reference exports from
 * /root/reference/src/runtime_api.h:20-56 and that python/magent/gridworld.py binds through ctypes
 * (python/magent/c_lib.py:11-21 loads "<pkg>/../../build/libmagent.so").  Plain pointers and ints only:
 * no torch / HIP types appear in any signature.
 *
 * Ownership (same as the reference): the caller owns every data buffer and sizes it from
 * env_get_info("num") x the *_space infos; the engine owns the opaque EnvHandle until env_delete_game.
 * PART 1 buffers are HOST memory; each call is synchronous (outputs valid on return).
 * PART 2 is additive: the same operations on DEVICE pointers, asynchronous on the environment's HIP stream.
 *
 * Errors: like the reference (LOG(FATAL) -> std::terminate under ctypes, utility.h:77-80) an invalid
 * argument or an unsupported feature prints "magent-amd FATAL: ..." to stderr and aborts; nothing returns
 * non-zero expecting the caller to look.
 */
#ifndef MAGENT_AMD_RUNTIME_API_H
#define MAGENT_AMD_RUNTIME_API_H

#include <stdbool.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void *EnvHandle;  /* reference: Environment*  (Environment.h:36) */
typedef int GroupHandle;  /* reference: Environment.h:12; -1 = walls / "no group" (GridWorld.cc:184) */

/* ---------------------------------------------------------------------------------------------------
 * PART 1 -- the reference C-ABI (host buffers, synchronous)
 * ------------------------------------------------------------------------------------------------- */

/* runtime_api.h:21  name must be "GridWorld" ("DiscreteSnake" is a different game: FATAL unsupported). */
int env_new_game(EnvHandle *game, const char *name);
/* runtime_api.h:22 */
int env_delete_game(EnvHandle game);
/* runtime_api.h:23 -> GridWorld::set_config (GridWorld.cc:120-149).  keys: map_width,map_height (int*),
 * food_mode,turn_mode,minimap_mode,goal_mode (bool*), embedding_size (int*), render_dir (char*), seed (int*).
 * (turn_mode and food_mode are implemented; so is goal_mode, "deprecated" at GridWorld.cc:137 and set by no shipped game: what it
 * does in the reference is add two feature slots that nothing ever writes, GridWorld.cc:926-934.)
 * Additive key: device_id (int*) selects the HIP device (before env_reset). */
int env_config_game(EnvHandle game, const char *name, void *p_value);

/* runtime_api.h:26 -> GridWorld::reset (GridWorld.cc:72-118) */
int env_reset(EnvHandle game);
/* runtime_api.h:27 -> GridWorld::get_observation (GridWorld.cc:292-401)
 * buffer[0] = view   float[n][view_h][view_w][n_channel], buffer[1] = feature float[n][feature_size] */
int env_get_observation(EnvHandle game, GroupHandle group, float **buffer);
/* runtime_api.h:28 -> GridWorld::set_action (GridWorld.cc:403-454); actions int32[n].
 * A SECOND set_action for the same group before env_step appends, as in the reference: every agent of the group then acts once per
 * call (two entries in the shuffled attack list, two turns / moves in list order), and last_action is the latest call's.  No caller of the
 * reference does that, so it is served off the hot path: the step runs the reference's sequential loops on one lane of the device
 * (exact, about a microsecond per list entry; every game).  The same loops run a step in which a can_absorb group ("goals") was given
 * actions: such goals move like any agent in the reference (Map.cc:313-358), which the parallel move resolution does not cover.
 * An action outside [0, n_action) is reported at env_step (FATAL; the reference indexes out of range). */
int env_set_action(EnvHandle game, GroupHandle group, const int *actions);
/* runtime_api.h:29 -> GridWorld::step (GridWorld.cc:456-631) */
int env_step(EnvHandle game, int *done);
/* runtime_api.h:30 -> GridWorld::get_reward (GridWorld.cc:694-704); buffer float[n] */
int env_get_reward(EnvHandle game, GroupHandle group, float *buffer);

/* runtime_api.h:33 -> GridWorld::get_info (GridWorld.cc:709-894).  names: num,id,pos,alive,global_minimap,
 * walls_info,render_window_info,attack_event,action_space,view_space,feature_space,view2attack,attack_base,
 * groups_info,both_attack,mean_info ("deprecated" in the reference, GridWorld.cc:765-786: float[2 + n_action] = mean x,
 * mean y, share of every action; served since round 5 -- only ask for groups that have been given actions: the reference
 * counts an agent that never acted one past the end of a heap array).
 * Additive names (tuning / tests): engine_stats -> int32[8], round_hist -> int32[9] (magent_amd/gridworld.py). */
int env_get_info(EnvHandle game, GroupHandle group, const char *name, void *buffer);

/* runtime_api.h:36-37 -> RenderGenerator (text video dump; host-side, off the hot path) */
int env_render(EnvHandle game);
int env_render_next_file(EnvHandle game);

/* runtime_api.h:43 -> AgentType::AgentType (AgentType.cc:30-123) */
int gridworld_register_agent_type(EnvHandle game, const char *name, int n, const char **keys, float *values);
/* runtime_api.h:44 -> GridWorld::new_group (GridWorld.cc:160-169) */
int gridworld_new_group(EnvHandle game, const char *agent_type_name, GroupHandle *group);
/* runtime_api.h:45-46 -> GridWorld::add_agents (GridWorld.cc:180-290); method = random|custom|fill */
int gridworld_add_agents(EnvHandle game, GroupHandle group, int n, const char *method,
                         const int *pos_x, const int *pos_y, const int *dir);

/* runtime_api.h:49 -> GridWorld::clear_dead (GridWorld.cc:633-665) */
int gridworld_clear_dead(EnvHandle game);
/* runtime_api.h:50 -> GridWorld::set_goal (GridWorld.cc:667-679; deprecated in the reference).  method "random" draws a goal position
 * for every agent of the group from the engine's generator (two draws each, the uncleared dead included); nothing in the reference
 * reads a goal back, so the call's whole effect is on the generator -- on every shuffle and random placement after it -- with or
 * without goal_mode.  Any other method is FATAL, as there.  linear_buffer is unused, as there. */
int gridworld_set_goal(EnvHandle game, GroupHandle group, const char *method, const int *linear_buffer);

/* runtime_api.h:53-56 -> RewardEngine.cc:28-69 */
int gridworld_define_agent_symbol(EnvHandle game, int no, int group, int index);
int gridworld_define_event_node(EnvHandle game, int no, int op, int *inputs, int n_inputs);
int gridworld_add_reward_rule(EnvHandle game, int on, int *receiver, float *value, int n_receiver,
                              bool is_terminal, bool auto_value);

/* runtime_api.h:61-62: the second game of the reference; exported so that the symbol table matches,
 * FATAL "unsupported" when called. */
int discrete_snake_clear_dead(EnvHandle game);
int discrete_snake_add_object(EnvHandle game, int obj_id, int n, const char *method, const int *linear_buffer);

/* ---------------------------------------------------------------------------------------------------
 * PART 2 -- additive MI355X extensions (device-resident buffers; SURVEY.md 8b "extensions allowed")
 * All pointers below are DEVICE pointers on the environment's device.  Calls enqueue work on the
 * environment's HIP stream and return without waiting; env_sync() waits for the stream.
 * ------------------------------------------------------------------------------------------------- */

/* same layout as env_get_observation, written straight into caller-owned device memory */
int env_get_observation_device(EnvHandle game, GroupHandle group, float **device_buffer);
/* the same observation in the policy kernels' input format (include/magent_policy.h): device_buffer[0] = view as bf16
 * [n][view_h][view_w][8] -- the n_channel (<= 7) channels of env_get_observation rounded to nearest even, zeros, and 1.0 in
 * channel 7; 16-byte aligned -- device_buffer[1] = feature float[n][feature_size] as above.  2.7 KB per agent instead of 4.7:
 * the render is HBM-write bound, and a policy that computes in bf16 reads these cells as its MFMA operands. */
int env_get_observation_device_bf16(EnvHandle game, GroupHandle group, void **device_buffer);
/* actions int32[n] in device memory */
int env_set_action_device(EnvHandle game, GroupHandle group, const int *device_actions);
/* rewards float[n] in device memory */
int env_get_reward_device(EnvHandle game, GroupHandle group, float *device_buffer);
/* name = id (int32[n]) | pos (int32[n][2]) | alive (uint8[n]) | hp (float[n]) into device memory */
int env_get_info_device(EnvHandle game, GroupHandle group, const char *name, void *device_buffer);
/* env_step for n independent environments at once: all steps are enqueued (each on its environment's stream) before
 * the first is waited for, so their device work overlaps; done[i] as env_step */
int env_step_many(EnvHandle *games, int n, int *done);
/* One full cycle of n_env independent environments, exactly the call sequence
 *   for g: env_get_observation_device(e, g, {view, feat});   for g: env_set_action_device(e, g, actions);
 *   env_step(e, &done[e]);   for g: env_get_reward_device(e, g, rewards);   gridworld_clear_dead(e)
 * Device pointer arrays are indexed [e * n_group + g]; a NULL entry skips that call for that group.
 * Three forms, chosen per environment:
 *   - worlds small enough for the one-launch step -- <= 1536 agents for an environment cycled on its own (n_env == 1), <= 16384 as one
 *     of a batch (n_env >= 2); MAGENT_TUNE solo_max / batch_solo_max -- run the whole cycle in TWO launches (observation render of
 *     all groups; everything else), and with n_env >= 2 ALL such environments share one pair of launches on the first one's stream
 *     (one workgroup per environment for the step): small worlds are launch-latency bound, many of them fill the GPU;
 *   - with n_env >= 2, worlds of >= 1537 agents (MAGENT_TUNE batch_pipe_min) whose game the pipeline of plain games takes (one-cell
 *     bodies, no turn_mode / food_mode / goals / kill_supply, rules that pay the attacker: battle, gather) share ONE chain of launches,
 *     one per phase for all of them, without a host round trip inside the cycle (pipe.hip; MAGENT_TUNE batch_pipe=0 switches it off);
 *   - every other world runs the calls one after the other on n_threads host threads inside the library.
 * `actions` must hold the actions BEFORE the call (they are read by the second launch, in
 * stream order after the render -- the caller's policy reads the observation of the PREVIOUS cycle, or orders its own
 * stream with env_get_stream); the outputs are complete when the call returns (the host has waited for `done`). */
int env_cycle_many(EnvHandle *games, int n_env, int n_group, float **view, float **feat, const int **actions,
                   float **rewards, int *done, int n_threads);
/* out[e * n_group + g] = number of agents of group g in environment e (env_get_info "num" for a whole batch; host only) */
int env_num_many(EnvHandle *games, int n_env, int n_group, int *out);
/* wait until everything enqueued on the environment's stream has finished */
int env_sync(EnvHandle game);
/* the environment's hipStream_t, as an opaque pointer (for event timing / interop by the caller) */
int env_get_stream(EnvHandle game, void **stream);
/* the hipStream_t on which the NEXT env_set_action_device will read its actions.  Worlds too large for the one-launch step
 * run set_action and the read-only head of env_step (attack shuffle, hit gather, death-rank fixed point) on a second
 * stream, beside the observation renders on env_get_stream's stream; a caller that produces actions
 * asynchronously on a stream of its own orders THIS stream behind it.  Equal to env_get_stream's for small worlds or
 * with MAGENT_TUNE overlap=0 (the default).  All outputs (observations, rewards, infos) are ordered on env_get_stream's stream as before. */
int env_get_action_stream(EnvHandle game, void **stream);
/* both streams of n_env environments in one call (out: 2 * n_env pointers, [2 e] = env_get_stream, [2 e + 1] = env_get_action_stream):
 * what a batch of environments needs to order its inputs once per DISTINCT stream -- environments cycled together share one */
int env_streams_many(EnvHandle *games, int n_env, void **out);

/* Kernel timing with HIP events recorded on the environment's stream.
 * env_profile_enable(game, 1) starts recording one event pair per launch of each named kernel; (game, 2) records
 * only the observation render launches ("render", "features") -- an event pair costs ~10 us of stream time, so this
 * is the level to leave on inside a timed region; (game, 0) stops.  env_profile_read returns, for kernel `name` ("render", "paint", "minimap", "attack", "move", ...),
 * the number of recorded launches and their total duration in milliseconds, and resets the counters. */
int env_profile_enable(EnvHandle game, int on);
int env_profile_read(EnvHandle game, const char *name, int *n_launches, float *total_ms);

#ifdef __cplusplus
}
#endif
#endif /* MAGENT_AMD_RUNTIME_API_H */
