// tune.h -- ONE environment variable for everything that only tests and tuning runs ever set:
//     MAGENT_TUNE="key=value,key=value"
// The product reads its settings from the game configuration; these keys force a driver or a kernel that the defaults would not take,
// so that the test suites can run every path on every scenario (tests/test_gpu_fullsize.py: VARIANTS, tests/test_emu_parity.py).
// Unknown keys abort (a typo must not silently test the default path).
//   checked_step=1      the host-checked step driver (one convergence read per pair of rounds) instead of the single-sync driver
//   host_shuffle=1      the reference's literal Fisher-Yates loop on the host instead of the device replay (A/B of the shuffle)
//   attack_pairs=N      optimistic pairs of attack rounds (0: none -- every step continues on the host)      default 1, 2 after a run-out
//   move_batches=N      optimistic batches of generic move sweeps (0: none)                                   default 1, 2 after a run-out
//   solo_step=0         worlds that would step in ONE launch (k_step_solo) take the multi-launch drivers
//   solo_max=N          most agents a world may have to step in one launch                                   default 1536
//   batch_solo_max=N    ... when it is one of a batch (env_cycle_many)                                        default 16384
//   scan_solo_max=N     groups up to N agents compact in one workgroup (clear_dead)                          default 1024
//   overlap=1..3        set_action / the shuffle / the attack rounds on a second stream beside the renders   default 0 (measured: DESIGN 3.5)
//   fold_minimap=0      the next minimap by k_minimap instead of clear_dead's own launches
//   render=0|1|4        force k_render | k_render_fast | k_render_sweep2;  render_sweep=N workgroups, render_su=1..3 strips, render_depth=1..3
//   att_threads=64|128|256   workgroup size of k_attack_eval
//   policy_grid=N       workgroups of k_dqn_conv (tests: a few workgroups walk many tiles);  policy_stamps=1: per-phase cycle stamps
//   early_report=0      the plain pipeline's step report behind the moves instead of ahead of them (A/B of the early `done`)
//   batch_cycle=0       env_cycle_many runs its environments one after another instead of in one pair of launches
//   batch_pipe=0        env_cycle_many never takes the batched pipeline (pipe.hip): worlds beyond the one-launch step go one by one
//   batch_pipe_min=N    ... and takes it for worlds of N agents or more that could also step in one launch                 default 1537
//   pipe_sweep=N        the batched render of env_cycle_many's pipeline: N sweeping workgroups per (environment, group) segment; 0: the generic workgroups   default: ~256 sweeping workgroups over the launch when every observed group has the battle shape
//   pipe_own=N          window cells of a group (x 65536) from which a world of a batch renders by launches of its own instead of the batch's                          default 48
//   pipe_span=N         64-cell steps per workgroup of a batched render (env_cycle_many)                                          default: by size, <= 32
//   touch_map=0|1       never / before every render of a map beyond the L2s: the painted map streamed through the caches first (default: before the
//                       first render of a cycle when some group was placed at random)
#pragma once
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace magent_amd {

// The MAGENT_* variables of rounds 1-3 are gone; a script that still sets one would otherwise run the default path and believe it had not
// (ADVICE round 4).  Checked once, at the first use of any knob: a removed name aborts with its MAGENT_TUNE equivalent.
inline void tune_refuse_removed_variables() {
    static const char *const removed[][2] = {
        {"MAGENT_CHECKED_STEP", "checked_step"}, {"MAGENT_HOST_SHUFFLE", "host_shuffle"}, {"MAGENT_OPT_ATTACK_PAIRS", "attack_pairs"},
        {"MAGENT_OPT_MOVE_BATCHES", "move_batches"}, {"MAGENT_SOLO_STEP", "solo_step"}, {"MAGENT_SOLO_MAX", "solo_max"},
        {"MAGENT_SCAN_SOLO_MAX", "scan_solo_max"}, {"MAGENT_OVERLAP", "overlap"}, {"MAGENT_FOLD_MINIMAP", "fold_minimap"},
        {"MAGENT_RENDER_FAST", "render"}, {"MAGENT_RENDER_SWEEP", "render_sweep"}, {"MAGENT_RENDER_SU", "render_su"},
        {"MAGENT_RENDER_DEPTH", "render_depth"}, {"MAGENT_ATT_THREADS", "att_threads"}, {"MAGENT_POLICY_GRID", "policy_grid"},
        {"MAGENT_BATCH_CYCLE", "batch_cycle"}, {"MAGENT_HIP_POLICY", nullptr}, {"MAGENT_FEAT_SEPARATE", nullptr}, {"MAGENT_RENDER_NT", nullptr},
        {"MAGENT_RENDER_SPAN", nullptr}, {"MAGENT_RENDER_UNROLL", nullptr}, {"MAGENT_RENDER_PAD", nullptr}, {"MAGENT_RENDER_XCD", nullptr}};
    for (const auto &r : removed)
        if (std::getenv(r[0])) {
            if (r[1]) std::fprintf(stderr, "magent-amd FATAL: the environment variable %s is no longer read: use MAGENT_TUNE=%s=... (magent_amd/csrc/tune.h)\n", r[0], r[1]);
            else std::fprintf(stderr, "magent-amd FATAL: the environment variable %s is no longer read and has no successor: the knob was removed (magent_amd/csrc/tune.h)\n", r[0]);
            std::abort();
        }
}

inline int tune(const char *key, int dflt) {
    static const bool legacy_checked = (tune_refuse_removed_variables(), true);
    (void)legacy_checked;
    static const char *const known[] = {"checked_step", "host_shuffle", "attack_pairs", "move_batches", "solo_step", "solo_max", "batch_solo_max", "scan_solo_max", "overlap",
                                        "fold_minimap", "render", "render_sweep", "render_su", "render_depth", "att_threads", "policy_grid", "policy_stamps",
                                        "batch_cycle", "early_report", "touch_map", "batch_pipe", "batch_pipe_min", "pipe_sweep", "pipe_span", "pipe_own"};
    const char *s = std::getenv("MAGENT_TUNE");
    if (!s || !*s) return dflt;
    static bool checked = false;
    if (!checked) {
        checked = true;
        for (const char *p = s; *p;) {
            const char *e = std::strchr(p, ',');
            const size_t len = e ? (size_t)(e - p) : std::strlen(p);
            const char *eq = (const char *)std::memchr(p, '=', len);
            bool ok = false;
            if (eq)
                for (const char *k : known) ok |= std::strlen(k) == (size_t)(eq - p) && !std::strncmp(k, p, (size_t)(eq - p));
            if (!ok && len) { std::fprintf(stderr, "magent-amd FATAL: MAGENT_TUNE: unknown entry \"%.*s\" (see magent_amd/csrc/tune.h)\n", (int)len, p); std::abort(); }
            if (!e) break;
            p = e + 1;
        }
    }
    const size_t n = std::strlen(key);
    for (const char *p = s; *p;) {
        const char *e = std::strchr(p, ',');
        const size_t len = e ? (size_t)(e - p) : std::strlen(p);
        if (len > n && !std::strncmp(p, key, n) && p[n] == '=') return std::atoi(p + n + 1);
        if (!e) break;
        p = e + 1;
    }
    return dflt;
}
inline bool tune_set(const char *key) { return tune(key, -0x7FFFFFFF) != -0x7FFFFFFF; }

}  // namespace magent_amd
