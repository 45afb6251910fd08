// engine_impl.h -- what the translation units of the host engine share (engine.hip, engine_rules.hip, engine_step.hip, engine_batch.hip,
// engine_observe.hip): the error macro, the arena-aware allocation helpers, the profiling scope.  Not part of any interface.
#pragma once
#include <algorithm>
#include <atomic>
#include <chrono>
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

#define HIP_OK(expr)                                                                                   \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) fatal("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

// (every allocation names the arena of the environment it belongs to: an environment only ever frees what it allocated)
template <class T>
inline hipError_t dev_malloc(DevArena &arena, T **p, size_t bytes) {
    if (bytes <= DevArena::SMALL) { *p = (T *)arena.take(bytes ? bytes : 1); return hipSuccess; }
    return hipMalloc(p, bytes);
}
inline void dev_free(DevArena &arena, void *p) {
    if (arena.owns(p)) return;      // (arena memory goes back with the environment)
    (void)hipFree(p);
}

template <class T>
inline void dfree(DevArena &arena, T *&p) {
    if (p) { dev_free(arena, p); p = nullptr; }
}

template <class T>
inline void grow(DevArena &arena, T *&p, size_t &cap, size_t need, hipStream_t stream, bool keep = false, size_t keep_n = 0) {
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

template <class T>
inline void regrow(DevArena &arena, T *&p, size_t old_n, size_t ncap) {
    T *q = nullptr;
    HIP_OK(dev_malloc(arena, &q, sizeof(T) * ncap));
    if (p && old_n) HIP_OK(hipMemcpy(q, p, sizeof(T) * old_n, hipMemcpyDeviceToDevice));
    if (p) dev_free(arena, p);
    p = q;
}

// ------------------------------------------------------------------------------------------------ profiling
struct Env::ProfScope {
    Env &e; Env::ProfSlot *slot = nullptr; hipEvent_t a{}, b{}; hipStream_t s{};
    inline ProfScope(Env &env, const char *name, bool dominant = false, hipStream_t on = nullptr) : e(env), s(on ? on : env.stream) {
        if (!e.prof_level || (e.prof_level == 2 && !dominant)) return;   // an event pair costs ~10 us of stream time
        slot = &e.prof[name];
        a = e.prof_event(); b = e.prof_event();
        HIP_OK(hipEventRecord(a, s));
    }
    inline ~ProfScope() {
        if (!slot) return;
        HIP_OK(hipEventRecord(b, s));
        slot->pending.emplace_back(a, b);
    }
};


}  // namespace magent_amd
