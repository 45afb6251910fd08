// hipemu: a <hip/hip_runtime.h> that runs HIP kernels lane by lane on the CPU.  TEST INFRASTRUCTURE ONLY.
//
// The build container has no GPU and a round has ~90 GPU-minutes, so the kernels of magent_amd/csrc are ALSO compiled,
// unchanged, as plain C++ against this header (tests/hipemu/build.py -> tests/hipemu/_build/libmagent_emu.so) and the parity
// suite can be run against that library on the CPU.  It checks the kernels' LOGIC (index arithmetic, fixed points, atomics
// protocols, barrier placement); it says nothing about performance and it does not replace the `-m gpu` parity tests, which
// run the same sources compiled by hipcc on the MI355X.  Nothing outside tests/ loads the emulated library; the product
// library (magent_amd/lib/libmagent.so) is the hipcc build and has no CPU path.
//
// Execution model: one launch = its workgroups one after another (optionally in a scrambled order); the threads of a
// workgroup are fibers on ONE host thread.  A fiber runs until it reaches __syncthreads(), a wave-level operation (__ballot,
// __shfl_down, wave barrier, an MFMA -- the matrix instruction is a wave-wide meeting that gathers every lane's operands, see
// "matrix cores" below) or the end of the kernel; barriers and wave operations complete when every lane that can still
// reach them has arrived.  HIPEMU_SCRAMBLE=<seed> runs the lanes of a workgroup (and the workgroups of a launch) in a
// pseudo-random order that changes at every scheduling pass, which makes results that depend on an unsynchronised order show.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <tuple>
#include <type_traits>
#include <utility>

#define __host__
#define __device__
#define __global__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __constant__ static

// ------------------------------------------------------------------------------------------------ vector types
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
#define HIPEMU_VEC2(N, T) struct alignas(sizeof(T) * 2) N { T x, y; }; static inline N make_##N(T x, T y) { return N{x, y}; }
#define HIPEMU_VEC4(N, T) struct alignas(sizeof(T) * 4) N { T x, y, z, w; }; static inline N make_##N(T x, T y, T z, T w) { return N{x, y, z, w}; }
HIPEMU_VEC2(int2, int) HIPEMU_VEC2(uint2, unsigned) HIPEMU_VEC2(float2, float) HIPEMU_VEC2(short2, short) HIPEMU_VEC2(ushort2, unsigned short)
HIPEMU_VEC2(uchar2, unsigned char) HIPEMU_VEC2(char2, signed char)
HIPEMU_VEC4(int4, int) HIPEMU_VEC4(uint4, unsigned) HIPEMU_VEC4(float4, float) HIPEMU_VEC4(uchar4, unsigned char) HIPEMU_VEC4(ushort4, unsigned short)
struct int3 { int x, y, z; };
static inline int3 make_int3(int x, int y, int z) { return int3{x, y, z}; }

// ------------------------------------------------------------------------------------------------ the running lane
namespace hipemu {
struct Idx { unsigned x, y, z; };
extern Idx threadIdx_, blockIdx_, blockDim_, gridDim_;
extern int lane_;                       // flat thread id & 63
void sync_threads();                    // __syncthreads
// wave rendezvous at call site `site`: every lane posts `v`; afterwards posted(l) / mask() describe the lanes that took part
void wave_meet(const void *site, unsigned long long v);
unsigned long long wave_posted(int lane);
unsigned long long wave_mask();
void *dynamic_lds();
const void *kernarg();
typedef std::function<void()> Body;
void launch(dim3 grid, dim3 block, size_t lds, const Body &body, const void *kernarg, size_t kernarg_bytes);
}  // namespace hipemu
#define threadIdx (hipemu::threadIdx_)
#define blockIdx (hipemu::blockIdx_)
#define blockDim (hipemu::blockDim_)
#define gridDim (hipemu::gridDim_)
constexpr int warpSize = 64;

static inline void __syncthreads() { hipemu::sync_threads(); }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline void __threadfence_system() {}
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_s_sleep(...) ((void)0)
#define __builtin_amdgcn_kernarg_segment_ptr() (hipemu::kernarg())
#define HIPEMU_SITE() ([]() -> const void * { static const char here = 0; return &here; }())
#define __builtin_amdgcn_wave_barrier() hipemu::wave_meet(HIPEMU_SITE(), 0)
#define __ballot(p) hipemu::ballot_at(HIPEMU_SITE(), (p))
#define __shfl_down(v, d) hipemu::shfl_at(HIPEMU_SITE(), (v), hipemu::lane_ + (int)(d))
#define __shfl(v, l) hipemu::shfl_at(HIPEMU_SITE(), (v), (int)(l))
#define __shfl_xor(v, m) hipemu::shfl_at(HIPEMU_SITE(), (v), hipemu::lane_ ^ (int)(m))
#define __builtin_amdgcn_mbcnt_lo(m, acc) ((acc) + (unsigned)__builtin_popcount((unsigned)(m) & (hipemu::lane_ >= 32 ? 0xFFFFFFFFu : ((1u << hipemu::lane_) - 1u))))
#define __builtin_amdgcn_mbcnt_hi(m, acc) ((acc) + (unsigned)__builtin_popcount((unsigned)(m) & (hipemu::lane_ <= 32 ? 0u : ((1u << (hipemu::lane_ - 32)) - 1u))))
namespace hipemu {
static inline unsigned long long ballot_at(const void *site, bool p) {
    wave_meet(site, p ? 1ull : 0ull);
    unsigned long long m = wave_mask(), out = 0;
    for (int l = 0; l < 64; l++) if (((m >> l) & 1) && wave_posted(l)) out |= 1ull << l;
    return out;
}
template <typename T> static inline T shfl_at(const void *site, T v, int src) {
    static_assert(sizeof(T) <= 8, "shfl of a wide type");
    unsigned long long bits = 0;
    memcpy(&bits, &v, sizeof(T));
    wave_meet(site, bits);
    if (src < 0 || src >= 64 || !((wave_mask() >> src) & 1)) return v;
    bits = wave_posted(src);
    T out;
    memcpy(&out, &bits, sizeof(T));
    return out;
}
}  // namespace hipemu

// ------------------------------------------------------------------------------------------------ matrix cores
// v_mfma_f32_32x32x16_bf16 as a wave-wide meeting: every lane posts its 16 bytes of A and of B (four 8-byte posts), then computes
// its own 16 results from the gathered operands.  Lane maps as measured on the MI355X (tools/probe/mfma_layout.hip): lane l holds
// A[row = l & 31][k = 8 (l >> 5) + 0..7] and B[k = 8 (l >> 5) + 0..7][col = l & 31]; result register r of lane l is
// D[row = (r & 3) + 8 (r >> 2) + 4 (l >> 5)][col = l & 31].  The sum is taken in double and rounded once (the hardware's own
// internal order is not documented: tests compare with a tolerance, as they do on the GPU).
#define __builtin_amdgcn_readfirstlane(x) (x)
#define __builtin_amdgcn_s_waitcnt(x) ((void)0)
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z) hipemu::mfma_32x32x16_bf16((a), (b), (c))
namespace hipemu {
static inline float bf16_bits_to_float(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
template <typename V8, typename V16> static inline V16 mfma_32x32x16_bf16(V8 a, V8 b, V16 c) {
    static_assert(sizeof(V8) == 16 && sizeof(V16) == 64, "operand shapes of v_mfma_f32_32x32x16_bf16");
    static const char site[4] = {0, 0, 0, 0};
    unsigned long long mine[4], all[4][64];
    memcpy(mine, &a, 16);
    memcpy(mine + 2, &b, 16);
    for (int p = 0; p < 4; p++) {
        wave_meet(&site[p], mine[p]);
        if (wave_mask() != ~0ull) { fprintf(stderr, "hipemu: MFMA with inactive lanes\n"); abort(); }
        for (int l = 0; l < 64; l++) all[p][l] = wave_posted(l);
    }
    auto elem = [&](int operand, int lane, int e) {     // element e (0..7) of lane's A (operand 0) or B (operand 1)
        unsigned short h;
        memcpy(&h, (const unsigned char *)&all[2 * operand + (e >> 2)][lane] + 2 * (e & 3), 2);
        return (double)bf16_bits_to_float(h);
    };
    const int col = lane_ & 31, g = lane_ >> 5;
    V16 d;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        double acc = (double)c[r];
        for (int k = 0; k < 16; k++) acc += elem(0, row + 32 * (k >> 3), k & 7) * elem(1, col + 32 * (k >> 3), k & 7);
        d[r] = (float)acc;
    }
    return d;
}
// v_mfma_f32_32x32x2_f32: lane l holds A[row = l & 31][k = l >> 5] and B[k = l >> 5][col = l & 31], one float each; the result map is the
// 32 x 32 one above; D = fma(a_k1, b_k1, fma(a_k0, b_k0, C)) -- bit for bit a k-ordered chain of fmaf (cdna_hip_programming.md).
template <typename V16> static inline V16 mfma_32x32x2_f32(float a, float b, V16 c) {
    static_assert(sizeof(V16) == 64, "accumulator of v_mfma_f32_32x32x2_f32");
    static const char site = 0;
    unsigned long long mine, all[64];
    unsigned ua, ub;
    memcpy(&ua, &a, 4); memcpy(&ub, &b, 4);
    mine = ((unsigned long long)ub << 32) | ua;
    wave_meet(&site, mine);
    if (wave_mask() != ~0ull) { fprintf(stderr, "hipemu: MFMA with inactive lanes\n"); abort(); }
    for (int l = 0; l < 64; l++) all[l] = wave_posted(l);
    auto A = [&](int lane) { unsigned u = (unsigned)all[lane]; float f; memcpy(&f, &u, 4); return f; };
    auto B = [&](int lane) { unsigned u = (unsigned)(all[lane] >> 32); float f; memcpy(&f, &u, 4); return f; };
    const int col = lane_ & 31, g = lane_ >> 5;
    V16 d;
    for (int r = 0; r < 16; r++) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * g;
        d[r] = __builtin_fmaf(A(row + 32), B(col + 32), __builtin_fmaf(A(row), B(col), c[r]));
    }
    return d;
}
}  // namespace hipemu
#define __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, x, y, z) hipemu::mfma_32x32x2_f32((a), (b), (c))

// ------------------------------------------------------------------------------------------------ arithmetic intrinsics
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
static inline int __mulhi(int a, int b) { return (int)(((long long)a * b) >> 32); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __ffs(int v) { return __builtin_ffs(v); }
static inline int __ffsll(long long v) { return __builtin_ffsll(v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline float __fdiv_rn(float a, float b) { return a / b; }
static inline float __fadd_rn(float a, float b) { return a + b; }
static inline float __fmul_rn(float a, float b) { return a * b; }
static inline float __int_as_float(int v) { float f; memcpy(&f, &v, 4); return f; }
static inline float __uint_as_float(unsigned v) { float f; memcpy(&f, &v, 4); return f; }
static inline int __float_as_int(float f) { int v; memcpy(&v, &f, 4); return v; }
static inline unsigned __float_as_uint(float f) { unsigned v; memcpy(&v, &f, 4); return v; }
#define HIPEMU_MINMAX(T) static inline T min(T a, T b) { return b < a ? b : a; } static inline T max(T a, T b) { return a < b ? b : a; }
HIPEMU_MINMAX(int) HIPEMU_MINMAX(unsigned) HIPEMU_MINMAX(long long) HIPEMU_MINMAX(unsigned long long) HIPEMU_MINMAX(long) HIPEMU_MINMAX(unsigned long)
static inline float min(float a, float b) { return fminf(a, b); }
static inline float max(float a, float b) { return fmaxf(a, b); }
static inline long long wall_clock64() { return 0; }
static inline long long clock64() { return 0; }

// ------------------------------------------------------------------------------------------------ atomics (one host thread: plain read-modify-write)
#define HIPEMU_ATOMIC(NAME, EXPR) template <typename T, typename U> static inline T NAME(T *p, U v_) { T old = *p; T v = (T)v_; *p = (EXPR); return old; }
HIPEMU_ATOMIC(atomicAdd, old + v) HIPEMU_ATOMIC(atomicSub, old - v) HIPEMU_ATOMIC(atomicOr, old | v) HIPEMU_ATOMIC(atomicAnd, old & v)
HIPEMU_ATOMIC(atomicXor, old ^ v) HIPEMU_ATOMIC(atomicMin, v < old ? v : old) HIPEMU_ATOMIC(atomicMax, v > old ? v : old) HIPEMU_ATOMIC(atomicExch, v)
template <typename T, typename U, typename V> static inline T atomicCAS(T *p, U cmp, V val) { T old = *p; if (old == (T)cmp) *p = (T)val; return old; }
static inline unsigned atomicInc(unsigned *p, unsigned lim) { unsigned old = *p; *p = old >= lim ? 0 : old + 1; return old; }
#define __ATOMIC_RELAXED_HIPEMU 0
#define __HIP_MEMORY_SCOPE_SINGLETHREAD 1
#define __HIP_MEMORY_SCOPE_WAVEFRONT 2
#define __HIP_MEMORY_SCOPE_WORKGROUP 3
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) (*(volatile std::remove_reference_t<decltype(*(p))> *)(p))
#define __hip_atomic_store(p, v, order, scope) ((void)(*(volatile std::remove_reference_t<decltype(*(p))> *)(p) = (v)))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))
#define __hip_atomic_fetch_min(p, v, order, scope) atomicMin((p), (v))
#define __hip_atomic_fetch_max(p, v, order, scope) atomicMax((p), (v))
#define __hip_atomic_exchange(p, v, order, scope) atomicExch((p), (v))

// ------------------------------------------------------------------------------------------------ runtime API (host memory stands in for HBM)
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0, hipErrorNotReady = 600, hipErrorInvalidValue = 1;
struct hipemuStream;
typedef hipemuStream *hipStream_t;
struct hipemuEvent { double t; };
typedef hipemuEvent *hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
constexpr unsigned hipStreamNonBlocking = 1, hipStreamDefault = 0, hipHostMallocDefault = 0, hipEventDisableTiming = 2, hipEventDefault = 0;
enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
enum hipDeviceAttribute_t { hipDeviceAttributeMaxSharedMemoryPerBlock = 1, hipDeviceAttributeMultiprocessorCount = 2 };
struct hipFuncAttributes { size_t sharedSizeBytes, localSizeBytes, constSizeBytes; int numRegs, maxThreadsPerBlock, maxDynamicSharedSizeBytes; };

const char *hipGetErrorString(hipError_t e);
hipError_t hipGetLastError();
hipError_t hipGetDeviceCount(int *n);
hipError_t hipSetDevice(int d);
static inline hipError_t hipGetDevice(int *d) { *d = 0; return hipSuccess; }
static inline hipError_t hipStreamGetDevice(hipStream_t, int *d) { *d = 0; return hipSuccess; }
hipError_t hipDeviceSynchronize();
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int dev);
hipError_t hipMalloc(void **p, size_t n);
template <typename T> static inline hipError_t hipMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
hipError_t hipFree(void *p);
hipError_t hipHostMalloc(void **p, size_t n, unsigned flags = 0);
template <typename T> static inline hipError_t hipHostMalloc(T **p, size_t n, unsigned flags = 0) { return hipHostMalloc((void **)p, n, flags); }
hipError_t hipHostFree(void *p);
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k);
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st = nullptr);
hipError_t hipMemset(void *d, int v, size_t n);
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st = nullptr);
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags);
hipError_t hipStreamCreate(hipStream_t *s);
hipError_t hipStreamDestroy(hipStream_t s);
hipError_t hipStreamSynchronize(hipStream_t s);
hipError_t hipStreamQuery(hipStream_t s);
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags = 0);
hipError_t hipEventCreate(hipEvent_t *e);
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags);
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s = nullptr);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float *ms, hipEvent_t a, hipEvent_t b);
hipError_t hipFuncSetAttribute(const void *f, hipFuncAttribute a, int v);
hipError_t hipFuncGetAttributes(hipFuncAttributes *a, const void *f);

// ------------------------------------------------------------------------------------------------ kernel launch
namespace hipemu {
// the kernarg segment as the AMDGPU ABI lays it out: every argument at its natural alignment, in order
template <typename T> static inline void pack_arg(unsigned char *buf, size_t &off, const T &v) {
    off = (off + alignof(T) - 1) / alignof(T) * alignof(T);
    memcpy(buf + off, &v, sizeof(T));
    off += sizeof(T);
}
template <typename... KArgs, typename... Args>
static inline void launch_kernel(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t lds, hipStream_t, Args &&...args) {
    static_assert(sizeof...(KArgs) == sizeof...(Args), "kernel launched with the wrong number of arguments");
    std::tuple<std::decay_t<KArgs>...> packed{static_cast<std::decay_t<KArgs>>(args)...};
    static thread_local unsigned char karg[1 << 16];
    size_t off = 0;
    std::apply([&](const auto &...a) { (pack_arg(karg, off, a), ...); }, packed);
    launch(grid, block, lds, [&]() { std::apply(kernel, packed); }, karg, off);
}
}  // namespace hipemu
#define hipLaunchKernelGGL(kernel, ...) hipemu::launch_kernel(kernel, __VA_ARGS__)
